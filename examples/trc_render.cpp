// trc_render.cpp -- a C++ host driving the path through the C ABI only (include/tracer_abi.h), the way
// -[AAPLRenderer render:] drives the Metal path: scene prep on the host (libtrc_host.so), upload, N samples,
// output stage, PNG.  Nothing but the two shared libraries is involved.
//
//   trc_render [--scene cornell|spheres|volume] [--integrator path|mis|volume] [--size W H] [--spp N]
//              [--mesh file.obj|file.pbrt] [--density cloud.pbrt] [--lbvh | --device-sah] [--sobol] [--out frame.png]
//   trc_render --pbrt scene.pbrt [--integrator path|mis] [--spp N] [--size W H] [--out frame.png]
//              a whole pbrt-v3 scene (camera, film, lights, materials, spheres, meshes: trc_host_scene_load_pbrt)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "tracer_abi.h"

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        trc_status st_ = (call);                                                                     \
        if (st_ != TRC_OK) {                                                                         \
            std::fprintf(stderr, "%s failed: %s (%s)\n", #call, trc_status_string(st_), ctx ? trc_last_error(ctx) : ""); \
            return 1;                                                                                \
        }                                                                                            \
    } while (0)

int main(int argc, char** argv) {
    std::string scene_name = "spheres", integ_name = "path", out = "frame.png", mesh_path, density_path, pbrt_path, hdr_path;
    uint32_t W = 640, H = 360, spp = 64;
    bool lbvh = false, device_sah = false, sobol = false, size_given = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--scene" && i + 1 < argc) scene_name = argv[++i];
        else if (a == "--integrator" && i + 1 < argc) integ_name = argv[++i];
        else if (a == "--size" && i + 2 < argc) { W = (uint32_t)std::atoi(argv[++i]); H = (uint32_t)std::atoi(argv[++i]); size_given = true; }
        else if (a == "--pbrt" && i + 1 < argc) pbrt_path = argv[++i];          // a whole pbrt-v3 scene
        else if (a == "--spp" && i + 1 < argc) spp = (uint32_t)std::atoi(argv[++i]);
        else if (a == "--mesh" && i + 1 < argc) mesh_path = argv[++i];          // Wavefront OBJ, PLY or pbrt-v3 trianglemeshes
        else if (a == "--hdr" && i + 1 < argc) hdr_path = argv[++i];            // Radiance .hdr backdrop (the reference's texHDR)
        else if (a == "--density" && i + 1 < argc) density_path = argv[++i];    // pbrt-v3 heterogeneous medium
        else if (a == "--lbvh") lbvh = true;
        else if (a == "--device-sah") device_sah = true;
        else if (a == "--sobol") sobol = true;
        else if (a == "--out" && i + 1 < argc) out = argv[++i];
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    int kind = scene_name == "cornell" ? TRC_SCENE_CORNELL : scene_name == "volume" ? TRC_SCENE_CORNELL_VOLUME : TRC_SCENE_CORNELL_SPHERES;
    if (!mesh_path.empty() && kind != TRC_SCENE_CORNELL_VOLUME) kind = TRC_SCENE_CORNELL_MESH;   // the mesh slot of AAPLRenderer.mm:474-603
    const uint32_t integrator = integ_name == "mis" ? TRC_INTEGRATOR_MIS : integ_name == "volume" ? TRC_INTEGRATOR_VOLUME : TRC_INTEGRATOR_PATH;

    trc_ctx* ctx = nullptr;
    trc_host_scene* hs = nullptr;
    trc_host_mesh* mesh = nullptr;
    const trc_TriangleVertex* mv = nullptr; const uint32_t* mi = nullptr;
    uint32_t n_mv = 0, n_mi = 0;
    if (!mesh_path.empty()) {
        const bool pbrt = mesh_path.size() > 5 && mesh_path.compare(mesh_path.size() - 5, 5, ".pbrt") == 0;
        const bool ply = mesh_path.size() > 4 && mesh_path.compare(mesh_path.size() - 4, 4, ".ply") == 0;
        if ((pbrt ? trc_host_mesh_load_pbrt(mesh_path.c_str(), &mesh) : ply ? trc_host_mesh_load_ply(mesh_path.c_str(), &mesh)
                  : trc_host_mesh_load_obj(mesh_path.c_str(), &mesh)) != TRC_OK) {
            std::fprintf(stderr, "cannot read a triangle mesh from %s\n", mesh_path.c_str());
            return 1;
        }
        trc_host_mesh_view(mesh, &mv, &n_mv, &mi, &n_mi);
    }
    trc_Camera cam;
    if (!pbrt_path.empty()) {
        trc_pbrt_info info;
        if (trc_host_scene_load_pbrt(pbrt_path.c_str(), &hs, &cam, &info, nullptr, 0) != TRC_OK) {
            std::fprintf(stderr, "cannot read a scene from %s\n", pbrt_path.c_str());
            return 1;
        }
        if (!size_given) { W = info.xres; H = info.yres; }         // the camera's aspect is the film's
        if (integrator != TRC_INTEGRATOR_PATH && !info.mis_ready) {
            std::fprintf(stderr, "%s has no rectangular area light: traceMIS samples squareList[5] / [6]; use --integrator path\n", pbrt_path.c_str());
            return 1;
        }
        scene_name = pbrt_path;
        std::fprintf(stderr, "%s: %u shapes (%u not handled), %u materials and %u textures not handled\n", pbrt_path.c_str(), info.n_shapes,
                     info.n_unsupported_shapes, info.n_unsupported_materials, info.n_unsupported_textures);
    } else {
        // --device-sah: the host prepares its analytic primitives and the mesh only -- no leaf record per triangle, no tree
        if (trc_host_scene_create_leaves(kind, mv, n_mv, mi, n_mi, device_sah ? 1 : 0, &hs) != TRC_OK) { std::fprintf(stderr, "scene prep failed\n"); return 1; }
        trc_host_prepare_camera(&cam, (float)W, (float)H);
    }
    trc_scene scene;
    trc_host_scene_view(hs, &scene);

    CHECK(trc_create(0, &ctx));
    if (device_sah && pbrt_path.empty()) {        // triangle leaves + BVH::buildTree itself, both on the device
        CHECK(trc_upload_scene_device(ctx, &scene, TRC_TREE_SAH | TRC_TREE_TRIANGLE_LEAVES));
    } else if (lbvh || device_sah) {              // hand over the leaf records only; the tree is built on the GPU
        trc_scene leaves = scene;
        leaves.bvhList = scene.bvhList + 1;       // BVH::buildTree keeps the leaves at [1, n]
        leaves.n_bvh = (scene.n_bvh + 1) / 2;
        if (device_sah) CHECK(trc_upload_scene_sah(ctx, &leaves));
        else CHECK(trc_upload_scene_lbvh(ctx, &leaves));
    } else {
        CHECK(trc_upload_scene(ctx, &scene));
    }
    std::vector<float> cloud;
    if (kind == TRC_SCENE_CORNELL_VOLUME) {
        uint32_t nx = 100, ny = 100, nz = 40;
        if (!density_path.empty()) {              // what AAPLRenderer.mm:629-636 reads through minipbrt
            float* grid = nullptr;
            if (trc_host_load_density_pbrt(density_path.c_str(), &nx, &ny, &nz, &grid) != TRC_OK) {
                std::fprintf(stderr, "cannot read a density grid from %s\n", density_path.c_str());
                return 1;
            }
            cloud.assign(grid, grid + (size_t)nx * ny * nz);
            trc_host_free(grid);
        } else {
            cloud.resize((size_t)nx * ny * nz);
            trc_host_make_cloud(nx, ny, nz, 1, cloud.data());
        }
        trc_GridDensityInfo info;
        trc_host_make_density_info(10.0f, 90.0f, 0.5f, nx, ny, nz, cloud.data(), &info);
        CHECK(trc_upload_density(ctx, &info, cloud.data()));
    }
    if (!hdr_path.empty()) {                     // AAPLRenderer.mm:352-383: the equirectangular backdrop, rows bottom-up
        uint32_t ew = 0, eh = 0;
        float* env = nullptr;
        if (trc_host_load_hdr(hdr_path.c_str(), &ew, &eh, &env) != TRC_OK) { std::fprintf(stderr, "cannot read a Radiance .hdr image from %s\n", hdr_path.c_str()); return 1; }
        const trc_status es = trc_set_environment_map(ctx, ew, eh, env);
        trc_host_free(env);
        CHECK(es);
    }
    CHECK(trc_set_camera(ctx, &cam));
    CHECK(trc_resize(ctx, W, H));
    CHECK(trc_seed(ctx, 0x5EED0000ull));

    trc_params prm;
    std::memset(&prm, 0, sizeof prm);
    prm.spp = spp; prm.max_depth = 8; prm.integrator = integrator; prm.tile_nranks = 1;
    if (sobol) prm.flags |= TRC_FLAG_SOBOL;      // pbrt::SobolSampler instead of the random sampler (Render.metal:529-530)
    const auto t0 = std::chrono::steady_clock::now();
    CHECK(trc_render(ctx, &prm));
    CHECK(trc_synchronize(ctx));
    const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    trc_stats st;
    CHECK(trc_get_stats(ctx, &st));

    std::vector<uint8_t> rgba8((size_t)W * H * 4);
    float exposure = 0;
    CHECK(trc_tonemap(ctx, rgba8.data(), &exposure));
    if (trc_host_write_png(out.c_str(), rgba8.data(), W, H) != TRC_OK) { std::fprintf(stderr, "cannot write %s\n", out.c_str()); return 1; }
    std::printf("%s %ux%u %u spp %s: %llu rays, kernel %.2f ms (wall %.2f ms), %.1f Mrays/s, exposure %.4f -> %s\n",
                scene_name.c_str(), W, H, spp, integ_name.c_str(), (unsigned long long)st.rays, st.kernel_ms, wall_ms,
                st.rays / st.kernel_ms / 1e3, exposure, out.c_str());
    trc_destroy(ctx);
    trc_host_scene_destroy(hs);
    trc_host_mesh_destroy(mesh);
    return 0;
}

// tools/sanitize/driver.cpp -- the CPU half of the repository (libtrc_host sources + the oracle, which is test
// infrastructure) as ONE program for the sanitizers: `make asan` / `make tsan` compile the sources themselves with
// -fsanitize=address,undefined / -fsanitize=thread (no Python in the process, so no interposer noise), and this driver
// runs what the threaded and the parsing code paths do:
//   1. scene assembly + the multi-threaded SAH build on a 150 k-triangle mesh (host/bvh_builder.cpp, the GCD build of
//      BVH.hh:35-244 restated with std::thread), tree depth, a second build in parallel from another thread
//   2. the oracle's row-band workers: tracePath / traceMIS / traceVolume frames with 8 threads, the SPPM pass, LBVH
//   3. every file reader on well-formed files it writes itself (pbrt scene with plymesh / disk / cylinder / texture, PLY in
//      three encodings, OBJ, pbrt density, Radiance .hdr), then on `--fuzz N` mutations of each: a reader must return a
//      status, never crash, over-read or leak
// "build host code under -fsanitize=thread,address".
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "tracer_abi.h"
#include "../../oracle/oracle.h"

static int g_fail = 0;
#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "EXPECT failed: %s (%s:%d)\n", #c, __FILE__, __LINE__); ++g_fail; } } while (0)

static void write_file(const std::string& p, const std::string& data) {
    FILE* f = std::fopen(p.c_str(), "wb");
    if (!f) { std::perror(p.c_str()); std::exit(2); }
    std::fwrite(data.data(), 1, data.size(), f);
    std::fclose(f);
}
static std::string read_file(const std::string& p) {
    FILE* f = std::fopen(p.c_str(), "rb");
    std::string s;
    if (!f) return s;
    char buf[65536]; size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
    std::fclose(f);
    return s;
}

static void build_and_render(bool second_thread) {
    trc_host_mesh *ball = nullptr, *many = nullptr;
    EXPECT(trc_host_mesh_make_ball(80, 80, 0.07f, &ball) == TRC_OK);
    EXPECT(trc_host_mesh_replicate(ball, second_thread ? 2 : 3, 2.4f, &many) == TRC_OK);     // 12 800 x 9 = 115 200 triangles
    const trc_TriangleVertex* v; const uint32_t* idx; uint32_t nv, ni;
    trc_host_mesh_view(many, &v, &nv, &idx, &ni);
    trc_host_scene* hs = nullptr;
    EXPECT(trc_host_scene_create(TRC_SCENE_CORNELL_MESH, v, nv, idx, ni, &hs) == TRC_OK);
    trc_scene sc;
    trc_host_scene_view(hs, &sc);
    uint32_t depth = 0;
    EXPECT(trc_host_tree_depth(sc.bvhList, sc.n_bvh, &depth) == TRC_OK && depth > 10 && depth <= TRC_MAX_BVH_DEPTH);
    if (!second_thread) {
        // the oracle on that scene: 8 row-band workers, all three integrators
        const uint32_t W = 48, H = 32;
        trc_Camera cam;
        trc_host_prepare_camera(&cam, (float)W, (float)H);
        const float env[3] = {0.1f, 0.2f, 0.3f};
        for (uint32_t integ = 0; integ < 3; ++integ) {
            std::vector<uint32_t> rng((size_t)W * H * 4);
            std::vector<float> acc((size_t)W * H * 4, 0.0f), acc1((size_t)W * H * 4, 0.0f);
            trc_host_fill_rng(7, W, H, rng.data());
            std::vector<uint32_t> rng1 = rng;
            trc_params prm; std::memset(&prm, 0, sizeof prm);
            prm.spp = 2; prm.max_depth = 8; prm.integrator = integ; prm.tile_nranks = 1;
            trc_stats st, st1;                     // orc_render ADDS to the counters
            std::memset(&st, 0, sizeof st); std::memset(&st1, 0, sizeof st1);
            orc_render(&sc, &cam, env, W, H, rng.data(), acc.data(), &prm, &st, 8);
            orc_render(&sc, &cam, env, W, H, rng1.data(), acc1.data(), &prm, &st1, 1);
            EXPECT(st.rays == st1.rays);                                                  // thread-count independent
            size_t diff = 0;
            for (size_t k = 0; k < acc.size(); ++k) diff += std::memcmp(&acc[k], &acc1[k], 4) != 0;
            if (diff || st.rays != st1.rays) std::fprintf(stderr, "integrator %u: rays %llu vs %llu, %zu of %zu accumulator words differ\n", integ,
                                                          (unsigned long long)st.rays, (unsigned long long)st1.rays, diff, acc.size());
            EXPECT(diff == 0);
        }
        // LBVH oracle on the leaves
        const uint32_t n_leaves = (sc.n_bvh + 1) / 2;
        std::vector<trc_BVH> out(2 * (size_t)n_leaves - 1);
        uint32_t h = 0;
        orc_lbvh_build(sc.bvhList + 1, n_leaves, out.data(), &h);
        EXPECT(h > 10);
    }
    trc_host_scene_destroy(hs);
    trc_host_mesh_destroy(many);
    trc_host_mesh_destroy(ball);
}

static void sppm_pass() {
    trc_host_scene* hs = nullptr;
    EXPECT(trc_host_scene_create(TRC_SCENE_CORNELL_SPHERES, nullptr, 0, nullptr, 0, &hs) == TRC_OK);
    trc_scene sc; trc_host_scene_view(hs, &sc);
    const uint32_t W = 40, H = 24;
    trc_Camera cam; trc_host_prepare_camera(&cam, (float)W, (float)H);
    std::vector<uint32_t> rng((size_t)W * H * 4);
    std::vector<float> acc((size_t)W * H * 4, 0.0f);
    trc_host_fill_rng(3, W, H, rng.data());
    const float env[3] = {0, 0, 0};
    orc_sppm* s = orc_sppm_create(W, H, 5);
    orc_sppm_frames(s, &sc, &cam, env, rng.data(), acc.data(), 2);
    trc_Complex cx;
    orc_sppm_download(s, nullptr, nullptr, nullptr, nullptr, &cx);
    EXPECT(cx.frame_count == 2);
    orc_sppm_destroy(s);
    std::vector<uint8_t> rgba8((size_t)W * H * 4);
    float expose = 0;
    orc_tonemap(acc.data(), W, H, rgba8.data(), &expose);
    trc_host_scene_destroy(hs);
}

// ---------------------------------------------------------------- readers
static const char* kScene =
    "LookAt 0 3 -12  0 1 0  0 1 0\nCamera \"perspective\" \"float fov\" [ 40 ]\nFilm \"image\" \"integer xresolution\" [ 64 ] \"integer yresolution\" [ 48 ]\n"
    "WorldBegin\nAttributeBegin\n AreaLightSource \"diffuse\" \"rgb L\" [ 9 9 8 ]\n"
    " Shape \"trianglemesh\" \"integer indices\" [ 0 1 2 0 2 3 ] \"point P\" [ -3 8 -3  3 8 -3  3 8 3  -3 8 3 ]\nAttributeEnd\n"
    "Texture \"tiles\" \"spectrum\" \"checkerboard\" \"rgb tex1\" [ 0.9 0.5 0.1 ]\nMakeNamedMaterial \"m\" \"string type\" \"glass\"\n"
    "Material \"matte\" \"texture Kd\" \"tiles\"\nShape \"sphere\" \"float radius\" 1.5\nNamedMaterial \"m\"\n"
    "AttributeBegin\n Translate 2.5 0 0 Rotate -90 1 0 0\n Shape \"cylinder\" \"float radius\" 1.2 \"float zmin\" 0 \"float zmax\" 2.5 \"float phimax\" 200\n"
    " Shape \"disk\" \"float radius\" 1.2 \"float innerradius\" 0.3\n Shape \"plymesh\" \"string filename\" \"mesh.ply\"\nAttributeEnd\n"
    "Include \"medium.pbrt\"\nWorldEnd\n";
static const char* kMedium =
    "MakeNamedMedium \"smoke\" \"string type\" \"heterogeneous\" \"integer nx\" 2 \"integer ny\" 2 \"integer nz\" 2\n \"float density\" [ 0 1 .5 .25 1 1 0 0 ]\n";
static const char* kObj = "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvn 0 0 1\nf 1/1/1 2/2/1 3/1/1 4/2/1\nf -4 -3 -2\n";

static std::string make_ply(int fmt) {      // 0 ascii, 1 little, 2 big endian
    const float P[4][3] = {{0, 0, 0}, {1, 0, 0}, {1, 1, 0.2f}, {0, 1, 0}};
    std::string s = "ply\nformat ";
    s += fmt == 0 ? "ascii" : fmt == 1 ? "binary_little_endian" : "binary_big_endian";
    s += " 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nproperty uchar c\nelement face 2\nproperty list uchar int vertex_indices\nend_header\n";
    auto put = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t k = 0; k < n; ++k) s.push_back((char)b[fmt == 2 ? n - 1 - k : k]); };
    if (fmt == 0) { for (auto& p : P) { char l[96]; std::snprintf(l, sizeof l, "%g %g %g 7\n", p[0], p[1], p[2]); s += l; } s += "3 0 1 2\n4 0 1 2 3\n"; }
    else {
        for (auto& p : P) { for (float f : p) put(&f, 4); s.push_back(7); }
        const int tri[3] = {0, 1, 2}, quad[4] = {0, 1, 2, 3};
        s.push_back(3); for (int i : tri) put(&i, 4);
        s.push_back(4); for (int i : quad) put(&i, 4);
    }
    return s;
}
static std::string make_hdr() {
    std::string s = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 3 +X 10\n";
    for (int y = 0; y < 3; ++y) {
        s += std::string("\x02\x02\x00\x0a", 4);
        for (int c = 0; c < 4; ++c) { s.push_back((char)(128 + 6)); s.push_back((char)(40 * c + y)); s.push_back((char)4); for (int k = 0; k < 4; ++k) s.push_back((char)(100 + k + c)); }
    }
    return s;
}

struct Reader { const char* name; std::string file; int (*load)(const std::string&); };
static int load_scene(const std::string& p) {
    trc_host_scene* hs = nullptr; trc_Camera cam; trc_pbrt_info info; trc_pbrt_shape shapes[8];
    const trc_status st = trc_host_scene_load_pbrt(p.c_str(), &hs, &cam, &info, shapes, 8);
    if (st == TRC_OK) { trc_scene v; trc_host_scene_view(hs, &v); EXPECT(v.n_bvh >= 3); trc_host_scene_destroy(hs); }
    return st;
}
static int load_mesh_pbrt(const std::string& p) { trc_host_mesh* m = nullptr; const trc_status st = trc_host_mesh_load_pbrt(p.c_str(), &m); if (st == TRC_OK) trc_host_mesh_destroy(m); return st; }
static int load_obj(const std::string& p) { trc_host_mesh* m = nullptr; const trc_status st = trc_host_mesh_load_obj(p.c_str(), &m); if (st == TRC_OK) trc_host_mesh_destroy(m); return st; }
static int load_ply(const std::string& p) { trc_host_mesh* m = nullptr; const trc_status st = trc_host_mesh_load_ply(p.c_str(), &m); if (st == TRC_OK) trc_host_mesh_destroy(m); return st; }
static int load_density(const std::string& p) { uint32_t nx, ny, nz; float* g = nullptr; const trc_status st = trc_host_load_density_pbrt(p.c_str(), &nx, &ny, &nz, &g); if (st == TRC_OK) trc_host_free(g); return st; }
static int load_hdr(const std::string& p) { uint32_t w, h; float* g = nullptr; const trc_status st = trc_host_load_hdr(p.c_str(), &w, &h, &g); if (st == TRC_OK) trc_host_free(g); return st; }

int main(int argc, char** argv) {
    int n_fuzz = 2000;
    std::string dir = "/tmp/trc_sanitize";
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--fuzz") && i + 1 < argc) n_fuzz = std::atoi(argv[++i]);
        else if (!std::strcmp(argv[i], "--dir") && i + 1 < argc) dir = argv[++i];
    }
    (void)!std::system(("mkdir -p " + dir).c_str());

    std::thread other([] { build_and_render(true); });       // two SAH builds at once: the builder's threads + ours
    build_and_render(false);
    other.join();
    sppm_pass();
    { uint32_t m32[40 * 52]; trc_host_sobol_matrices32(m32); uint64_t a[52], b[52]; EXPECT(trc_host_sobol_interval_tables(11, a, b) == TRC_OK); }

    write_file(dir + "/scene.pbrt", kScene);
    write_file(dir + "/medium.pbrt", std::string("WorldBegin\n") + kMedium + "WorldEnd\n");
    write_file(dir + "/mesh.ply", make_ply(1));
    std::vector<Reader> readers = {
        {"pbrt scene", dir + "/scene.pbrt", load_scene}, {"pbrt meshes", dir + "/scene.pbrt", load_mesh_pbrt}, {"pbrt density", dir + "/medium.pbrt", load_density},
        {"ply ascii", dir + "/a.ply", load_ply}, {"ply little", dir + "/mesh.ply", load_ply}, {"ply big", dir + "/b.ply", load_ply},
        {"obj", dir + "/m.obj", load_obj}, {"hdr", dir + "/sky.hdr", load_hdr}};
    write_file(dir + "/a.ply", make_ply(0)); write_file(dir + "/b.ply", make_ply(2));
    write_file(dir + "/m.obj", kObj); write_file(dir + "/sky.hdr", make_hdr());
    for (const Reader& r : readers) { const int st = r.load(r.file); if (st != TRC_OK) { std::fprintf(stderr, "%s: well-formed file rejected (%d)\n", r.name, st); ++g_fail; } }

    std::mt19937 gen(12345);
    size_t loaded = 0, rejected = 0;
    for (const Reader& r : readers) {
        const std::string good = read_file(r.file);
        const std::string victim = dir + "/fuzz_input" + r.file.substr(r.file.find_last_of('.'));
        for (int k = 0; k < n_fuzz; ++k) {
            std::string m = good;
            const int edits = 1 + (int)(gen() % 4);
            for (int e = 0; e < edits && !m.empty(); ++e) {
                const size_t at = gen() % m.size();
                switch (gen() % 6) {
                    case 0: m[at] = (char)(gen() & 0xFF); break;                              // flip a byte
                    case 1: m.erase(at, 1 + gen() % 8); break;                                // drop a few
                    case 2: m.insert(at, std::string(1 + gen() % 6, (char)('0' + gen() % 10))); break;   // digits
                    case 3: m.resize(at); break;                                              // truncate
                    case 4: m.insert(at, m.substr(gen() % m.size(), gen() % 16)); break;      // duplicate a piece
                    default: m[at] = "[]\"-.e \n"[gen() % 8]; break;                          // a syntax character
                }
            }
            write_file(victim, m);
            if (r.load(victim) == TRC_OK) ++loaded; else ++rejected;
        }
    }
    std::printf("sanitize driver: %d expectation failures; fuzz: %zu mutated files loaded, %zu rejected with a status\n", g_fail, loaded, rejected);
    return g_fail ? 1 : 0;
}

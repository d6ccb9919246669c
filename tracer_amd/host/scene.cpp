// scene.cpp -- host scene construction with the reference's constants and ordering.
//
//   cubes + materials 0-2      RT_Metal/Tracer/Tracer.mm:174-243  (prepareCubeList)
//   Cornell squares + mat 3-6  Tracer.mm:245-304                  (prepareCornellBox)
//   spheres + materials 7-18   Tracer.mm:306-369                  (prepareSphereList)
//   material 19                RT_Metal/Tracer/AAPLRenderer.mm:233-242 (testMaterial)
//   leaves and tree            AAPLRenderer.mm:454-468,546-610
//   mesh placement             AAPLRenderer.mm:513-525,560-573
//   camera                     Tracer.mm:87-125,371-411
//   RNG texture                AAPLRenderer.mm:296-344 (arc4random -> deterministic per-pixel PCG32)
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <new>
#include <vector>

#include "host_math.hpp"
#include "host_scene.hpp"
#include "pbrt_text.hpp"
#include "trc_sobol.h"

using namespace trc;

namespace {

void prepare_cube_list(trc_host_scene& s) {
    const float pi = (float)M_PI;
    trc_Material metal = make_material(TRC_MAT_METAL);
    metal.textureInfo.albedo = f3(0.8f, 0.85f, 0.88f);
    metal.specular = 1;
    const uint32_t metal_index = (uint32_t)s.materials.size();
    s.materials.push_back(metal);
    s.cubes.push_back(make_cube(metal_index, translation4x4(265, 1, 295),
                                rotation4x4((float)(M_PI * 15 / 180), f3(0, 1, 0)), scale4x4(165, 330, 165)));

    trc_Material white = make_material(TRC_MAT_GLASS);
    white.textureInfo.albedo = f3(1, 1, 1);
    white.eta = 0.01f;
    s.materials.push_back(white);
    // the smaller cube uses material 19 (testMaterial), Tracer.mm:208-209
    s.cubes.push_back(make_cube(19, translation4x4(130, 1, 65),
                                rotation4x4((float)(-0.1 * M_PI), f3(0, 1, 0)), scale4x4(165, 165, 165)));

    trc_Material density = make_material(TRC_MAT_NIL);
    density.medium = TRC_MEDIUM_GRIDDENSITY;
    density.textureInfo.albedo = f3(1, 1, 1);
    const uint32_t density_index = (uint32_t)s.materials.size();
    s.materials.push_back(density);
    // density container: present in the cube list, never inserted in the BVH (AAPLRenderer.mm:459)
    s.cubes.push_back(make_cube(density_index, translation4x4(0, 200, 65),
                                rotation4x4(0.0f * pi, f3(0, 1, 0)), scale4x4(200, 200, 80)));
}

void prepare_cornell_box(trc_host_scene& s) {
    trc_Material light = make_material(TRC_MAT_DIFFUSE);
    light.textureInfo.albedo = f3(11.0f);
    const uint32_t light_index = (uint32_t)s.materials.size();
    s.materials.push_back(light);

    trc_Material red = make_material(TRC_MAT_LAMBERT);
    red.textureInfo.albedo = f3(0.65f, 0.05f, 0.05f);
    const uint32_t red_index = (uint32_t)s.materials.size();
    s.materials.push_back(red);

    trc_Material green = make_material(TRC_MAT_LAMBERT);
    green.textureInfo.albedo = f3(0.05f, 0.65f, 0.05f);
    const uint32_t green_index = (uint32_t)s.materials.size();
    s.materials.push_back(green);

    trc_Material white = make_material(TRC_MAT_LAMBERT);
    white.textureInfo.type = TRC_TEX_CHECKER;
    white.textureInfo.albedo = f3(0.73f);
    const uint32_t white_index = (uint32_t)s.materials.size();
    s.materials.push_back(white);

    // list order fixes the indices the integrators hard-code (squareList[5], [6] are the lights)
    s.squares.push_back(make_square(1, 0, 555, 2, 0, 555, 0, -245, green_index));            // 0 left
    s.squares.push_back(make_square(1, 0, 555, 2, 0, 555, 0, 800, red_index));               // 1 right
    s.squares.push_back(make_square(0, -245, 800, 2, 0, 555, 1, 555, white_index));          // 2 top
    s.squares.push_back(make_square(0, -245, 800, 1, 0, 555, 2, 555, white_index));          // 3 back
    s.squares.push_back(make_square(0, -245, 800, 2, 0, 555, 1, 0, white_index));            // 4 bottom
    s.squares.push_back(make_square(0, 400, 555, 2, 200, 355, 1, (float)(555 - 0.1), light_index));  // 5 light
    s.squares.push_back(make_square(1, 200, 300, 2, 200, 300, 0, -300, light_index));        // 6 little light
}

void prepare_sphere_list(trc_host_scene& s, bool remap_materials) {
    // Reference material types here (Dielectric, Demofox) make Material::S_F return 0
    // (Material.hh:143-144), i.e. black absorbers.  BASELINE config 2 remaps them to the
    // supported lobes so every BSDF of the path is exercised; albedos are kept.
    static const int32_t bottom_row[6] = {TRC_MAT_LAMBERT, TRC_MAT_PLASTIC, TRC_MAT_METAL,
                                          TRC_MAT_GLASS, TRC_MAT_LAMBERT, TRC_MAT_PLASTIC};
    static const int32_t top_row[5] = {TRC_MAT_METAL, TRC_MAT_GLASS, TRC_MAT_PLASTIC,
                                       TRC_MAT_METAL, TRC_MAT_LAMBERT};

    trc_Material glass = make_material(remap_materials ? TRC_MAT_GLASS : TRC_MAT_DIELECTRIC);
    glass.textureInfo.albedo = f3(1.0f);
    glass.textureInfo.type = remap_materials ? TRC_TEX_CONSTANT : TRC_TEX_NOISE;
    glass.eta = 1.5f;
    uint32_t index = (uint32_t)s.materials.size();
    s.materials.push_back(glass);
    s.spheres.push_back(make_sphere(64, f3(200, 250, 200), index));

    for (int i = 0; i < 6; ++i) {
        trc_Material specu = make_material(remap_materials ? bottom_row[i] : TRC_MAT_DEMOFOX);
        specu.textureInfo.albedo = f3(0.9f, 0.25f, 0.25f);
        index = (uint32_t)s.materials.size();
        s.materials.push_back(specu);
        s.spheres.push_back(make_sphere(40, f3(0.0f + 100.0f * (5 - i), 50, 50), index));
    }
    for (int i = 0; i < 5; ++i) {
        trc_Material gloss = make_material(remap_materials ? top_row[i] : TRC_MAT_DEMOFOX);
        gloss.textureInfo.albedo = f3(1.0f);
        index = (uint32_t)s.materials.size();
        s.materials.push_back(gloss);
        s.spheres.push_back(make_sphere(40, f3(-10.0f + 150.0f * i, 500, 400), index));
    }
}

}  // namespace

extern "C" {

void trc_host_make_camera(trc_Camera* camera, const float lookFrom[3], const float lookAt[3],
                          const float viewUp[3], float aperture, float aspect,
                          float vfov, float focus_dist) {
    std::memset(camera, 0, sizeof(*camera));
    const trc_float3 from = f3(lookFrom[0], lookFrom[1], lookFrom[2]);
    const trc_float3 at = f3(lookAt[0], lookAt[1], lookAt[2]);
    const trc_float3 up = f3(viewUp[0], viewUp[1], viewUp[2]);
    camera->lookFrom = from; camera->lookAt = at; camera->viewUp = up;
    camera->aperture = aperture; camera->aspect = aspect; camera->vfov = vfov;
    camera->focus_dist = focus_dist;
    camera->lenRadius = aperture / 2;
    const float halfHeight = std::tan(vfov / 2);      // vfov already in radians (Tracer.mm:107)
    const float halfWidth = aspect * halfHeight;
    const trc_float3 w = normalize(from - at);
    const trc_float3 u = normalize(cross(up, w));
    const trc_float3 v = cross(w, u);
    camera->u = u; camera->v = v; camera->w = w;
    const trc_float3 vertical = v * (2 * halfHeight * focus_dist);
    const trc_float3 horizontal = u * (2 * halfWidth * focus_dist);
    camera->vertical = vertical;
    camera->horizontal = horizontal;
    camera->cornerLowLeft = from - vertical / 2.0f - horizontal / 2.0f - w * focus_dist;
}

// prepareCamera with zero rotation / zero wasd offset (Tracer.mm:371-411)
void trc_host_prepare_camera(trc_Camera* out, float width, float height) {
    const float from[3] = {278, 278, -800}, at[3] = {278, 278, 278}, up[3] = {0, 1, 0};
    trc_host_make_camera(out, from, at, up, 0.0f, width / height, (float)(45 * (M_PI / 180)), 10.0f);
}

// PCG32 (RT_Metal/Tracer/pcg_basic.c:42-72)
static inline uint32_t pcg32_next(uint64_t& state, uint64_t inc) {
    uint64_t old = state;
    state = old * 6364136223846793005ULL + inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
}

void trc_host_fill_rng(uint64_t seed, uint32_t width, uint32_t height, uint32_t* rgba) {
    const uint64_t n = (uint64_t)width * height;
    for (uint64_t p = 0; p < n; ++p) {
        // pcg32_srandom_r(rng, seed, seq = pixel index)
        uint64_t state = 0, inc = (p << 1u) | 1u;
        pcg32_next(state, inc);
        state += seed;
        pcg32_next(state, inc);
        for (int c = 0; c < 4; ++c) rgba[4 * p + c] = pcg32_next(state, inc);
    }
}

trc_status trc_host_scene_create(int32_t kind, const trc_TriangleVertex* mesh_vertices, uint32_t n_vertices,
                                 const uint32_t* mesh_indices, uint32_t n_indices, trc_host_scene** out) {
    return trc_host_scene_create_leaves(kind, mesh_vertices, n_vertices, mesh_indices, n_indices, 0, out);
}

trc_status trc_host_scene_create_leaves(int32_t kind, const trc_TriangleVertex* mesh_vertices, uint32_t n_vertices,
                                        const uint32_t* mesh_indices, uint32_t n_indices, int32_t analytic_leaves_only, trc_host_scene** out) {
    if (!out) return TRC_ERR_INVALID_ARG;
    if (kind < TRC_SCENE_CORNELL || kind > TRC_SCENE_CORNELL_VOLUME) return TRC_ERR_INVALID_ARG;
    const bool with_mesh = kind == TRC_SCENE_CORNELL_MESH || (kind == TRC_SCENE_CORNELL_VOLUME && mesh_vertices != nullptr);
    if (with_mesh && (!mesh_vertices || !mesh_indices || n_indices < 3 || n_indices % 3))
        return TRC_ERR_INVALID_ARG;
    trc_host_scene* s = new (std::nothrow) trc_host_scene();
    if (!s) return TRC_ERR_OOM;

    const bool with_spheres = (kind == TRC_SCENE_CORNELL_SPHERES);
    prepare_cube_list(*s);                     // materials 0-2
    prepare_cornell_box(*s);                   // materials 3-6
    prepare_sphere_list(*s, with_spheres);     // materials 7-18
    trc_Material test = make_material(TRC_MAT_GLASS);   // material 19
    test.medium = TRC_MEDIUM_HOMOGENEOUS;
    test.textureInfo.albedo = f3(1, 1, 1);
    test.specular = 1;
    test.eta = 1.5f;
    s->materials.push_back(test);

    std::vector<trc_BVH> leaves;
    trc_BVH leaf;
    // all cubes but the last (density container), AAPLRenderer.mm:459-462; kind VOLUME puts the container in too
    const uint32_t n_cube_leaves = (uint32_t)s->cubes.size() - (kind == TRC_SCENE_CORNELL_VOLUME ? 0u : 1u);
    for (uint32_t i = 0; i < n_cube_leaves; ++i) {
        trc_host_build_node(&s->cubes[i].box, &s->cubes[i].model_matrix, TRC_PRIM_CUBE, i, &leaf);
        leaves.push_back(leaf);
    }
    for (uint32_t i = 0; i < s->squares.size(); ++i) {
        trc_host_build_node(&s->squares[i].boundingBOX, &s->squares[i].model_matrix, TRC_PRIM_SQUARE, i, &leaf);
        leaves.push_back(leaf);
    }
    if (with_spheres) {
        // the commented-out loop of AAPLRenderer.mm:454-457, from i = 0 (BASELINE config 2)
        for (uint32_t i = 0; i < s->spheres.size(); ++i) {
            trc_host_build_node(&s->spheres[i].boundingBOX, &s->spheres[i].model_matrix, TRC_PRIM_SPHERE, i, &leaf);
            leaves.push_back(leaf);
        }
    }
    if (with_mesh) {
        s->vertices.assign(mesh_vertices, mesh_vertices + n_vertices);
        s->indices.assign(mesh_indices, mesh_indices + n_indices);
        for (uint32_t i = 0; i < n_indices; ++i)
            if (mesh_indices[i] >= n_vertices) { delete s; return TRC_ERR_INVALID_ARG; }

        trc_AABB mesh_box = empty_box();
        for (auto& e : s->vertices) mesh_box = box_grow(mesh_box, f3(e.v[0], e.v[1], e.v[2]));
        const trc_float3 centroid = box_centroid(mesh_box);
        const float max_dim = get(box_diagonal(mesh_box), box_max_extent(mesh_box));
        const double mesh_scale = 300.0 / max_dim;
        trc_float3 mesh_offset = f3(278.0f) - centroid;
        mesh_offset.y = (float)(180 - mesh_box.mini.y * mesh_scale);
        // each vertex transformed ONCE (the reference transforms per referencing triangle, B-14)
        for (auto& e : s->vertices) {
            e.v[0] = (float)(e.v[0] * mesh_scale);
            e.v[1] = (float)(e.v[1] * mesh_scale);
            e.v[2] = (float)(e.v[2] * -mesh_scale);
            e.n[2] *= -1;
            e.v[0] += mesh_offset.x;
            e.v[1] += mesh_offset.y;
            e.v[2] += mesh_offset.z;
            e.v[0] += -200;
        }
        const trc_float4x4 ident = identity4x4();
        const uint32_t n_tri = analytic_leaves_only ? 0u : n_indices / 3;      // else: trc_upload_scene_device writes them on the GPU
        leaves.reserve(leaves.size() + n_tri);
        for (uint32_t t = 0; t < n_tri; ++t) {
            const trc_TriangleVertex& a = s->vertices[s->indices[3 * t]];
            const trc_TriangleVertex& b = s->vertices[s->indices[3 * t + 1]];
            const trc_TriangleVertex& c = s->vertices[s->indices[3 * t + 2]];
            trc_AABB box;
            box.maxi = f3(std::max({a.v[0], b.v[0], c.v[0]}), std::max({a.v[1], b.v[1], c.v[1]}),
                          std::max({a.v[2], b.v[2], c.v[2]}));
            box.mini = f3(std::min({a.v[0], b.v[0], c.v[0]}), std::min({a.v[1], b.v[1], c.v[1]}),
                          std::min({a.v[2], b.v[2], c.v[2]}));
            trc_host_build_node(&box, &ident, TRC_PRIM_TRIANGLE, t, &leaf);
            leaves.push_back(leaf);
        }
    }

    if (analytic_leaves_only) {                // no tree either: bvhList = the analytic primitives' leaf records
        s->bvh = leaves;
        *out = s;
        return TRC_OK;
    }
    const uint32_t n_leaves = (uint32_t)leaves.size();
    s->bvh.resize(2 * (size_t)n_leaves - 1);
    std::copy(leaves.begin(), leaves.end(), s->bvh.begin());
    uint32_t n_nodes = 0;
    trc_status st = trc_host_build_tree(s->bvh.data(), n_leaves, &n_nodes);
    if (st != TRC_OK) { delete s; return st; }
    *out = s;
    return TRC_OK;
}

void trc_host_scene_destroy(trc_host_scene* s) { delete s; }

void trc_host_scene_view(const trc_host_scene* s, trc_scene* out) {
    out->bvhList = s->bvh.data();           out->n_bvh = (uint32_t)s->bvh.size();
    out->sphereList = s->spheres.data();    out->n_sphere = (uint32_t)s->spheres.size();
    out->squareList = s->squares.data();    out->n_square = (uint32_t)s->squares.size();
    out->cubeList = s->cubes.data();        out->n_cube = (uint32_t)s->cubes.size();
    out->triList = s->vertices.data();      out->n_vertex = (uint32_t)s->vertices.size();
    out->idxList = s->indices.data();       out->n_index = (uint32_t)s->indices.size();
    out->materials = s->materials.data();   out->n_material = (uint32_t)s->materials.size();
}


// GridDensityInfo::GridDensityInfo, Medium.hh:92-105
void trc_host_make_density_info(float sigma_a, float sigma_s, float g, uint32_t nx, uint32_t ny, uint32_t nz,
                                const float* density, trc_GridDensityInfo* out) {
    out->sigma_a = sigma_a; out->sigma_s = sigma_s; out->g = g;
    out->nx = nx; out->ny = ny; out->nz = nz;
    out->sigma_t = sigma_a + sigma_s;
    float maxDensity = 0;
    for (size_t i = 0; i < (size_t)nx * ny * nz; ++i) maxDensity = std::fmax(maxDensity, density[i]);
    out->invMaxDensity = 1 / maxDensity;
}

// procedural stand-in for the reference's cloud grid: a handful of Gaussian blobs, deterministic in `seed`
void trc_host_make_cloud(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t seed, float* out) {
    uint64_t state = 0, inc = ((uint64_t)seed << 1u) | 1u;
    pcg32_next(state, inc); state += 0x853c49e6748fea9bULL; pcg32_next(state, inc);
    auto rnd = [&]() { return (float)(pcg32_next(state, inc) >> 8) * (1.0f / 16777216.0f); };
    const int n_blobs = 9;
    float cx[n_blobs], cy[n_blobs], cz[n_blobs], rad[n_blobs], amp[n_blobs];
    for (int b = 0; b < n_blobs; ++b) {
        cx[b] = 0.2f + 0.6f * rnd(); cy[b] = 0.2f + 0.6f * rnd(); cz[b] = 0.25f + 0.5f * rnd();
        rad[b] = 0.08f + 0.14f * rnd(); amp[b] = 0.4f + 0.6f * rnd();
    }
    for (uint32_t z = 0; z < nz; ++z)
        for (uint32_t y = 0; y < ny; ++y)
            for (uint32_t x = 0; x < nx; ++x) {
                const float px = (x + 0.5f) / nx, py = (y + 0.5f) / ny, pz = (z + 0.5f) / nz;
                float d = 0;
                for (int b = 0; b < n_blobs; ++b) {
                    const float dx = px - cx[b], dy = py - cy[b], dz = pz - cz[b];
                    d += amp[b] * std::exp(-(dx * dx + dy * dy + dz * dz) / (rad[b] * rad[b]));
                }
                out[((size_t)z * ny + y) * nx + x] = d < 0.05f ? 0.0f : d;      // empty space stays exactly empty
            }
}

// `MakeNamedMedium "..." ... "integer nx" N "integer ny" N "integer nz" N ... "float density" [ v v v ... ]`
trc_status trc_host_load_density_pbrt(const char* path, uint32_t* nx, uint32_t* ny, uint32_t* nz, float** out) {
    if (!path || !nx || !ny || !nz || !out) return TRC_ERR_INVALID_ARG;
    *out = nullptr;
    std::string text;
    if (!read_pbrt_text(path, 0, text)) return TRC_ERR_INVALID_ARG;
    auto int_param = [&](const char* name, uint32_t* v) {
        const std::string key = std::string("\"integer ") + name + "\"";
        size_t p = text.find(key);
        if (p == std::string::npos) return false;
        p += key.size();
        while (p < text.size() && (std::isspace((unsigned char)text[p]) || text[p] == '[')) ++p;
        char* end = nullptr;
        const long val = std::strtol(text.c_str() + p, &end, 10);
        if (end == text.c_str() + p || val <= 0) return false;
        *v = (uint32_t)val;
        return true;
    };
    if (!int_param("nx", nx) || !int_param("ny", ny) || !int_param("nz", nz)) return TRC_ERR_INVALID_ARG;
    size_t p = text.find("\"float density\"");
    if (p == std::string::npos) return TRC_ERR_INVALID_ARG;
    p = text.find('[', p);
    if (p == std::string::npos) return TRC_ERR_INVALID_ARG;
    if (*nx > 65536u || *ny > 65536u || *nz > 65536u || (uint64_t)*nx * *ny > (1ull << 31) / *nz) return TRC_ERR_INVALID_ARG;
    const size_t count = (size_t)*nx * *ny * *nz;
    float* data = (float*)std::malloc(count * sizeof(float));
    if (!data) return TRC_ERR_OOM;
    const char* cur = text.c_str() + p + 1;
    for (size_t i = 0; i < count; ++i) {
        char* end = nullptr;
        data[i] = std::strtof(cur, &end);
        if (end == cur) { std::free(data); return TRC_ERR_INVALID_ARG; }
        cur = end;
    }
    *out = data;
    return TRC_OK;
}
void trc_host_free(void* p) { std::free(p); }

// Radiance RGBE (.hdr) -> float RGB.  The reference's backdrop is such a file (vulture_hide_4k.hdr through NSImage +
// MTKTextureLoader with MTKTextureLoaderOriginFlippedVertically, AAPLRenderer.mm:352-383), sampled by direction through
// SampleSphericalMap (Render.hh:42-48).  Published format (G. Ward, Graphics Gems II "Real Pixels"): text header up to an
// empty line (needs FORMAT=32-bit_rle_rgbe), a resolution line "-Y H +X W", then H scanlines, each either flat (4 bytes
// per pixel; old run-length markers 1 1 1 n are honoured) or new-style RLE (2 2 hi lo, then the four channels separately:
// count > 128 -> a run of count - 128 copies of the next byte, else `count` literal bytes).  A pixel is
// mantissa * 2^(e - 136), e = 0 -> black.  Rows come back BOTTOM-UP -- row 0 is the last scanline of the file -- which is
// both what the reference's flipped texture holds and what trc_set_environment_map expects (v grows with the direction's y).
trc_status trc_host_load_hdr(const char* path, uint32_t* width, uint32_t* height, float** rgb) {
    if (!path || !width || !height || !rgb) return TRC_ERR_INVALID_ARG;
    *rgb = nullptr; *width = *height = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) return TRC_ERR_INVALID_ARG;
    std::vector<unsigned char> d;
    {
        unsigned char chunk[1 << 16];
        size_t got;
        while ((got = std::fread(chunk, 1, sizeof chunk, f)) > 0) d.insert(d.end(), chunk, chunk + got);
        std::fclose(f);
    }
    size_t pos = 0;
    auto line = [&](std::string& l) {
        l.clear();
        if (pos >= d.size()) return false;
        while (pos < d.size() && d[pos] != '\n') l.push_back((char)d[pos++]);
        if (pos < d.size()) ++pos;
        if (!l.empty() && l.back() == '\r') l.pop_back();
        return true;
    };
    std::string l;
    if (!line(l) || (l.rfind("#?RADIANCE", 0) != 0 && l.rfind("#?RGBE", 0) != 0)) return TRC_ERR_INVALID_ARG;
    bool format_ok = false;
    for (;;) {
        if (!line(l)) return TRC_ERR_INVALID_ARG;
        if (l.empty()) break;
        if (l.rfind("FORMAT=", 0) == 0) format_ok = l == "FORMAT=32-bit_rle_rgbe";
    }
    if (!format_ok || !line(l)) return TRC_ERR_UNSUPPORTED;
    long H = 0, W = 0;
    char sy = 0, sx = 0;
    if (std::sscanf(l.c_str(), "%cY %ld %cX %ld", &sy, &H, &sx, &W) != 4 || sy != '-' || sx != '+') return TRC_ERR_UNSUPPORTED;   // the standard orientation only
    if (W <= 0 || H <= 0 || (unsigned long long)W * (unsigned long long)H > (1ull << 28)) return TRC_ERR_INVALID_ARG;
    float* out = (float*)std::malloc((size_t)W * H * 3 * sizeof(float));
    if (!out) return TRC_ERR_OOM;
    std::vector<unsigned char> scan((size_t)W * 4);
    auto bad = [&]() { std::free(out); return TRC_ERR_INVALID_ARG; };
    for (long y = 0; y < H; ++y) {
        if (pos + 4 > d.size()) return bad();
        if (W >= 8 && W < 32768 && d[pos] == 2 && d[pos + 1] == 2 && !(d[pos + 2] & 0x80)) {        // new-style RLE scanline
            if (((long)d[pos + 2] << 8 | d[pos + 3]) != W) return bad();
            pos += 4;
            for (int c = 0; c < 4; ++c) {
                long x = 0;
                while (x < W) {
                    if (pos >= d.size()) return bad();
                    unsigned count = d[pos++];
                    if (count > 128) {
                        count -= 128;
                        if (count == 0 || x + (long)count > W || pos >= d.size()) return bad();
                        const unsigned char v = d[pos++];
                        for (unsigned k = 0; k < count; ++k) scan[(size_t)(x++) * 4 + c] = v;
                    } else {
                        if (count == 0 || x + (long)count > W || pos + count > d.size()) return bad();
                        for (unsigned k = 0; k < count; ++k) scan[(size_t)(x++) * 4 + c] = d[pos++];
                    }
                }
            }
        } else {                                                                                     // flat, with old-style runs
            long x = 0;
            int shift = 0;
            while (x < W) {
                if (pos + 4 > d.size()) return bad();
                const unsigned char* px = &d[pos]; pos += 4;
                if (px[0] == 1 && px[1] == 1 && px[2] == 1 && x > 0) {
                    const long n = (long)px[3] << shift;
                    if (n <= 0 || x + n > W) return bad();
                    for (long k = 0; k < n; ++k, ++x) std::memcpy(&scan[(size_t)x * 4], &scan[(size_t)(x - 1) * 4], 4);
                    shift += 8;
                    if (shift > 24) return bad();
                } else { std::memcpy(&scan[(size_t)x * 4], px, 4); ++x; shift = 0; }
            }
        }
        float* row = out + (size_t)(H - 1 - y) * W * 3;                                             // bottom-up
        for (long x = 0; x < W; ++x) {
            const unsigned char* px = &scan[(size_t)x * 4];
            const float scale = px[3] ? std::ldexp(1.0f, (int)px[3] - 136) : 0.0f;
            row[3 * x] = (float)px[0] * scale; row[3 * x + 1] = (float)px[1] * scale; row[3 * x + 2] = (float)px[2] * scale;
        }
    }
    *width = (uint32_t)W; *height = (uint32_t)H; *rgb = out;
    return TRC_OK;
}

// PNG with stored deflate blocks: signature, IHDR, one IDAT (zlib stream of filter-0 scanlines), IEND
trc_status trc_host_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || width == 0 || height == 0) return TRC_ERR_INVALID_ARG;
    struct CrcTable {                    // built once, by whichever thread gets here first (C++11 static initialisation)
        uint32_t t[256];
        CrcTable() {
            for (uint32_t n = 0; n < 256; ++n) {
                uint32_t c = n;
                for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
                t[n] = c;
            }
        }
    };
    static const CrcTable crc;
    const uint32_t* crc_table = crc.t;
    auto be32 = [](std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); };
    auto chunk = [&](std::vector<uint8_t>& file, const char type[4], const std::vector<uint8_t>& data) {
        be32(file, (uint32_t)data.size());
        const size_t start = file.size();
        file.insert(file.end(), type, type + 4);
        file.insert(file.end(), data.begin(), data.end());
        uint32_t c = 0xFFFFFFFFu;
        for (size_t i = start; i < file.size(); ++i) c = crc_table[(c ^ file[i]) & 0xFFu] ^ (c >> 8);
        be32(file, c ^ 0xFFFFFFFFu);
    };
    // raw scanlines with filter byte 0
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw;
    raw.reserve((stride + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba8 + y * stride, rgba8 + (y + 1) * stride);
    }
    std::vector<uint8_t> z;
    z.reserve(raw.size() + raw.size() / 65535 * 5 + 16);
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;                                   // adler32
    for (size_t pos = 0; pos < raw.size();) {
        const size_t len = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + len == raw.size() ? 1 : 0);
        z.push_back(len & 0xFF); z.push_back(len >> 8); z.push_back(~len & 0xFF); z.push_back((~len >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + len);
        for (size_t i = pos; i < pos + len; ++i) { a = (a + raw[i]) % 65521u; b = (b + a) % 65521u; }
        pos += len;
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, width); be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(6); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGBA
    chunk(file, "IHDR", ihdr);
    chunk(file, "IDAT", z);
    chunk(file, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return TRC_ERR_INVALID_ARG;
    const bool ok = std::fwrite(file.data(), 1, file.size(), f) == file.size();
    std::fclose(f);
    return ok ? TRC_OK : TRC_ERR_INVALID_ARG;
}

// C-ABI view of include/trc_sobol.h (tables of pbrt::SobolSampler, SobolSampler.hh:126-160) for hosts and tests; the
// device library builds the same tables itself
void trc_host_sobol_matrices32(uint32_t* out) { trc_sobol_matrices32(out); }

trc_status trc_host_sobol_interval_tables(uint32_t log2res, uint64_t* vdc, uint64_t* inv) {
    if (!vdc || !inv) return TRC_ERR_INVALID_ARG;
    return trc_sobol_interval_tables(log2res, vdc, inv) == 0 ? TRC_OK : TRC_ERR_INVALID_ARG;
}

}  // extern "C"

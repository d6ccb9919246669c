// ref_minipbrt_shim.cpp -- C entry point over the REFERENCE's own pbrt-v3 parser (RT_Metal/Tracer/minipbrt.{h,cpp},
// vendored there from vilya/minipbrt), compiled together with minipbrt.cpp from where it lies into
// oracle/_ref/libminipbrt_ref.so (oracle/Makefile).  TEST INFRASTRUCTURE: it is the yardstick for
// trc_host_load_density_pbrt; nothing of minipbrt is copied into this repository.
//
// Does what AAPLRenderer.mm:629-636 does: Loader::load, take_scene, mediums[0] as HeterogeneousMedium.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "minipbrt.h"

extern "C" int ref_minipbrt_load_density(const char* path, int* nx, int* ny, int* nz, float** out) {
    minipbrt::Loader loader;
    if (!loader.load(path)) return -1;
    minipbrt::Scene* scene = loader.take_scene();
    if (!scene) return -2;
    int rc = -3;
    if (!scene->mediums.empty()) {
        auto* medium = dynamic_cast<minipbrt::HeterogeneousMedium*>(scene->mediums[0]);
        if (medium && medium->density && medium->nx > 0 && medium->ny > 0 && medium->nz > 0) {
            const size_t n = (size_t)medium->nx * medium->ny * medium->nz;
            *nx = medium->nx; *ny = medium->ny; *nz = medium->nz;
            *out = (float*)std::malloc(n * sizeof(float));
            if (*out) { std::memcpy(*out, medium->density, n * sizeof(float)); rc = 0; }
        }
    }
    delete scene;
    return rc;
}

// every TriangleMesh of the scene in file order: raw P / N / uv as parsed, its indices rebased onto the concatenated
// vertex array, and each vertex's shapeToWorld matrix (16 floats, row-major) -- the test applies it itself.
extern "C" int ref_minipbrt_triangle_meshes(const char* path, unsigned* n_vertices, unsigned* n_indices, float** P, float** N,
                                            float** uv, float** matrices, unsigned** indices) {
    minipbrt::Loader loader;
    if (!loader.load(path)) return -1;
    minipbrt::Scene* scene = loader.take_scene();
    if (!scene) return -2;
    // shapes of object definitions (ObjectBegin .. ObjectEnd) are templates, not part of the world
    std::vector<bool> in_object(scene->shapes.size(), false);
    for (minipbrt::Object* o : scene->objects)
        if (o && o->firstShape != minipbrt::kInvalidIndex)
            for (unsigned k = 0; k < o->numShapes; ++k) in_object[o->firstShape + k] = true;
    std::vector<minipbrt::Shape*> world;
    for (size_t k = 0; k < scene->shapes.size(); ++k) if (!in_object[k]) world.push_back(scene->shapes[k]);
    size_t nv = 0, ni = 0;
    for (minipbrt::Shape* s : world)
        if (s->type() == minipbrt::ShapeType::TriangleMesh) {
            auto* m = static_cast<minipbrt::TriangleMesh*>(s);
            nv += m->num_vertices; ni += m->num_indices;
        }
    *n_vertices = (unsigned)nv; *n_indices = (unsigned)ni;
    *P = (float*)std::calloc(nv * 3 + 1, sizeof(float)); *N = (float*)std::calloc(nv * 3 + 1, sizeof(float));
    *uv = (float*)std::calloc(nv * 2 + 1, sizeof(float)); *matrices = (float*)std::calloc(nv * 16 + 1, sizeof(float));
    *indices = (unsigned*)std::calloc(ni + 1, sizeof(unsigned));
    size_t v0 = 0, i0 = 0;
    for (minipbrt::Shape* s : world) {
        if (s->type() != minipbrt::ShapeType::TriangleMesh) continue;
        auto* m = static_cast<minipbrt::TriangleMesh*>(s);
        for (unsigned v = 0; v < m->num_vertices; ++v) {
            for (int k = 0; k < 3; ++k) {
                (*P)[(v0 + v) * 3 + k] = m->P[v * 3 + k];
                (*N)[(v0 + v) * 3 + k] = m->N ? m->N[v * 3 + k] : 0.0f;
            }
            for (int k = 0; k < 2; ++k) (*uv)[(v0 + v) * 2 + k] = m->uv ? m->uv[v * 2 + k] : 0.0f;
            std::memcpy(*matrices + (v0 + v) * 16, &m->shapeToWorld.start[0][0], 16 * sizeof(float));
        }
        for (unsigned k = 0; k < m->num_indices; ++k) (*indices)[i0 + k] = (unsigned)(v0 + (unsigned)m->indices[k]);
        v0 += m->num_vertices; i0 += m->num_indices;
    }
    delete scene;
    return 0;
}

extern "C" void ref_minipbrt_free(void* p) { std::free(p); }

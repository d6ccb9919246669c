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

// What the REFERENCE's parser makes of a whole scene file, flattened for tests/test_pbrt_scene.py (the yardstick of
// trc_host_scene_load_pbrt): camera (cameraToWorld, fov, lensradius, focaldistance), film resolution, and per world
// shape in file order: type, shapeToWorld (row-major), sphere radius, mesh sizes, material type + its colour
// (Kd / Kr / Kt), area light L.  `shapes` holds 32 floats per shape:
//   [0] type (0 sphere, 3 trianglemesh, -1 other)  [1..16] shapeToWorld  [17] radius  [18] n_vertices  [19] n_indices
//   [20] material (0 matte 1 plastic 2 metal 3 mirror 4 glass 5 other, -1 none)  [21..23] colour
//   [24] has area light  [25..27] L * scale
extern "C" int ref_minipbrt_describe(const char* path, float camera[20], int film[2], float** shapes, unsigned* n_shapes) {
    minipbrt::Loader loader;
    if (!loader.load(path)) return -1;
    minipbrt::Scene* scene = loader.take_scene();
    if (!scene) return -2;
    std::memset(camera, 0, 20 * sizeof(float));
    if (scene->camera) {
        std::memcpy(camera, &scene->camera->cameraToWorld.start[0][0], 16 * sizeof(float));
        if (scene->camera->type() == minipbrt::CameraType::Perspective) {
            auto* pc = static_cast<minipbrt::PerspectiveCamera*>(scene->camera);
            camera[16] = pc->fov; camera[17] = pc->lensradius; camera[18] = pc->focaldistance; camera[19] = 1.0f;
        }
    }
    film[0] = film[1] = 0;
    if (scene->film) scene->film->get_resolution(film[0], film[1]);
    std::vector<bool> in_object(scene->shapes.size(), false);
    for (minipbrt::Object* o : scene->objects)
        if (o && o->firstShape != minipbrt::kInvalidIndex)
            for (unsigned k = 0; k < o->numShapes; ++k) in_object[o->firstShape + k] = true;
    std::vector<float> out;
    unsigned n = 0;
    for (size_t k = 0; k < scene->shapes.size(); ++k) {
        if (in_object[k]) continue;
        minipbrt::Shape* s = scene->shapes[k];
        float rec[32] = {0};
        rec[0] = -1.0f;
        std::memcpy(rec + 1, &s->shapeToWorld.start[0][0], 16 * sizeof(float));
        if (s->type() == minipbrt::ShapeType::Sphere) { rec[0] = 0.0f; rec[17] = static_cast<minipbrt::Sphere*>(s)->radius; }
        else if (s->type() == minipbrt::ShapeType::TriangleMesh) {
            auto* m = static_cast<minipbrt::TriangleMesh*>(s);
            rec[0] = 3.0f; rec[18] = (float)m->num_vertices; rec[19] = (float)m->num_indices;
        }
        rec[20] = -1.0f;
        if (s->material != minipbrt::kInvalidIndex && s->material < scene->materials.size()) {
            minipbrt::Material* m = scene->materials[s->material];
            const float* c = nullptr;
            const float one[3] = {1, 1, 1};
            switch (m->type()) {
                case minipbrt::MaterialType::Matte: rec[20] = 0; c = static_cast<minipbrt::MatteMaterial*>(m)->Kd.value; break;
                case minipbrt::MaterialType::Plastic: rec[20] = 1; c = static_cast<minipbrt::PlasticMaterial*>(m)->Kd.value; break;
                case minipbrt::MaterialType::Metal: rec[20] = 2; c = one; break;
                case minipbrt::MaterialType::Mirror: rec[20] = 3; c = static_cast<minipbrt::MirrorMaterial*>(m)->Kr.value; break;
                case minipbrt::MaterialType::Glass: rec[20] = 4; c = static_cast<minipbrt::GlassMaterial*>(m)->Kt.value; break;
                default: rec[20] = 5; c = nullptr; break;
            }
            if (c) { rec[21] = c[0]; rec[22] = c[1]; rec[23] = c[2]; }
        }
        if (s->areaLight != minipbrt::kInvalidIndex && s->areaLight < scene->areaLights.size()) {
            minipbrt::AreaLight* al = scene->areaLights[s->areaLight];
            rec[24] = 1.0f;
            if (al->type() == minipbrt::AreaLightType::Diffuse) {
                auto* dl = static_cast<minipbrt::DiffuseAreaLight*>(al);
                for (int j = 0; j < 3; ++j) rec[25 + j] = dl->L[j] * dl->scale[j];
            }
        }
        out.insert(out.end(), rec, rec + 32);
        ++n;
    }
    *n_shapes = n;
    *shapes = (float*)std::calloc(out.size() + 1, sizeof(float));
    if (!out.empty()) std::memcpy(*shapes, out.data(), out.size() * sizeof(float));
    delete scene;
    return 0;
}

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
    scene->load_all_ply_meshes();          // `Shape "plymesh"` -> TriangleMesh in place, through minipbrt's own PLY reader
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

// one shape of the parsed scene as the 48 floats documented at ref_minipbrt_describe
static void describe_shape(minipbrt::Scene* scene, minipbrt::Shape* s, float rec[48]) {
    std::memset(rec, 0, 48 * sizeof(float));
    rec[0] = -1.0f;
    std::memcpy(rec + 1, &s->shapeToWorld.start[0][0], 16 * sizeof(float));
    if (s->type() == minipbrt::ShapeType::Sphere) { rec[0] = 0.0f; rec[17] = static_cast<minipbrt::Sphere*>(s)->radius; }
    else if (s->type() == minipbrt::ShapeType::TriangleMesh) {
        auto* m = static_cast<minipbrt::TriangleMesh*>(s);
        rec[0] = 3.0f; rec[18] = (float)m->num_vertices; rec[19] = (float)m->num_indices;
    } else if (s->type() == minipbrt::ShapeType::Disk) {
        auto* d = static_cast<minipbrt::Disk*>(s);
        rec[0] = 6.0f; rec[17] = d->radius; rec[35] = rec[36] = d->height; rec[37] = d->innerradius; rec[38] = d->phimax;
    } else if (s->type() == minipbrt::ShapeType::Cylinder) {
        auto* c = static_cast<minipbrt::Cylinder*>(s);
        rec[0] = 7.0f; rec[17] = c->radius; rec[35] = c->zmin; rec[36] = c->zmax; rec[38] = c->phimax;
    } else if (s->type() == minipbrt::ShapeType::Cone) {
        auto* c = static_cast<minipbrt::Cone*>(s);
        rec[0] = 9.0f; rec[17] = c->radius; rec[35] = 0.0f; rec[36] = c->height; rec[38] = c->phimax;
    } else if (s->type() == minipbrt::ShapeType::Paraboloid) {
        auto* c = static_cast<minipbrt::Paraboloid*>(s);
        rec[0] = 10.0f; rec[17] = c->radius; rec[35] = c->zmin; rec[36] = c->zmax; rec[38] = c->phimax;
    } else if (s->type() == minipbrt::ShapeType::Hyperboloid) {
        auto* h = static_cast<minipbrt::Hyperboloid*>(s);
        rec[0] = 11.0f; rec[38] = h->phimax;
        for (int k = 0; k < 3; ++k) { rec[39 + k] = h->p1[k]; rec[42 + k] = h->p2[k]; }
    } else if (s->type() == minipbrt::ShapeType::PLYMesh) {
        rec[0] = 8.0f;
        if (minipbrt::TriangleMesh* m = s->triangle_mesh()) {        // reads the PLY file (minipbrt.cpp:4380-4450)
            rec[18] = (float)m->num_vertices; rec[19] = (float)m->num_indices;
            delete m;
        }
    }
    rec[20] = -1.0f;
    if (s->material != minipbrt::kInvalidIndex && s->material < scene->materials.size()) {
        minipbrt::Material* m = scene->materials[s->material];
        const float* c = nullptr;
        const float one[3] = {1, 1, 1};
        uint32_t tex = minipbrt::kInvalidIndex;
        switch (m->type()) {
            case minipbrt::MaterialType::Matte: { auto& k = static_cast<minipbrt::MatteMaterial*>(m)->Kd; rec[20] = 0; c = k.value; tex = k.texture; break; }
            case minipbrt::MaterialType::Plastic: { auto& k = static_cast<minipbrt::PlasticMaterial*>(m)->Kd; rec[20] = 1; c = k.value; tex = k.texture; break; }
            case minipbrt::MaterialType::Metal: rec[20] = 2; c = one; break;
            case minipbrt::MaterialType::Mirror: { auto& k = static_cast<minipbrt::MirrorMaterial*>(m)->Kr; rec[20] = 3; c = k.value; tex = k.texture; break; }
            case minipbrt::MaterialType::Glass: { auto& k = static_cast<minipbrt::GlassMaterial*>(m)->Kt; rec[20] = 4; c = k.value; tex = k.texture; break; }
            default: rec[20] = 5; c = nullptr; break;
        }
        if (c) { rec[21] = c[0]; rec[22] = c[1]; rec[23] = c[2]; }
        if (tex != minipbrt::kInvalidIndex && tex < scene->textures.size()) {
            minipbrt::Texture* t = scene->textures[tex];
            rec[28] = 2.0f;
            if (t->type() == minipbrt::TextureType::Checkerboard2D) {
                auto* cb = static_cast<minipbrt::Checkerboard2DTexture*>(t);
                rec[28] = 1.0f;
                for (int j = 0; j < 3; ++j) { rec[29 + j] = cb->tex1.value[j]; rec[32 + j] = cb->tex2.value[j]; }
            }
        }
    }
    if (s->areaLight != minipbrt::kInvalidIndex && s->areaLight < scene->areaLights.size()) {
        minipbrt::AreaLight* al = scene->areaLights[s->areaLight];
        rec[24] = 1.0f;
        if (al->type() == minipbrt::AreaLightType::Diffuse) {
            auto* dl = static_cast<minipbrt::DiffuseAreaLight*>(al);
            for (int j = 0; j < 3; ++j) rec[25 + j] = dl->L[j] * dl->scale[j];
        }
    }
}

// What the REFERENCE's parser makes of a whole scene file, flattened for tests/test_pbrt_scene.py (the yardstick of
// trc_host_scene_load_pbrt): camera (cameraToWorld, fov, lensradius, focaldistance), film resolution, and per world
// shape in file order: type, shapeToWorld (row-major), sphere radius, mesh sizes, material type + its colour
// (Kd / Kr / Kt), area light L.  `shapes` holds 48 floats per shape:
//   [0] type (0 sphere, 3 trianglemesh, 6 disk, 7 cylinder, 8 plymesh, 9 cone, 10 paraboloid, 11 hyperboloid, -1 other)  [1..16] shapeToWorld
//   [17] radius (sphere, disk, cylinder)  [18] n_vertices  [19] n_indices (trianglemesh; plymesh: of the PLY file as
//   minipbrt's PLYMesh::triangle_mesh() loads it)
//   [20] material (0 matte 1 plastic 2 metal 3 mirror 4 glass 5 other, -1 none)  [21..23] colour
//   [24] has area light  [25..27] L * scale
//   [28] what the colour parameter names: 0 a constant, 1 a 2-D checkerboard texture, 2 another texture
//   [29..31] / [32..34] the checkerboard's tex1 / tex2 values
//   [35] zmin (cylinder, paraboloid) / height (disk) / 0 (cone)  [36] zmax / height  [37] innerradius (disk)  [38] phimax (quadrics)
//   [39..41] p1  [42..44] p2 (hyperboloid)
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
        float rec[48];
        describe_shape(scene, scene->shapes[k], rec);
        out.insert(out.end(), rec, rec + 48);
        ++n;
    }
    *n_shapes = n;
    *shapes = (float*)std::calloc(out.size() + 1, sizeof(float));
    if (!out.empty()) std::memcpy(*shapes, out.data(), out.size() * sizeof(float));
    delete scene;
    return 0;
}

// The Texture directives of a scene file in file order, 8 floats each: [0] class (1 = 2-D checkerboard, 2 = anything else)
// [1..3] tex1  [4..6] tex2  [7] 1 = spectrum data.  (The vendored minipbrt parses them but never resolves a material's
// "texture Kd" reference to them -- find_texture walks per-attribute lists that nothing appends to, minipbrt.cpp:8145-8172,
// 2912-2913 -- so ref_minipbrt_describe reports [28] = 0 for every shape; tests compare the declarations instead.)
extern "C" int ref_minipbrt_textures(const char* path, float** out, unsigned* n_textures) {
    minipbrt::Loader loader;
    if (!loader.load(path)) return -1;
    minipbrt::Scene* scene = loader.take_scene();
    if (!scene) return -2;
    const size_t n = scene->textures.size();
    *n_textures = (unsigned)n;
    *out = (float*)std::calloc(n * 8 + 1, sizeof(float));
    for (size_t k = 0; k < n; ++k) {
        minipbrt::Texture* t = scene->textures[k];
        float* r = *out + k * 8;
        r[0] = 2.0f;
        r[7] = t->dataType == minipbrt::TextureData::Spectrum ? 1.0f : 0.0f;
        if (t->type() == minipbrt::TextureType::Checkerboard2D) {
            auto* cb = static_cast<minipbrt::Checkerboard2DTexture*>(t);
            r[0] = 1.0f;
            for (int j = 0; j < 3; ++j) { r[1 + j] = cb->tex1.value[j]; r[4 + j] = cb->tex2.value[j]; }
        }
    }
    delete scene;
    return 0;
}

// The ObjectInstance directives of a scene file in file order, one record of 64 floats per (instance, shape of its object):
// [0..47] the template's shape as ref_minipbrt_describe writes a shape (its own shapeToWorld: the CTM at its Shape directive),
// [48..63] the instance's instanceToWorld (row-major).  pbrt-v3 places the copy at instanceToWorld x shapeToWorld (api.cpp
// pbrtObjectInstance: TransformedPrimitive); the test multiplies.
extern "C" int ref_minipbrt_instances(const char* path, float** out, unsigned* n_records) {
    minipbrt::Loader loader;
    if (!loader.load(path)) return -1;
    minipbrt::Scene* scene = loader.take_scene();
    if (!scene) return -2;
    std::vector<float> recs;
    unsigned n = 0;
    for (minipbrt::Instance* in : scene->instances) {
        if (!in || in->object == minipbrt::kInvalidIndex || in->object >= scene->objects.size()) continue;
        minipbrt::Object* o = scene->objects[in->object];
        if (!o || o->firstShape == minipbrt::kInvalidIndex) continue;
        for (unsigned k = 0; k < o->numShapes; ++k) {
            float rec[64];
            describe_shape(scene, scene->shapes[o->firstShape + k], rec);
            std::memcpy(rec + 48, &in->instanceToWorld.start[0][0], 16 * sizeof(float));
            recs.insert(recs.end(), rec, rec + 64);
            ++n;
        }
    }
    *n_records = n;
    *out = (float*)std::calloc(recs.size() + 1, sizeof(float));
    if (!recs.empty()) std::memcpy(*out, recs.data(), recs.size() * sizeof(float));
    delete scene;
    return 0;
}

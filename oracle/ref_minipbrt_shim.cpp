// ref_minipbrt_shim.cpp -- C entry point over the REFERENCE's own pbrt-v3 parser (RT_Metal/Tracer/minipbrt.{h,cpp},
// vendored there from vilya/minipbrt), compiled together with minipbrt.cpp from where it lies into
// oracle/_ref/libminipbrt_ref.so (oracle/Makefile).  TEST INFRASTRUCTURE: it is the yardstick for
// trc_host_load_density_pbrt; nothing of minipbrt is copied into this repository.
//
// Does what AAPLRenderer.mm:629-636 does: Loader::load, take_scene, mediums[0] as HeterogeneousMedium.
#include <cstdlib>
#include <cstring>

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

extern "C" void ref_minipbrt_free(void* p) { std::free(p); }

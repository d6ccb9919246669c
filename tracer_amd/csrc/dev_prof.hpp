// dev_prof.hpp -- exact work counters and the divergence / cycle profile of the INSTRUMENTED kernels
// (template parameter STATS = true); in production kernels every use compiles to nothing.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_vec.hpp"

namespace trcdev {

// divergence profile sites (instrumented kernels only): how many LANES vs how many WAVEFRONTS executed a site
enum ProfSite { kProfLoop = 0, kProfDescend, kProfSquare, kProfSphere, kProfCube, kProfTriangle, kProfShade,
                kProfLambert, kProfMetal, kProfBeckSample, kProfBeckEval, kProfFinish, kProfCount };

struct TravCounters {   // only live in instrumented kernels
    uint32_t rays, shaded, n_descend, n_return, leaf[4], hit_triangle, hit_cube;
    uint32_t prof_lane[kProfCount], prof_wave[kProfCount];
    uint64_t prof_cycles[kProfCount];     // shader-clock cycles the wavefront spent inside the site (leader lane only)
};
TRC_DEV void counters_zero(TravCounters& c) {
    c.rays = c.shaded = c.n_descend = c.n_return = 0; c.leaf[0] = c.leaf[1] = c.leaf[2] = c.leaf[3] = 0;
    c.hit_triangle = c.hit_cube = 0;
    for (int i = 0; i < kProfCount; ++i) { c.prof_lane[i] = 0; c.prof_wave[i] = 0; c.prof_cycles[i] = 0; }
}
template <bool STATS>
TRC_DEV void prof(TravCounters& c, int site) {
    if (STATS) {
        c.prof_lane[site]++;
        const unsigned long long m = __ballot(1);
        if (__lane_id() == (unsigned)(__ffsll((long long)m) - 1)) c.prof_wave[site]++;
    }
}

// RAII site marker of the instrumented kernels: counts lanes / wavefronts like prof<> and adds the cycles between
// entry and the reconvergence point at the end of the scope (the scope body must not return / break out).
template <bool STATS>
struct ProfScope {
    TravCounters& c; int site; uint64_t t0; bool lead;
    TRC_DEV ProfScope(TravCounters& cnt, int s) : c(cnt), site(s), t0(0), lead(false) {
        if (STATS) {
            c.prof_lane[site]++;
            const unsigned long long m = __ballot(1);
            lead = __lane_id() == (unsigned)(__ffsll((long long)m) - 1);
            if (lead) { c.prof_wave[site]++; t0 = clock64(); }
        }
    }
    TRC_DEV ~ProfScope() { if (STATS && lead) c.prof_cycles[site] += (uint64_t)clock64() - t0; }
};

}  // namespace trcdev

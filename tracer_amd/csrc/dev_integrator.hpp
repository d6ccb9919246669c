// dev_integrator.hpp -- camera ray, tracePath / traceMIS and the per-pixel accumulation loop.
//
// Reference (RT_Metal/Metal/): Random.metal:3-26, RandomSampler.hh:6-46, Camera.hh:59-69,
// Render.metal:277-409 (traceMIS), :411-492 (tracePath), :495-558 (kernelPathTracing).
#pragma once

#include "dev_bsdf.hpp"
#include "dev_intersect.hpp"
#include "trc_sobol.h"

// traceVolume: steps of the GridDensity medium's delta tracker a lane takes per iteration of the render loop before the other lanes
// get their next Scene::hit (dev_integrator.hpp grid_sample); 0 = the whole tracker at once
#ifndef TRC_TRACK_SLICE
#define TRC_TRACK_SLICE 20
#endif

namespace trcdev {

// ---------------------------------------------------------------- PCG32 (Random.metal:3-26)
struct Pcg { uint64_t state, inc; };
TRC_DEV uint32_t pcg_next(Pcg& r) {
    uint64_t old = r.state;
    r.state = old * 6364136223846793005ULL + r.inc;
    uint32_t xorshifted = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
}
// randomF: ldexp(float(u32), -32); float(u32) rounds to nearest-even so 1.0f is reachable (B-4)
TRC_DEV float pcg_float(Pcg& r) { return ldexpf((float)pcg_next(r), -32); }

// Work counters of a lane (rays, shaded hits): a register, or -- in the kernels that park per-pixel state in LDS
// (trc_render_kernels.hpp: PARK) -- a word of the lane's LDS column bumped by one ds_add_u32, which costs no register at all.
struct LdsCount { uint32_t* p; };
TRC_DEV void bump(uint32_t& c) { c++; }
TRC_DEV void bump(LdsCount& c) { (void)__hip_atomic_fetch_add(c.p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }

struct Shade {            // what the integrators need from the material table
    const uint32_t* mats; // kMaterialDwords per material: type, texType, albedo.rgb
};
TRC_DEV int mat_type(const Shade& sh, uint32_t m) { return (int)sh.mats[m * kMaterialDwords]; }
TRC_DEV int mat_tex(const Shade& sh, uint32_t m) { return (int)sh.mats[m * kMaterialDwords + 1]; }
TRC_DEV bool mat_specular(const Shade& sh, uint32_t m) { return sh.mats[m * kMaterialDwords + 5] != 0u; }
TRC_DEV int mat_medium(const Shade& sh, uint32_t m) { return (int)sh.mats[m * kMaterialDwords + 6]; }
TRC_DEV F3 mat_albedo(const Shade& sh, uint32_t m) {
    const uint32_t* p = sh.mats + m * kMaterialDwords;
    return f3(__uint_as_float(p[2]), __uint_as_float(p[3]), __uint_as_float(p[4]));
}
// texture colour of a hit; the sphere's uv (atan2 + asin, Sphere.hh:19-31) is only materialised here,
// and only when a texture consumes it -- same value as computing it inside hit_test
// ... and so is a square's (two divisions of rec.p's in-plane coordinates, Square.hh; square_uv)
TRC_DEV F3 hit_color(const SceneRef& S, const Shade& sh, const HitRec& rec) {
    const int tex = mat_tex(sh, rec.material);
    F2 uv = rec.uv;
    if (tex == kTexChecker) {
        const uint32_t type = rec.tag >> kTagIndexBits;
        if (type == 0u) uv = sphere_uv(rec.gn);
        else if (type == 1u) uv = square_uv(S, rec.tag & kTagIndexMask, rec.p);
    }
    return texture_value(tex, mat_albedo(sh, rec.material), uv);
}

TRC_DEV float rgb_to_y(F3 rgb) { return 0.212671f * rgb.x + 0.715160f * rgb.y + 0.072169f * rgb.z; }   // Spectrum.hh:186-190

// Camera.hh:59-69 + RandomSampler.hh:40-46 (sampleUnitInDisk returns a point ON the unit circle, B-2)
TRC_DEV Ray cast_ray(const DCamera& cam, float s, float t, Pcg& rng) {
    float px, py;
    do {
        float a = pcg_float(rng);
        float b = pcg_float(rng);
        px = 2.0f * a - 1.0f;
        py = 2.0f * b - 1.0f;
    } while (px * px + py * py >= 1.0f);
    float inv = rsqrt_cr(px * px + py * py);
    float rdx = cam.lenRadius * (px * inv), rdy = cam.lenRadius * (py * inv);
    F3 offset = f3(cam.u[0], cam.u[1], cam.u[2]) * rdx + f3(cam.v[0], cam.v[1], cam.v[2]) * rdy;
    F3 origin = f3(cam.lookFrom[0], cam.lookFrom[1], cam.lookFrom[2]) + offset;
    F3 sample = f3(cam.cornerLowLeft[0], cam.cornerLowLeft[1], cam.cornerLowLeft[2]) +
                f3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]) * s +
                f3(cam.vertical[0], cam.vertical[1], cam.vertical[2]) * t;
    return make_ray(origin, sample - origin);
}

// texHDR (Render.hh:25,42-48) as an equirectangular RGB float image; null -> constant radiance `ambient`
struct EnvMap { const float* rgb; uint32_t w, h; };
TRC_DEV F3 env_radiance(const EnvMap& em, F3 ambient, F3 direction) {
    if (!em.rgb) return ambient;
    const F3 v = normalize(direction);
    const float u = dm_atan2f(v.z, v.x) * 0.1591f + 0.5f;        // SampleSphericalMap
    const float w = dm_asinf(v.y) * 0.3183f + 0.5f;
    const float x = u * (float)em.w - 0.5f, y = w * (float)em.h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float fx = x - fx0, fy = y - fy0;
    auto clampi = [](float f, uint32_t n) { return f < 0.0f ? 0u : (f > (float)(n - 1) ? n - 1 : (uint32_t)f); };
    const uint32_t x0 = clampi(fx0, em.w), x1 = clampi(fx0 + 1.0f, em.w);
    const uint32_t y0 = clampi(fy0, em.h), y1 = clampi(fy0 + 1.0f, em.h);
    auto texel = [&](uint32_t xx, uint32_t yy) { const float* t = em.rgb + 3 * ((size_t)yy * em.w + xx); return f3(t[0], t[1], t[2]); };
    const F3 top = (1 - fx) * texel(x0, y0) + fx * texel(x1, y0);
    const F3 bot = (1 - fx) * texel(x0, y1) + fx * texel(x1, y1);
    return (1 - fy) * top + fy * bot;
}

struct PathCtx {
    SceneRef S;
    F3 root_min, root_max;
    Shade sh;
    F3 ambient;
    EnvMap env;
    uint32_t* stack;
    uint32_t* lvstack;
    uint32_t max_depth;
    // traceVolume: density grid of the GridDensity medium (PackageEnv ids 3/4, Render.hh:30-31); null when absent
    const float* density;
    trc_GridDensityInfo dinfo;
    const uint8_t* occupancy;     // 1 byte per 4x4x4 brick of the grid: 0 = every cell a lookup inside could touch is zero
    // TRC_FLAG_SOBOL: tables of include/trc_sobol.h (null otherwise)
    const uint32_t* sobol32;      // [40][52] generator matrices
    const uint64_t* sobol_vdc;    // [52] VdCSobolMatrices[m - 1], then [52] VdCSobolMatricesInv[m - 1]
    uint32_t sobol_m, sobol_res;  // log2Resolution, resolution (SobolSampler.hh:56-58)
    uint32_t sobol_xy[2];         // thread_pos within its view
};

// ---------------------------------------------------------------- path state machine
// The reference runs, per pixel and per frame, castRay -> tracePath/traceMIS (a bounce loop around
// Scene::hit).  Executed literally on a 64-wide wavefront, every lane waits for the longest path of the
// wave in every sample (measured: 14.6 % VALU lane utilisation).  Here each lane is a small state
// machine that performs ONE Scene::hit per iteration of a single flat loop and immediately starts its
// next sample when a path ends (path regeneration), so lanes stay busy until their pixel's samples
// are exhausted.  Per-path arithmetic and RNG consumption are exactly the reference's.
struct PathState {
    Ray ray;
    HitRec rec;
    F3 ratio, color;
    F3 scat_attenuation;     // traceMIS: BxRecord carried from the BSDF sample to the next hit (:364-365,395-401)
    float scat_bxPDF;
    int depth_left;          // bounce rays the do-while may still trace (Render.metal:406,489)
    bool primary;            // the ray in flight is the camera ray
    int medium;              // traceVolume: Ray::medium (Ray.hh:18) of the ray in flight
    bool from_bsdf;          // traceVolume: the ray in flight left the BSDF-sampling branch (Render.metal:255-271 applies)
    bool tracking;           // traceVolume, TRC_TRACK_SLICE: the delta tracker of the GridDensity medium is part-way (grid_sample_slice)
    float trk_t;
    int trk_step;
    uint64_t sobol_index;    // TRC_FLAG_SOBOL: mSobolIndex of this sample and the next dimension (SobolSampler.hh:37-41)
    uint32_t sobol_dim;
};

// ---------------------------------------------------------------- pbrt::SobolSampler (SobolSampler.hh:26-167)
// SobolIntervalToIndex, :126-148
TRC_DEV uint64_t sobol_interval_to_index(const PathCtx& cx, uint64_t sampleIndex) {
    const uint32_t m = cx.sobol_m;
    if (m == 0) return 0;
    uint64_t index = sampleIndex << (m << 1);
    uint64_t delta = 0;
    for (int c = 0; sampleIndex; sampleIndex >>= 1, ++c)
        if (sampleIndex & 1) delta ^= cx.sobol_vdc[c];
    uint64_t b = (((uint64_t)cx.sobol_xy[0] << m) | cx.sobol_xy[1]) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= cx.sobol_vdc[TRC_SOBOL_MATRIX_SIZE + c];
    return index;
}
// SampleDimension, :152-163 over SobolSampleFloat, :150-160 (the column walk stops at 52 columns: indices stay
// below 2^52, see oracle/oracle.cpp SobolSampleFloat)
TRC_DEV float sobol_dimension(const PathCtx& cx, uint64_t index, uint32_t dimension) {
    if (dimension >= TRC_SOBOL_DIMS) return 0;
    uint32_t v = 0;
    const uint32_t* col = cx.sobol32 + dimension * TRC_SOBOL_MATRIX_SIZE;
    for (; index != 0; index >>= 1, ++col)
        if (index & 1) v ^= *col;
    float s = fminf((float)v * 2.3283064365386963e-10f, 1.0f - FLT_EPSILON);
    if (dimension <= 1) {
        s = s * (float)cx.sobol_res + 0.0f;
        s = fminf(fmaxf(s - (float)(dimension == 0 ? cx.sobol_xy[0] : cx.sobol_xy[1]), 0.0f), 1.0f - FLT_EPSILON);
    }
    return s;
}
// XSampler::sample2D(): RandomSampler.hh:19-24 or SobolSampler.hh:67-72
template <bool SOBOL>
TRC_DEV F2 sample_2d(const PathCtx& cx, PathState& ps, Pcg& rng) {
    F2 uu;
    if (SOBOL) {
        uu.x = sobol_dimension(cx, ps.sobol_index, ps.sobol_dim++);
        uu.y = sobol_dimension(cx, ps.sobol_index, ps.sobol_dim++);
    } else {
        uu.x = pcg_float(rng);
        uu.y = pcg_float(rng);
    }
    return uu;
}

TRC_DEV void path_begin(PathState& ps, const Ray& camera_ray, uint32_t max_depth) {
    ps.ray = camera_ray;
    hit_init(ps.rec);
    ps.ratio = f3(1.0f);
    ps.color = f3(0.0f);
    ps.scat_attenuation = f3(0.0f);
    ps.scat_bxPDF = 1.0f;
    ps.depth_left = (int)max_depth;
    ps.primary = true;
    ps.medium = TRC_MEDIUM_NIL;
    ps.from_bsdf = false;
    ps.tracking = false; ps.trk_t = 0.0f; ps.trk_step = 0;
}

// What happens between two Scene::hit calls of tracePath (Render.metal:432-489).  Returns true when the
// path is finished; `result` is then the sample's radiance.
template <bool STATS, bool SOBOL = false, class COUNT = uint32_t>
TRC_DEV bool path_step(const PathCtx& cx, PathState& ps, bool hitted, Pcg& rng, TravCounters& cnt, COUNT& n_shaded, F3& result) {
    if (!ps.primary) {                                               // } while ((--depth) > 0), :489
        if (--ps.depth_left <= 0) { result = ps.color; return true; }
    }
    ps.primary = false;
    if (!hitted) { result = ps.color + ps.ratio * env_radiance(cx.env, cx.ambient, ps.ray.d); return true; }        // :434-439
    HitRec& rec = ps.rec;
    const int mtype = mat_type(cx.sh, rec.material);
    if (mtype == kMatDiffuse) {                                      // emitter, :441-445
        F3 le = mat_albedo(cx.sh, rec.material);
        float w = dot(-ps.ray.d, -rec.gn);
        result = ps.ratio * le * fabsf(w);
        return true;
    }
    const F2 uu = sample_2d<SOBOL>(cx, ps, rng);                     // sample2D, :447
    const F3 hit_origin = rec.p;
    F3 _origin = offset_ray(rec.p, rec.sn);
    F3 nx, ny;
    coordinate_system(rec.sn, nx, ny);
    F3 minus_d = -ps.ray.d;
    F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));   // wts * (-dir)
    F3 wi = f3(0);
    float bxPDF = 0;                                                 // uninitialised in the reference (B-3)
    bump(n_shaded);
    prof<STATS>(cnt, kProfShade);
    F3 attenuation = material_S_F<STATS>(mtype, hit_color(cx.S, cx.sh, rec), wo, wi, uu, bxPDF, cnt);
    if (bxPDF <= 0) { result = ps.color; return true; }
    F3 wiw = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;                // stw * wi
    if (wi.z < 0) ps.ray = make_ray(offset_ray(hit_origin, -rec.sn), wiw);   // transmission
    else ps.ray = make_ray(_origin, wiw);
    ps.ratio = ps.ratio * div_shared(attenuation, fmaxf(FLT_EPSILON, bxPDF));
    {   // Russian roulette on luminance, :479-485
        float p = rgb_to_y(ps.ratio);
        if (pcg_float(rng) > p) { result = ps.color; return true; }
        ps.ratio = ps.ratio * rcp1(p);
    }
    return false;
}

// ---------------------------------------------------------------- participating media (traceVolume)
TRC_DEV float phase_hg(float cosTheta, float g) {                    // HitRecord.hh:45-49
    float gg = g * g;
    float denom = 1 + gg + 2 * g * cosTheta;
    return (0.25f / kPi) * (1 - gg) / (denom * sqrt_cr(denom));
}
TRC_DEV void hg_sample_p(float g, F3 wo, F3& wi, F2 uu) {            // HitRecord.hh:58-77 (the returned pdf is unused)
    float cosTheta;
    if (fabsf(g) < 1e-3f) cosTheta = 1 - 2 * uu.x;
    else {
        float gg = g * g;
        float sqrTerm = (1 - gg) / (1 + g - 2 * g * uu.x);
        cosTheta = -(1 + gg - sqrTerm * sqrTerm) / (2 * g);
    }
    float sinTheta = sqrt_cr(fmaxf(0.0f, 1 - cosTheta * cosTheta));
    float phi = 2 * kPi * uu.y;
    F3 v1, v2;
    coordinate_system(wo, v1, v2);
    float sp, cp;
    dm_sincosf(phi, &sp, &cp);
    wi = (sinTheta * cp * v1 + sinTheta * sp * v2) + cosTheta * wo;  // SphericalDirection, Sampling.hh:40-43
}
struct MediumHit { F3 p; float phaseG; bool sampled; };
// HomogeneousMedium(0.02, 0.08, 0.5).Sample, Medium.hh:38-74 / Render.metal:118-119
TRC_DEV F3 homogeneous_sample(const Ray& ray, const HitRec& rec, MediumHit& mi, Pcg& rng) {
    const F3 sigma_a = f3(0.02f), sigma_s = f3(0.08f), sigma_t = sigma_s + sigma_a;
    const int nSamples = 3;
    int channel = (int)(pcg_float(rng) * nSamples);
    if (channel > nSamples - 1) channel = nSamples - 1;
    float dist = -dm_logf(1 - pcg_float(rng)) / comp(sigma_t, (uint32_t)channel);
    float t = fminf(dist, rec.t);
    const bool sampledMedium = t < rec.t;
    const float tt = fminf(t, FLT_MAX);
    F3 Tr = f3(dm_expf(-sigma_t.x * tt), dm_expf(-sigma_t.y * tt), dm_expf(-sigma_t.z * tt));
    F3 density = Tr, result = Tr;
    if (sampledMedium) {
        mi.p = point_at(ray, t);
        mi.phaseG = 0.5f;
        mi.sampled = true;
        density = density * sigma_t;
        result = result * sigma_s;
    }
    float pdf = dot(f3(1.0f), density);
    if (0.0f >= pdf) pdf = 1.0f; else pdf = pdf / nSamples;
    return result / pdf;
}
// GridDensityMedium::D / Density / Sample, Medium.hh:111-199
TRC_DEV float grid_D(const trc_GridDensityInfo& info, const float* density, int x, int y, int z) {
    const int nx = (int)info.nx, ny = (int)info.ny, nz = (int)info.nz;
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0;
    return density[((size_t)z * ny + y) * nx + x];
}
TRC_DEV float lerp_f(float t, float s1, float s2) { return (1 - t) * s1 + t * s2; }          // Sampling.hh:13-16
TRC_DEV int grid_to_int(float f) { return !(f > -2.0e9f) ? -2000000000 : (f > 2.0e9f ? 2000000000 : (int)f); }
TRC_DEV float grid_density(const trc_GridDensityInfo& info, const float* density, F3 p) {
    const float nx = (float)info.nx, ny = (float)info.ny, nz = (float)info.nz;
    const F3 ps = f3(p.x * nx - 0.5f, p.y * ny - 0.5f, p.z * nz - 0.5f);
    const int ix = grid_to_int(floorf(ps.x)), iy = grid_to_int(floorf(ps.y)), iz = grid_to_int(floorf(ps.z));
    const F3 d = f3(ps.x - (float)ix, ps.y - (float)iy, ps.z - (float)iz);
    // the 8 corners: GridDensityMedium::D returns 0 outside the grid (Medium.hh:111-127), six compares per corner; a lookup
    // whose cell and its +1 neighbours all lie inside -- every lookup but those in the outermost layer -- needs one range test
    float c000, c100, c010, c110, c001, c101, c011, c111;
    const int gx = (int)info.nx, gy = (int)info.ny, gz = (int)info.nz;
    if (ix >= 0 && iy >= 0 && iz >= 0 && ix + 1 < gx && iy + 1 < gy && iz + 1 < gz) {
        const float* b = density + ((size_t)iz * gy + iy) * gx + ix;
        const size_t row = (size_t)gx, slab = (size_t)gx * gy;
        c000 = b[0]; c100 = b[1]; c010 = b[row]; c110 = b[row + 1];
        c001 = b[slab]; c101 = b[slab + 1]; c011 = b[slab + row]; c111 = b[slab + row + 1];
    } else {
        c000 = grid_D(info, density, ix, iy, iz); c100 = grid_D(info, density, ix + 1, iy, iz);
        c010 = grid_D(info, density, ix, iy + 1, iz); c110 = grid_D(info, density, ix + 1, iy + 1, iz);
        c001 = grid_D(info, density, ix, iy, iz + 1); c101 = grid_D(info, density, ix + 1, iy, iz + 1);
        c011 = grid_D(info, density, ix, iy + 1, iz + 1); c111 = grid_D(info, density, ix + 1, iy + 1, iz + 1);
    }
    float d00 = lerp_f(d.x, c000, c100);
    float d10 = lerp_f(d.x, c010, c110);
    float d01 = lerp_f(d.x, c001, c101);
    float d11 = lerp_f(d.x, c011, c111);
    float d0 = lerp_f(d.y, d00, d10);
    float d1 = lerp_f(d.y, d01, d11);
    return lerp_f(d.z, d0, d1);
}
constexpr int kGridSampleMaxSteps = 1 << 16;      // same bound as oracle/oracle.cpp (the reference loop is unbounded)
// The tracker in SLICES of at most `budget` steps (budget < 0: to the end): t and the step count live in the path state, a slice that
// runs out returns kTrackMore and the lane comes back for the next one WITHOUT a Scene::hit in between (trc_render_kernels.hpp) --
// the lanes that are not in the cloud go on with their paths instead of waiting for ~100 steps of somebody else's.  The lane's own
// sequence of operations (and its RNG stream) is the unsliced loop's.
constexpr float kTrackMore = -1.0f;
TRC_DEV float grid_sample(const PathCtx& cx, const HitRec& rec, MediumHit& mi, Pcg& rng, float& t, int& step, int budget) {
    if (!cx.density) return 1.0f;
    const trc_GridDensityInfo& info = cx.dinfo;
    const float tMax = rec.vol_t;
    for (; step < kGridSampleMaxSteps; ++step) {
        if (budget == 0) return kTrackMore;
        --budget;
        t -= dm_logf(1 - pcg_float(rng)) * info.invMaxDensity / info.sigma_t;
        if (t >= tMax) break;
        const F3 p = rec.vol_o + rec.vol_d * t;
        // empty-space shortcut: a lookup whose 2x2x2 footprint lies in an all-zero brick interpolates zeros to +0 (a
        // majorant-only "null collision"), so the 8 gathers and 7 lerps are skipped; the random number is still drawn.
        // (Round 6 also skipped the LOOKUP for the steps a ray certainly stays inside an empty brick -- exit distance worked out
        // once per brick: profiles/r06/exp_empty_runs.patch.  Same bits, +1.6 .. +4.8 %: the brick byte is an L1 hit, the extra
        // divergent branch and the exit arithmetic are not free.  Removed.)
        float dens = 0.0f;
        {
            const int ix = grid_to_int(floorf(p.x * (float)info.nx - 0.5f)), iy = grid_to_int(floorf(p.y * (float)info.ny - 0.5f)),
                      iz = grid_to_int(floorf(p.z * (float)info.nz - 0.5f));
            bool occupied = true;
            if (cx.occupancy) {
                const int bx = (ix + 1) >> 2, by = (iy + 1) >> 2, bz = (iz + 1) >> 2;
                const int nbx = ((int)info.nx + 4) >> 2, nby = ((int)info.ny + 4) >> 2, nbz = ((int)info.nz + 4) >> 2;
                occupied = (ix >= -1 && iy >= -1 && iz >= -1 && bx < nbx && by < nby && bz < nbz) ? cx.occupancy[((size_t)bz * nby + by) * nbx + bx] != 0 : false;
            }
            if (occupied) dens = grid_density(info, cx.density, p);
        }
        if (dens * info.invMaxDensity > pcg_float(rng)) {
            F3 world = p;
            if (rec.vol_cube != kTagNone) {                        // hitRecord.modelMatrix * float4(p, 1)
                const uint32_t* cb = cx.S.small_base + cx.S.off_cubes + rec.vol_cube * kCubeDwords;
                const float4 m0 = ld4(cb + 12), m1 = ld4(cb + 16), m2 = ld4(cb + 20);
                const F3 mc0 = f3(m0.x, m0.y, m0.z), mc1 = f3(m0.w, m1.x, m1.y), mc2 = f3(m1.z, m1.w, m2.x), mc3 = f3(m2.y, m2.z, m2.w);
                world = ((mc0 * p.x + mc1 * p.y) + mc2 * p.z) + mc3;
            }
            mi.p = world;
            mi.phaseG = info.g;
            mi.sampled = true;
            return info.sigma_s / info.sigma_t;
        }
    }
    return 1.0f;
}

// Same for traceMIS (Render.metal:298-406) and, with VOLUME, traceVolume (Render.metal:78-275 = traceMIS + the
// medium block :114-158).  Lights are literally squareList[5] and [6] (:320-324, B-12).
// The shadow ray (any-hit Scene::hit) is traced here, inside the step.
template <bool ALL_LDS, bool STATS, bool VOLUME = false, bool SOBOL = false, bool HYB = false, class COUNT = uint32_t>
TRC_DEV bool mis_step(const PathCtx& cx, PathState& ps, bool hitted, Pcg& rng, TravCounters& cnt, COUNT& n_rays,
                      COUNT& n_shaded, F3& result) {
    HitRec& rec = ps.rec;
    const bool resume = VOLUME && TRC_TRACK_SLICE > 0 && ps.tracking;      // back for the next slice of the delta tracker: nothing below was left undone
    if (!resume) {
    if (!ps.primary) {
        if ((!VOLUME || ps.from_bsdf) && hitted && mat_type(cx.sh, rec.material) == kMatDiffuse) {   // MIS-weighted emitter hit, :390-404
            F3 Li = mat_albedo(cx.sh, rec.material);
            float cosOnLight = dot(-ps.ray.d, rec.sn);
            F3 weight = ps.scat_attenuation * Li * cosOnLight;
            F3 d = rec.p - ps.ray.o;
            float dist2 = dot(d, d);
            float lightPDF = rec.PDF * dist2 / cosOnLight;
            weight = weight * power_heuristic(1, ps.scat_bxPDF, 1, lightPDF);
            result = ps.color + ps.ratio * weight / ps.scat_bxPDF;
            return true;
        }
        if (--ps.depth_left <= 0) { result = ps.color; return true; }   // } while ((--depth) > 0), :406
    }
    ps.primary = false;
    if (!hitted) { result = ps.color + ps.ratio * env_radiance(cx.env, cx.ambient, ps.ray.d); return true; }
    }
    const int mtype = mat_type(cx.sh, rec.material);
    if (!resume && mtype == kMatDiffuse) {
        F3 le = mat_albedo(cx.sh, rec.material);
        float w = dot(-ps.ray.d, -rec.gn);
        result = ps.ratio * le * fabsf(w);
        return true;
    }
    if (VOLUME) {                                                    // Render.metal:114-158
        MediumHit mi;
        mi.p = f3(0); mi.phaseG = 0; mi.sampled = false;
        if (!resume && ps.medium == TRC_MEDIUM_HOMOGENEOUS) ps.ratio = ps.ratio * homogeneous_sample(ps.ray, rec, mi, rng);
        else if (ps.medium == TRC_MEDIUM_GRIDDENSITY) {
            if (!resume) { ps.trk_t = 0.0f; ps.trk_step = 0; }
            const float beam = grid_sample(cx, rec, mi, rng, ps.trk_t, ps.trk_step, TRC_TRACK_SLICE > 0 ? TRC_TRACK_SLICE : -1);
            if (TRC_TRACK_SLICE > 0) {
                ps.tracking = beam == kTrackMore;
                if (ps.tracking) return false;
            }
            ps.ratio = ps.ratio * f3(beam);
        }
        if (mi.sampled) {                                            // scatter inside the medium
            F2 u2; u2.x = pcg_float(rng); u2.y = pcg_float(rng);
            F3 wi;
            hg_sample_p(mi.phaseG, -ps.ray.d, wi, u2);
            ps.ray = make_ray(mi.p, wi);
            ps.medium = mat_medium(cx.sh, rec.material);
            ps.from_bsdf = false;
            return false;                                            // need_test, then `continue`
        }
        if (mtype == kMatNil) {                                      // medium boundary without a surface
            if (dot(ps.ray.d, rec.gn) < 0) {
                ps.ray = make_ray(offset_ray(rec.p, -rec.gn), ps.ray.d);
                ps.medium = mat_medium(cx.sh, rec.material);
            } else {
                ps.ray = make_ray(offset_ray(rec.p, rec.gn), ps.ray.d);
                ps.medium = TRC_MEDIUM_NIL;
            }
            ps.from_bsdf = false;
            return false;
        }
        ps.from_bsdf = true;
    }
    LightSample lsr;
    const F2 uu = sample_2d<SOBOL>(cx, ps, rng);                     // :314
    const F3 hit_origin = rec.p;
    F3 _origin = offset_ray(rec.p, rec.sn);
    if (pcg_float(rng) < 0.5f) square_sample(cx.S, 5, uu, _origin, lsr);
    else square_sample(cx.S, 6, uu, _origin, lsr);
    F3 _dir = lsr.p - _origin;
    F3 _nor = normalize(_dir);
    F3 nx, ny;
    coordinate_system(rec.sn, nx, ny);
    const float _tr = 1.0f;
    const float _dis = length(_dir);
    const Ray _ray = make_ray(_origin, _nor);
    // what the light sample contributes if the shadow ray gets through, reduced to four values BEFORE the walk (the same expressions
    // on the same operands, :345-352): the sample itself (point, normal, material, area pdf) does not have to survive the walk
    float cosOnLight = fabsf(dot(lsr.n, -_nor));
    F3 light_term = mat_albedo(cx.sh, lsr.material) * cosOnLight;
    float liPDF = (_dis * _dis) * lsr.areaPDF / cosOnLight;
    TRC_PIN(light_term.x); TRC_PIN(light_term.y); TRC_PIN(light_term.z); TRC_PIN(liPDF);
    bump(n_rays);
    bool blocked;
    if (STATS) {                                  // the reference's walk (the exact counters are defined on it)
        HitRec shr;
        hit_init(shr);
        blocked = scene_hit<ALL_LDS, STATS, true, false, false, HYB>(cx.S, cx.root_min, cx.root_max, _ray, shr, _dis, cx.stack, cx.lvstack, cnt);
    } else {                                                          // any-hit: the answer does not depend on the order (dev_intersect.hpp)
        blocked = scene_occluded<ALL_LDS, false, HYB>(cx.S, cx.root_min, cx.root_max, _ray, _dis, cx.stack, cx.S.stack_cap);
    }
    const F3 minus_d = -ps.ray.d;
    const F3 base_color = hit_color(cx.S, cx.sh, rec);
    if (!blocked) {                                                  // light sampling, :339-356
        F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
        F3 wi = f3(dot(nx, _ray.d), dot(ny, _ray.d), dot(rec.sn, _ray.d));
        float bxPDF = 0;
        bump(n_shaded);
        F3 weight = material_F(mtype, base_color, wo, wi, uu, bxPDF);
        weight = weight * light_term;
        weight = weight * power_heuristic(1, liPDF, 1, bxPDF);
        ps.color = ps.color + _tr * ps.ratio * weight / liPDF;
    }
    // BXDF sampling, :358-378
    F3 wi = f3(0);
    float bxPDF = 0;
    F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
    bump(n_shaded);
    ps.scat_attenuation = material_S_F(mtype, base_color, wo, wi, uu, bxPDF);
    ps.scat_bxPDF = bxPDF;
    if (bxPDF <= 0) { result = ps.color; return true; }
    F3 wiw = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;
    if (wi.z < 0) {
        ps.ray = make_ray(offset_ray(hit_origin, -rec.sn), wiw);
        if (VOLUME) ps.medium = (dot(wiw, rec.gn) < 0) ? mat_medium(cx.sh, rec.material) : (int)TRC_MEDIUM_NIL;   // :236-243
    } else {
        ps.ray = make_ray(_origin, wiw);
    }
    ps.ratio = ps.ratio * (ps.scat_attenuation / ps.scat_bxPDF);
    {
        float p = rgb_to_y(ps.ratio);
        if (pcg_float(rng) > p) { result = ps.color; return true; }
        ps.ratio = ps.ratio * rcp1(p);
    }
    return false;
}

}  // namespace trcdev

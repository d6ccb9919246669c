// dev_integrator.hpp -- camera ray, tracePath / traceMIS and the per-pixel accumulation loop.
//
// Reference (RT_Metal/Metal/): Random.metal:3-26, RandomSampler.hh:6-46, Camera.hh:59-69,
// Render.metal:277-409 (traceMIS), :411-492 (tracePath), :495-558 (kernelPathTracing).
#pragma once

#include "dev_bsdf.hpp"
#include "dev_intersect.hpp"

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

struct Shade {            // what the integrators need from the material table
    const uint32_t* mats; // kMaterialDwords per material: type, texType, albedo.rgb
};
TRC_DEV int mat_type(const Shade& sh, uint32_t m) { return (int)sh.mats[m * kMaterialDwords]; }
TRC_DEV int mat_tex(const Shade& sh, uint32_t m) { return (int)sh.mats[m * kMaterialDwords + 1]; }
TRC_DEV F3 mat_albedo(const Shade& sh, uint32_t m) {
    const uint32_t* p = sh.mats + m * kMaterialDwords;
    return f3(__uint_as_float(p[2]), __uint_as_float(p[3]), __uint_as_float(p[4]));
}
// texture colour of a hit; the sphere's uv (atan2 + asin, Sphere.hh:19-31) is only materialised here,
// and only when a texture consumes it -- same value as computing it inside hit_test
TRC_DEV F3 hit_color(const Shade& sh, const HitRec& rec) {
    const int tex = mat_tex(sh, rec.material);
    F2 uv = rec.uv;
    if (tex == kTexChecker && (rec.tag >> kTagIndexBits) == 0u) uv = sphere_uv(rec.gn);
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
    float inv = 1.0f / sqrtf(px * px + py * py);
    float rdx = cam.lenRadius * (px * inv), rdy = cam.lenRadius * (py * inv);
    F3 offset = f3(cam.u[0], cam.u[1], cam.u[2]) * rdx + f3(cam.v[0], cam.v[1], cam.v[2]) * rdy;
    F3 origin = f3(cam.lookFrom[0], cam.lookFrom[1], cam.lookFrom[2]) + offset;
    F3 sample = f3(cam.cornerLowLeft[0], cam.cornerLowLeft[1], cam.cornerLowLeft[2]) +
                f3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]) * s +
                f3(cam.vertical[0], cam.vertical[1], cam.vertical[2]) * t;
    return make_ray(origin, sample - origin);
}

struct PathCtx {
    SceneRef S;
    F3 root_min, root_max;
    Shade sh;
    F3 ambient;
    uint32_t* stack;
    uint32_t* lvstack;
    uint32_t max_depth;
};

// Render.metal:411-492
template <bool STATS>
TRC_DEV F3 trace_path(const PathCtx& cx, Ray ray, Pcg& rng, TravCounters& cnt, uint32_t& n_rays, uint32_t& n_shaded) {
    HitRec rec;
    hit_init(rec);
    F3 ratio = f3(1.0f);
    F3 color = f3(0.0f);
    int depth = (int)cx.max_depth;
    n_rays++;
    bool hitted = scene_hit<STATS, false, false>(cx.S, cx.root_min, cx.root_max, ray, rec, FLT_MAX, cx.stack, cx.lvstack, cnt);
    do {
        if (!hitted) { color = color + ratio * cx.ambient; break; }
        const int mtype = mat_type(cx.sh, rec.material);
        if (mtype == kMatDiffuse) {                                   // emitter, :441-445
            F3 le = mat_albedo(cx.sh, rec.material);
            float w = dot(-ray.d, -rec.gn);
            return ratio * le * fabsf(w);
        }
        F2 uu; uu.x = pcg_float(rng); uu.y = pcg_float(rng);          // sample2D, :447
        const F3 hit_origin = rec.p;
        F3 _origin = offset_ray(rec.p, rec.sn);
        F3 nx, ny;
        coordinate_system(rec.sn, nx, ny);
        F3 minus_d = -ray.d;
        F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));   // wts * (-dir)
        F3 wi = f3(0);
        float bxPDF = 0;                                              // uninitialised in the reference (B-3)
        n_shaded++;
        F3 attenuation = material_S_F(mtype, hit_color(cx.sh, rec), wo, wi, uu, bxPDF);
        if (bxPDF <= 0) break;
        F3 wiw = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;             // stw * wi
        if (wi.z < 0) ray = make_ray(offset_ray(hit_origin, -rec.sn), wiw);   // transmission
        else ray = make_ray(_origin, wiw);
        ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
        {   // Russian roulette on luminance, :479-485
            float p = rgb_to_y(ratio);
            if (pcg_float(rng) > p) break;
            ratio = ratio * (1.0f / p);
        }
        n_rays++;
        hitted = scene_hit<STATS, false, false>(cx.S, cx.root_min, cx.root_max, ray, rec, FLT_MAX, cx.stack, cx.lvstack, cnt);
    } while ((--depth) > 0);
    return color;
}

// Render.metal:277-409.  Lights are literally squareList[5] and [6] (:320-324, B-12).
template <bool STATS>
TRC_DEV F3 trace_mis(const PathCtx& cx, Ray ray, Pcg& rng, TravCounters& cnt, uint32_t& n_rays, uint32_t& n_shaded) {
    HitRec rec;
    hit_init(rec);
    F3 scat_attenuation = f3(0);
    float scat_bxPDF = 1.0f;
    F3 ratio = f3(1.0f);
    F3 color = f3(0.0f);
    int depth = (int)cx.max_depth;
    n_rays++;
    bool hitted = scene_hit<STATS, false, false>(cx.S, cx.root_min, cx.root_max, ray, rec, FLT_MAX, cx.stack, cx.lvstack, cnt);
    do {
        if (!hitted) { color = color + ratio * cx.ambient; break; }
        const int mtype = mat_type(cx.sh, rec.material);
        if (mtype == kMatDiffuse) {
            F3 le = mat_albedo(cx.sh, rec.material);
            float w = dot(-ray.d, -rec.gn);
            return ratio * le * fabsf(w);
        }
        LightSample lsr;
        F2 uu; uu.x = pcg_float(rng); uu.y = pcg_float(rng);
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
        HitRec shr;
        hit_init(shr);
        n_rays++;
        const bool blocked = scene_hit<STATS, true, false>(cx.S, cx.root_min, cx.root_max, _ray, shr, _dis, cx.stack, cx.lvstack, cnt);
        const F3 minus_d = -ray.d;
        const F3 base_color = hit_color(cx.sh, rec);
        if (!blocked) {                                               // light sampling, :339-356
            F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
            F3 wi = f3(dot(nx, _ray.d), dot(ny, _ray.d), dot(rec.sn, _ray.d));
            float bxPDF = 0;
            n_shaded++;
            F3 weight = material_F(mtype, base_color, wo, wi, uu, bxPDF);
            float cosOnLight = fabsf(dot(lsr.n, -_nor));
            F3 Li = mat_albedo(cx.sh, lsr.material);
            weight = weight * (Li * cosOnLight);
            float dist2 = _dis * _dis;
            float liPDF = dist2 * lsr.areaPDF / cosOnLight;
            weight = weight * power_heuristic(1, liPDF, 1, bxPDF);
            color = color + _tr * ratio * weight / liPDF;
        }
        // BXDF sampling, :358-378
        F3 wi = f3(0);
        float bxPDF = 0;
        F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
        n_shaded++;
        scat_attenuation = material_S_F(mtype, base_color, wo, wi, uu, bxPDF);
        scat_bxPDF = bxPDF;
        if (bxPDF <= 0) break;
        F3 wiw = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;
        if (wi.z < 0) ray = make_ray(offset_ray(hit_origin, -rec.sn), wiw);
        else ray = make_ray(_origin, wiw);
        ratio = ratio * (scat_attenuation / scat_bxPDF);
        {
            float p = rgb_to_y(ratio);
            if (pcg_float(rng) > p) break;
            ratio = ratio * (1.0f / p);
        }
        n_rays++;
        hitted = scene_hit<STATS, false, false>(cx.S, cx.root_min, cx.root_max, ray, rec, FLT_MAX, cx.stack, cx.lvstack, cnt);
        if (hitted && mat_type(cx.sh, rec.material) == kMatDiffuse) {   // MIS-weighted emitter hit, :390-404
            F3 Li = mat_albedo(cx.sh, rec.material);
            float cosOnLight = dot(-ray.d, rec.sn);
            F3 weight = scat_attenuation * Li * cosOnLight;
            F3 d = rec.p - ray.o;
            float dist2 = dot(d, d);
            float lightPDF = rec.PDF * dist2 / cosOnLight;
            weight = weight * power_heuristic(1, scat_bxPDF, 1, lightPDF);
            color = color + ratio * weight / scat_bxPDF;
            break;
        }
    } while ((--depth) > 0);
    return color;
}

}  // namespace trcdev

// dev_pool.hpp -- LDS ray pool with phase-sorted batches (tracePath on LDS-resident scenes).
//
// WHY (profiles/r01/v2_divergence_profile.txt): with one path per lane the VALUs are ~90 % busy but only 21 %
// of the lanes of an instruction do useful work: box steps run at 36 % lane utilisation, the rare expensive
// pieces (cube test, sphere test, Metal / Plastic / Glass sampling) at 6-12 %.
//
// HOW: every wavefront owns a POOL of kPoolSlots path states in LDS (176 B each, 16-byte groups so a phase
// loads what it needs with ds_read_b128).  A path is a small state machine over four PHASES:
//     T  traversal (box steps + square tests)        L  sphere / cube leaf test
//     R  resolve the hit + Lambert + end-of-sample    M  Metal / Plastic / Glass sampling + end-of-sample
// Per round the wavefront counts the phases of its slots with __ballot, picks the fullest one, writes the
// list of ALL slots in that phase to LDS (mbcnt ranks) and drains it: 64 at a time for L / R / M, and with
// in-loop refill for T (a lane whose path leaves the traversal immediately takes the next listed slot), so
// the expensive code runs on dense batches while other paths wait in LDS instead of idling lanes.
//
// Per-path arithmetic, RNG draws and their order are exactly those of path_step()/scene_hit(): results stay
// bit-identical to the oracle and to the one-path-per-lane kernel (tests compare both).
#pragma once

#include "dev_integrator.hpp"

namespace trcdev {

#ifndef TRC_POOL_SLOTS
#define TRC_POOL_SLOTS 192
#endif
constexpr uint32_t kPoolSlots = TRC_POOL_SLOTS;          // per wavefront (multiple of 32)
constexpr uint32_t kPoolK = (kPoolSlots + 63u) / 64u;    // slots a lane is "home" of
constexpr uint32_t kSlotDwords = 44;                     // 176 B
constexpr uint32_t kPoolStack = 8;                       // traversal stack entries per slot (tree depth <= 8)

// slot layout (dword offsets); every group of 4 is 16-byte aligned
constexpr uint32_t SL_O = 0;       // o.xyz, ry
constexpr uint32_t SL_D = 4;       // d.xyz, tag
constexpr uint32_t SL_INV = 8;     // inv.xyz, sp
constexpr uint32_t SL_RNG = 12;    // state lo, state hi, inc lo, inc hi   (live Pcg during a sample)
constexpr uint32_t SL_RATIO = 16;  // ratio.xyz, meta
constexpr uint32_t SL_P = 20;      // rec.p.xyz, rec.material
constexpr uint32_t SL_GN = 24;     // rec.gn.xyz, rec.tag
constexpr uint32_t SL_SN = 28;     // rec.sn.xyz, pixel index
constexpr uint32_t SL_UV = 32;     // rec.uv.xy, uu.xy
constexpr uint32_t SL_STACK = 36;  // 8 deferred siblings

// meta: s [0,16) | depth_left [16,24) | primary bit 24 | fresh bit 25 | rare material type [26,29)
constexpr uint32_t META_PRIMARY = 1u << 24, META_FRESH = 1u << 25;

enum PoolPhase : uint32_t { PP_T = 0, PP_L, PP_R, PP_M, PP_DONE, PP_COUNT };

struct PoolCounters { uint32_t rays, shaded, paths; };

TRC_DEV void st4(uint32_t* p, float a, float b, float c, uint32_t d) {
    float4 v; v.x = a; v.y = b; v.z = c; v.w = __uint_as_float(d);
    *reinterpret_cast<float4*>(p) = v;
}
TRC_DEV uint32_t mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
TRC_DEV uint32_t pool_phase_of_tag(uint32_t tag) {      // Sphere0 Square1 Cube2 Interior4 (no triangles in pool scenes)
    const uint32_t type = tag >> kTagIndexBits;
    return (type == 0u || type == 2u) ? PP_L : PP_T;
}

// Scene::hit entry for a fresh ray: root box test (Render.hh:143-145); writes the ray + traversal state
TRC_DEV uint32_t pool_start_ray(const PathCtx& cx, uint32_t* sl, const Ray& ray, PoolCounters& pc) {
    pc.rays++;
    const float ry = FLT_MAX;
    const bool in = box_hit(cx.root_min, cx.root_max, ray, FLT_MIN, ry);
    st4(sl + SL_O, ray.o.x, ray.o.y, ray.o.z, __float_as_uint(ry));
    st4(sl + SL_D, ray.d.x, ray.d.y, ray.d.z, kTagInterior << kTagIndexBits);
    st4(sl + SL_INV, ray.inv.x, ray.inv.y, ray.inv.z, 0u);
    return in ? PP_T : PP_R;
}

// pop the next deferred sibling of the slot or finish the traversal; returns the next phase
TRC_DEV uint32_t pool_pop(uint32_t* sl, uint32_t& tag, uint32_t& sp) {
    if (sp == 0) return PP_R;
    sp--;
    tag = sl[SL_STACK + sp];
    return pool_phase_of_tag(tag);
}
TRC_DEV void pool_store_rec(uint32_t* sl, const HitRec& rec, uint32_t tag, bool with_uv) {
    st4(sl + SL_P, rec.p.x, rec.p.y, rec.p.z, rec.material);
    st4(sl + SL_GN, rec.gn.x, rec.gn.y, rec.gn.z, tag);
    sl[SL_SN] = __float_as_uint(rec.sn.x); sl[SL_SN + 1] = __float_as_uint(rec.sn.y); sl[SL_SN + 2] = __float_as_uint(rec.sn.z);
    if (with_uv) { sl[SL_UV] = __float_as_uint(rec.uv.x); sl[SL_UV + 1] = __float_as_uint(rec.uv.y); }
}

// ---------------------------------------------------------------- PP_T
// Resumable Scene::hit over the whole list of slots in phase T: box steps and square tests with in-loop
// refill.  A lane leaves a slot when its traversal is finished (-> PP_R) or it holds a sphere / cube leaf
// (-> PP_L); it then writes the slot back and takes the next listed one.
TRC_DEV void pool_phase_T(const PathCtx& cx, uint32_t* pool, uint32_t* phase_of, const uint32_t* list, uint32_t n_list) {
    const SceneRef& S = cx.S;
    const uint32_t lane = threadIdx.x & 63u;
    const float rx = FLT_MIN;
    Ray ray;
    ray.o = f3(0); ray.d = f3(0); ray.inv = f3(0);
    float ry = 0;
    uint32_t tag = 0, sp = 0, slot = 0;
    uint32_t* sl = pool;
    bool active = false;
    uint32_t cursor = 0;              // wave-uniform: next unassigned list entry
    for (;;) {
        // ---- refill idle lanes from the list
        if (cursor < n_list) {
            const unsigned long long idle = __ballot(!active);
            if (idle) {
                const uint32_t r = cursor + mbcnt64(idle);
                if (!active && r < n_list) {
                    slot = list[r];
                    sl = pool + slot * kSlotDwords;
                    const float4 g0 = ld4(sl + SL_O), g1 = ld4(sl + SL_D), g2 = ld4(sl + SL_INV);
                    ray.o = f3(g0.x, g0.y, g0.z); ry = g0.w;
                    ray.d = f3(g1.x, g1.y, g1.z); tag = __float_as_uint(g1.w);
                    ray.inv = f3(g2.x, g2.y, g2.z); sp = __float_as_uint(g2.w);
                    active = true;
                }
                cursor += (uint32_t)__popcll(idle);
            }
        }
        if (!__ballot(active)) break;
        // ---- box steps until a quarter of the lanes went idle (or nobody is at an interior node)
        uint32_t phase = PP_T;
        for (;;) {
            const bool interior = active && (tag >> kTagIndexBits) == kTagInterior;
            if (!__ballot(interior)) break;
            if (interior) {
                float4 q0, q1, q2, q3;
                load_node<true>(S, tag & kTagIndexMask, q0, q1, q2, q3);
                float t_left = ry, t_right = ry;
                const bool left_test = box_hit_t(f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), ray, rx, ry, t_left);
                const bool right_test = box_hit_t(f3(q1.z, q1.w, q2.x), f3(q2.y, q2.z, q2.w), ray, rx, ry, t_right);
                if (left_test || right_test) {
                    const uint32_t tagL = __float_as_uint(q3.z), tagR = __float_as_uint(q3.w);
                    const bool left_first = t_left < t_right;            // Render.hh:174, literal
                    if (left_test && right_test) { sl[SL_STACK + sp] = left_first ? tagR : tagL; sp++; }
                    tag = left_first ? tagL : tagR;
                    phase = pool_phase_of_tag(tag);
                } else {
                    phase = pool_pop(sl, tag, sp);
                }
            }
            // leave the stepping loop when enough lanes want a square test / a write-back + refill
            const bool still = active && phase == PP_T && (tag >> kTagIndexBits) == kTagInterior;
            if ((uint32_t)__popcll(__ballot(still)) < 40u) break;
        }
        // ---- square leaves
        const bool at_square = active && phase == PP_T && (tag >> kTagIndexBits) == 1u;
        if (__ballot(at_square)) {
            if (at_square) {
                HitRec rec;
                if (square_hit_test(S, tag & kTagIndexMask, ray, rx, ry, rec)) {
                    pool_store_rec(sl, rec, tag, true);
                    sl[SL_O + 3] = __float_as_uint(ry);
                }
                phase = pool_pop(sl, tag, sp);
            }
        }
        // ---- lanes whose slot left phase T: write back, go idle
        if (active && phase != PP_T) {
            sl[SL_D + 3] = tag;
            sl[SL_INV + 3] = sp;
            phase_of[slot] = phase;
            active = false;
        }
    }
    (void)lane;
}

// ---------------------------------------------------------------- PP_L: one sphere / cube test, then pop
TRC_DEV void pool_phase_L(const PathCtx& cx, uint32_t* sl, bool have, uint32_t& next_phase) {
    if (!have) return;
    const float4 g0 = ld4(sl + SL_O), g1 = ld4(sl + SL_D), g2 = ld4(sl + SL_INV);
    Ray ray;
    ray.o = f3(g0.x, g0.y, g0.z); ray.d = f3(g1.x, g1.y, g1.z); ray.inv = f3(g2.x, g2.y, g2.z);
    float ry = g0.w;
    uint32_t tag = __float_as_uint(g1.w), sp = __float_as_uint(g2.w);
    HitRec rec;
    rec.uv.x = 0; rec.uv.y = 0;
    const bool is_sphere = (tag >> kTagIndexBits) == 0u;
    bool ok;
    if (is_sphere) {
        ok = sphere_hit_test<false>(cx.S, tag & kTagIndexMask, ray, FLT_MIN, ry, rec);
    } else {
        TravCounters dead;
        ok = cube_hit_test<false>(cx.S, tag & kTagIndexMask, ray, ry, rec, dead);
    }
    if (ok) {
        pool_store_rec(sl, rec, tag, !is_sphere);           // the sphere's uv is materialised lazily from gn
        sl[SL_O + 3] = __float_as_uint(ry);
    }
    next_phase = pool_pop(sl, tag, sp);
    sl[SL_D + 3] = tag;
    sl[SL_INV + 3] = sp;
}

// ---------------------------------------------------------------- shading helpers
struct PoolPath {       // registers of one path during PP_R / PP_M
    Pcg rng;
    F3 ratio;
    uint32_t meta;
    F3 d;               // direction of the ray that produced the hit
    HitRec rec;
};
TRC_DEV void pool_load_path(const uint32_t* sl, PoolPath& p) {
    const float4 g1 = ld4(sl + SL_D), g3 = ld4(sl + SL_RNG), g4 = ld4(sl + SL_RATIO);
    const float4 g5 = ld4(sl + SL_P), g6 = ld4(sl + SL_GN), g7 = ld4(sl + SL_SN), g8 = ld4(sl + SL_UV);
    p.d = f3(g1.x, g1.y, g1.z);
    p.rng.state = ((uint64_t)__float_as_uint(g3.y) << 32) | __float_as_uint(g3.x);
    p.rng.inc = ((uint64_t)__float_as_uint(g3.w) << 32) | __float_as_uint(g3.z);
    p.ratio = f3(g4.x, g4.y, g4.z); p.meta = __float_as_uint(g4.w);
    p.rec.t = 0; p.rec.PDF = 0;
    p.rec.p = f3(g5.x, g5.y, g5.z); p.rec.material = __float_as_uint(g5.w);
    p.rec.gn = f3(g6.x, g6.y, g6.z); p.rec.tag = __float_as_uint(g6.w);
    p.rec.sn = f3(g7.x, g7.y, g7.z);
    p.rec.uv.x = g8.x; p.rec.uv.y = g8.y;
}
TRC_DEV void pool_store_rng(uint32_t* sl, const Pcg& r) {
    st4(sl + SL_RNG, __uint_as_float((uint32_t)r.state), __uint_as_float((uint32_t)(r.state >> 32)),
        __uint_as_float((uint32_t)r.inc), (uint32_t)(r.inc >> 32));
}

// Render.metal:453-487 after uu was drawn: sample BSDF `mtype`, build the next ray, roulette.
// Returns true when the path ENDED (result = 0), else the slot holds the new ray and `next_phase`.
template <int MTYPE>
TRC_DEV bool pool_shade(const PathCtx& cx, uint32_t* sl, PoolPath& p, F2 uu, PoolCounters& pc, uint32_t& next_phase) {
    HitRec& rec = p.rec;
    const F3 hit_origin = rec.p;
    F3 _origin = offset_ray(rec.p, rec.sn);
    F3 nx, ny;
    coordinate_system(rec.sn, nx, ny);
    F3 minus_d = -p.d;
    F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
    F3 wi = f3(0);
    float bxPDF = 0;
    pc.shaded++;
    F3 attenuation = material_S_F(MTYPE, hit_color(cx.sh, rec), wo, wi, uu, bxPDF);
    if (bxPDF <= 0) return true;
    F3 wiw = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;
    Ray ray;
    if (wi.z < 0) ray = make_ray(offset_ray(hit_origin, -rec.sn), wiw);
    else ray = make_ray(_origin, wiw);
    p.ratio = p.ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
    const float pr = rgb_to_y(p.ratio);
    if (pcg_float(p.rng) > pr) return true;
    p.ratio = p.ratio * (1.0f / pr);
    st4(sl + SL_RATIO, p.ratio.x, p.ratio.y, p.ratio.z, p.meta);
    pool_store_rng(sl, p.rng);
    next_phase = pool_start_ray(cx, sl, ray, pc);
    return false;
}

// Render.metal:432-447 + the `while (--depth > 0)` of :489 for one slot.  Outcomes:
//   ended = true  with `result`      (depth exhausted, miss, emitter, unsupported material)
//   ended = false, next_phase = PP_T / PP_R (Lambert shaded inline, new ray stored) or PP_M (rare BSDF)
TRC_DEV void pool_resolve(const PathCtx& cx, uint32_t* sl, PoolPath& p, PoolCounters& pc, bool& ended, F3& result,
                          uint32_t& next_phase) {
    ended = true; result = f3(0.0f);
    const float ry = __uint_as_float(sl[SL_O + 3]);
    const bool hitted = ry < FLT_MAX;
    if (!(p.meta & META_PRIMARY)) {                                   // } while ((--depth) > 0)
        uint32_t depth_left = (p.meta >> 16) & 0xFFu;
        depth_left = depth_left > 0 ? depth_left - 1 : 0;
        p.meta = (p.meta & ~0x00FF0000u) | (depth_left << 16);
        if (depth_left == 0) return;
    }
    p.meta &= ~META_PRIMARY;
    if (!hitted) { result = f3(0.0f) + p.ratio * cx.ambient; return; }
    const int mtype = mat_type(cx.sh, p.rec.material);
    if (mtype == kMatDiffuse) {
        F3 le = mat_albedo(cx.sh, p.rec.material);
        float w = dot(-p.d, -p.rec.gn);
        result = p.ratio * le * fabsf(w);
        return;
    }
    F2 uu; uu.x = pcg_float(p.rng); uu.y = pcg_float(p.rng);
    if (mtype == kMatLambert) {
        ended = pool_shade<kMatLambert>(cx, sl, p, uu, pc, next_phase);
        return;
    }
    if (mtype == kMatMetal || mtype == kMatPlastic || mtype == kMatGlass) {
        sl[SL_UV + 2] = __float_as_uint(uu.x); sl[SL_UV + 3] = __float_as_uint(uu.y);
        sl[SL_RATIO + 3] = (p.meta & ~(7u << 26)) | ((uint32_t)mtype << 26);
        pool_store_rng(sl, p.rng);
        next_phase = PP_M;
        ended = false;
        return;
    }
    // Material::S_F default: returns 0 and leaves the pdf at 0 -> the loop breaks with color = 0
}

}  // namespace trcdev

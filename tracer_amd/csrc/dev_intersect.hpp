// dev_intersect.hpp -- ray/box, ray/primitive tests and the BVH traversal on the device.
//
// Reference functions restated here (RT_Metal/Metal/):
//   AABB::hit / hit_t            AABB.hh:73-112
//   AABB::hit(ray,range,record)  AABB.hh:114-209   (used by Cube)
//   Sphere::hit_test             Sphere.hh:33-78
//   Square::hit_test / sample    Square.hh:40-113
//   Cube::hit_test               Cube.hh:17-47
//   Triangle::hit_test           Triangle.hh:31-85
//   Scene::hit                   Render.hh:135-252
//
// Traversal design (MI355X): the reference walks parent pointers with a 32-bit "visit the
// sibling later" bitmask and spends one loop iteration per level on the way back up.  Here the
// deferred sibling is pushed on a per-lane short stack in LDS (column layout: entry e of lane l
// at stack[e * 64 + l], so a wavefront's push/pop is one conflict-free ds_write/ds_read_b32;
// the tracePath render kernels keep only the first 16 entries there and deeper ones in global
// rows: stack_put below), which removes every upward iteration while visiting boxes and primitives in EXACTLY the
// reference's order -- near child first by hit_t, far child later WITHOUT re-testing its box
// (Render.hh:171-174,204-208) -- so closest-hit ties resolve identically.  The upward
// iterations the reference would have executed are still COUNTED (exactly) in the
// instrumented build, because the roofline's algorithmic bytes are defined on them.
#pragma once

#include "dev_prof.hpp"
#include "dev_scene.hpp"
#include "dev_vec.hpp"

namespace trcdev {

constexpr int kBlock = 64;    // threads per workgroup = ONE wavefront = one 8x8 pixel block (a quarter of a 16x16 tile)

struct Ray {
    F3 o, d, inv;     // inv = 1.0 / direction, computed once per ray (AABB.hh:75,94 recompute it per box)
};
TRC_DEV Ray make_ray(F3 o, F3 dir) {     // Ray::Ray normalises (Ray.hh:21-23)
    Ray r; r.o = o; r.d = normalize(dir); r.inv = rcp_cr(r.d); return r;
}
TRC_DEV F3 point_at(const Ray& r, float t) { return r.o + r.d * t; }

struct HitRec {       // HitRecord.hh:9-30 (live fields) + the primitive tag
    float t;
    F3 p, gn, sn;
    F2 uv;
    uint32_t material;
    float PDF;
    uint32_t tag;
    // traceVolume only (HitRecord.hh:22-23, written by Cube::hit_test, Cube.hh:30-31,39): the object-space ray and
    // t of the last cube hit and that cube (for its model matrix); dead code in the other integrators
    F3 vol_o, vol_d;
    float vol_t;
    uint32_t vol_cube;
};
TRC_DEV void hit_init(HitRec& h) {
    h.t = 0; h.p = f3(0); h.gn = f3(0); h.sn = f3(0); h.uv.x = 0; h.uv.y = 0; h.material = 0; h.PDF = 0; h.tag = kTagNone;
    h.vol_o = f3(0); h.vol_d = f3(0); h.vol_t = 0; h.vol_cube = kTagNone;
}
TRC_DEV void check_face(HitRec& h, const Ray& ray) {   // HitRecord.hh:26-29
    bool f = dot(ray.d, h.gn) <= 0;
    h.sn = f ? h.gn : -h.gn;
}

// scene accessor: `small_base` is the LDS copy of the blob prefix (analytic prims, materials, top fat nodes)
struct SceneRef {
    const uint32_t* small_base;
    const uint32_t* blob;
    uint32_t off_nodes, off_spheres, off_squares, off_cubes, off_materials, off_tripos, off_triattr;
    uint32_t n_lds_nodes;
    uint32_t stack_lds;       // traversal-stack entries per lane that live in LDS ...
    uint32_t stack_cap;       // ... of stack_cap in all (the tree's depth)
    uint32_t* ovf;            // ... deeper ones in this lane's column of the workgroup's overflow rows (global memory)
    uint32_t descend_min;     // DScene::descend_min
};
TRC_DEV float4 ld4(const uint32_t* p) { return *reinterpret_cast<const float4*>(p); }
// The same 16-byte load with the address space spelled out.  Where a lane reads either the LDS copy or the blob, the
// compiler otherwise merges both sides into ONE flat_load through a selected pointer, and a FLAT instruction is issued to
// the LDS and the vector-memory path alike (and returns out of order: both counters must drain).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float trc_v4f __attribute__((ext_vector_type(4)));
TRC_DEV float4 ld4_global(const uint32_t* p) {
    const trc_v4f v = *(const __attribute__((address_space(1))) trc_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
TRC_DEV float4 ld4_lds(const uint32_t* p) {
    const trc_v4f v = *(const __attribute__((address_space(3))) trc_v4f*)p;
    return make_float4(v.x, v.y, v.z, v.w);
}
TRC_DEV uint32_t ld1_global(const uint32_t* p) { return *(const __attribute__((address_space(1))) uint32_t*)p; }
TRC_DEV void st1_global(uint32_t* p, uint32_t v) { *(__attribute__((address_space(1))) uint32_t*)p = v; }
#else
TRC_DEV float4 ld4_global(const uint32_t* p) { return ld4(p); }
TRC_DEV float4 ld4_lds(const uint32_t* p) { return ld4(p); }
TRC_DEV uint32_t ld1_global(const uint32_t* p) { return *p; }
TRC_DEV void st1_global(uint32_t* p, uint32_t v) { *p = v; }
#endif

// fat-node fetch: LDS for the staged top of the tree, global memory below it.  ALL_LDS = whole tree staged.
template <bool ALL_LDS>
TRC_DEV void load_node(const SceneRef& S, uint32_t idx, float4& q0, float4& q1, float4& q2, float4& q3) {
    if (ALL_LDS || idx < S.n_lds_nodes) {
        const uint32_t* np = S.small_base + S.off_nodes + idx * kNodeDwords;
        q0 = ld4_lds(np); q1 = ld4_lds(np + 4); q2 = ld4_lds(np + 8); q3 = ld4_lds(np + 12);
    } else {
        const uint32_t* np = S.blob + S.off_nodes + (size_t)idx * kNodeDwords;
        q0 = ld4_global(np); q1 = ld4_global(np + 4); q2 = ld4_global(np + 8); q3 = ld4_global(np + 12);
    }
}

// ---------------------------------------------------------------- boxes
TRC_DEV bool box_hit(F3 mn, F3 mx, const Ray& r, float rx, float ry) {      // AABB.hh:73-90
    F3 ts = (mn - r.o) * r.inv;
    F3 te = (mx - r.o) * r.inv;
    float tmin = fmax3(fminf(ts.x, te.x), fminf(ts.y, te.y), fminf(ts.z, te.z));
    float tmax = fmin3(fmaxf(ts.x, te.x), fmaxf(ts.y, te.y), fmaxf(ts.z, te.z));
    tmin = fmaxf(tmin, rx);
    tmax = fminf(tmax, ry);
    return !(tmax < tmin || tmax < 0);
}
TRC_DEV bool box_hit_t(F3 mn, F3 mx, const Ray& r, float rx, float ry, float& t) {   // AABB.hh:92-112
    F3 ts = (mn - r.o) * r.inv;
    F3 te = (mx - r.o) * r.inv;
    float tmin = fmax3(fminf(ts.x, te.x), fminf(ts.y, te.y), fminf(ts.z, te.z));
    float tmax = fmin3(fmaxf(ts.x, te.x), fmaxf(ts.y, te.y), fmaxf(ts.z, te.z));
    tmin = fmaxf(tmin, rx);
    tmax = fminf(tmax, ry);
    if (tmax < tmin || tmax < 0) return false;
    t = (tmin < 0) ? tmax : tmin;
    return true;
}

// Math.hh:51-55: gamma(3) with MachineEpsilon = FLT_EPSILON * 0.5 (unparenthesised macro)
TRC_DEV float box_pad() { return 1 + 2 * ((3 * FLT_EPSILON * 0.5f) / (1 - 3 * FLT_EPSILON * 0.5f)); }

// AABB.hh:114-209: object-space box test of Cube; o/d are the object-space ray (d normalised)
TRC_DEV bool box_hit_record(F3 mini, F3 maxi, F3 o, F3 d, float range_y, float& out_t, F3& out_gn, F3& out_p, F2& out_uv) {
    float tmin = -FLT_MAX;
    float tmax = range_y;
    uint32_t axisPick = 0;
    const F3 ddd = o - mini, bbb = o - maxi;
    const float pad = box_pad();
    const bool inside = (ddd.x > 0 && ddd.y > 0 && ddd.z > 0) && (bbb.x < 0 && bbb.y < 0 && bbb.z < 0);
#if TRC_WAVE_GUARDS
    // the six slab quotients share three divisors: reciprocal + Newton step once per axis, the quotients' own corrections,
    // no operand guards -- unless some lane's operand is outside [2^-60, 2^60] (a direction component that is 0 or tiny, an
    // origin ON a slab plane), then everybody divides the long way.  Same bits either way (tests/test_gpu_divby.py).
    const F3 nlo = mini - o, nhi = maxi - o;
    F3 qlo, qhi;
    {
        const GuardedDivBy bx = guarded_div_by(d.x), by = guarded_div_by(d.y), bz = guarded_div_by(d.z);
        qlo = f3(div_core(nlo.x, bx), div_core(nlo.y, by), div_core(nlo.z, bz));
        qhi = f3(div_core(nhi.x, bx), div_core(nhi.y, by), div_core(nhi.z, bz));
        const float small = fminf(fmin3(fmin3(fabsf(nlo.x), fabsf(nlo.y), fabsf(nlo.z)), fmin3(fabsf(nhi.x), fabsf(nhi.y), fabsf(nhi.z)),
                                        fmin3(fabsf(d.x), fabsf(d.y), fabsf(d.z))), 0x1p60f);
        const float large = fmax3(fmax3(fabsf(nlo.x), fabsf(nlo.y), fabsf(nlo.z)), fmax3(fabsf(nhi.x), fabsf(nhi.y), fabsf(nhi.z)),
                                  fmax3(fabsf(d.x), fabsf(d.y), fabsf(d.z)));
        if (__builtin_expect(!wave_all(small >= 0x1p-60f && large <= 0x1p60f), 0)) { qlo = nlo / d; qhi = nhi / d; }
    }
#endif
#pragma unroll
    for (uint32_t i = 0; i < 3; ++i) {
        float oi = comp(o, i), di = comp(d, i);
#if TRC_WAVE_GUARDS
        (void)oi; (void)di;
        float min_bound = comp(qlo, i);
        float max_bound = comp(qhi, i);
#else
        float min_bound = (comp(mini, i) - oi) / di;
        float max_bound = (comp(maxi, i) - oi) / di;
#endif
        float ts = fminf(max_bound, min_bound);
        float te = fmaxf(max_bound, min_bound);
        te *= pad;
        if (inside) {
            tmin = fmaxf(ts, tmin);
            if (te < tmax) { tmax = te; axisPick = i; }
        } else {
            tmax = fminf(te, tmax);
            if (ts > tmin) { tmin = ts; axisPick = i; }
        }
        if (tmax < tmin || tmax < 0) return false;
    }
    const float dpick = comp(d, axisPick);
    const float t = inside ? tmax : tmin;
    out_t = t;
    F3 gn = f3(0);
    set_comp(gn, axisPick, inside ? (dpick > 0 ? 1.0f : -1.0f) : (dpick > 0 ? -1.0f : 1.0f));
    out_gn = gn;
    F3 hitPoint = o + d * t;
    F3 p = hitPoint;
    set_comp(p, axisPick, inside ? (dpick > 0 ? comp(maxi, axisPick) : comp(mini, axisPick))
                                 : (dpick > 0 ? comp(mini, axisPick) : comp(maxi, axisPick)));
    out_p = p;
    uint32_t ax = axisPick + 1; ax = ax >= 3 ? ax - 3 : ax;
    uint32_t ay = axisPick + 2; ay = ay >= 3 ? ay - 3 : ay;
    out_uv.x = comp(hitPoint, ax);
    out_uv.y = comp(hitPoint, ay);
    return true;
}

// ---------------------------------------------------------------- primitives
TRC_DEV F2 sphere_uv(F3 p) {    // Sphere.hh:19-31
    float phi = dm_atan2f(p.z, p.x);
    float theta = dm_asinf(p.y);
    F2 uv;
    uv.x = 1 - (phi + kPi) / (2 * kPi);
    uv.y = (theta + kPi2) / kPi;
    return uv;
}

template <bool EAGER_UV>
TRC_DEV bool sphere_hit_test(const SceneRef& S, uint32_t index, const Ray& ray, float rx, float& ry, HitRec& rec) {
    const uint32_t* sp = S.small_base + S.off_spheres + index * kSphereDwords;
    const float4 cr = ld4(sp);
    const F3 center = f3(cr.x, cr.y, cr.z);
    const float radius = cr.w;
    F3 oc = ray.o - center;
    float a = dot(ray.d, ray.d);
    float half_b = dot(oc, ray.d);
    float c = dot(oc, oc) - radius * radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant <= 0) return false;
    float root = sqrt_cr(discriminant);
    float temp = (-half_b - root) / a;
    if (!(temp < ry && temp > rx)) {
        temp = (-half_b + root) / a;
        if (!(temp < ry && temp > rx)) return false;
    }
    rec.t = temp;
    rec.p = point_at(ray, temp);
    rec.gn = div_shared(rec.p - center, radius);
    check_face(rec, ray);
    if (EAGER_UV) rec.uv = sphere_uv(rec.gn);    // the render kernel derives it lazily from gn
    rec.material = sp[4];
    ry = temp;
    return true;
}

// uv of a point on square `index` (Square.hh: the two quotients of the accepted test).  The render kernels take it from here at
// shading time, and only when a texture consumes it: rec.p holds a and b exactly (set_comp below), so the value is the one the
// test would have stored -- and the two divisions leave the traversal loop.
TRC_DEV F2 square_uv(const SceneRef& S, uint32_t index, F3 p) {
    const uint32_t* sq = S.small_base + S.off_squares + index * kSquareDwords;
    const float4 rg = ld4(sq);
    const uint32_t axes = sq[6];
    const uint32_t axis_i = axes & 3u, axis_j = (axes >> 2) & 3u;
    F2 uv;
    uv.x = (comp(p, axis_i) - rg.x) / (rg.y - rg.x);
    uv.y = (comp(p, axis_j) - rg.z) / (rg.w - rg.z);
    return uv;
}
template <bool EAGER_UV = true>
TRC_DEV bool square_hit_test(const SceneRef& S, uint32_t index, const Ray& ray, float rx, float& ry, HitRec& rec) {
    const uint32_t* sq = S.small_base + S.off_squares + index * kSquareDwords;
    const float4 rg = ld4(sq);          // range_i.x, range_i.y, range_j.x, range_j.y
    const float4 kx = ld4(sq + 4);      // value_k, 1/area, axes, material
    const uint32_t axes = __float_as_uint(kx.z);
    const uint32_t axis_i = axes & 3u, axis_j = (axes >> 2) & 3u, axis_k = (axes >> 4) & 3u;
    const float value_k = kx.x;
    float t = (value_k - comp(ray.o, axis_k)) / comp(ray.d, axis_k);
    if (is_inf(t) || is_nan(t)) return false;
    if (t < rx || t > ry) return false;
    float a = comp(ray.o, axis_i) + t * comp(ray.d, axis_i);
    if (a < rg.x || a > rg.y) return false;
    float b = comp(ray.o, axis_j) + t * comp(ray.d, axis_j);
    if (b < rg.z || b > rg.w) return false;
    if (EAGER_UV) {
        rec.uv.x = (a - rg.x) / (rg.y - rg.x);
        rec.uv.y = (b - rg.z) / (rg.w - rg.z);
    }
    rec.t = t;
    F3 gn = f3(0);
    set_comp(gn, axis_k, 1.0f);
    rec.gn = gn;
    check_face(rec, ray);
    rec.gn = rec.sn;
    set_comp(rec.p, axis_k, value_k);
    set_comp(rec.p, axis_i, a);
    set_comp(rec.p, axis_j, b);
    ry = t;
    rec.PDF = kx.y;
    rec.material = __float_as_uint(kx.w);
    return true;
}

struct LightSample { F3 p, n; float areaPDF; uint32_t material; };
// Math.hh:57-74 (Ray Tracing Gems ch. 6)
TRC_DEV F3 offset_ray(const F3 p, const F3 n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_x = (int32_t)(int_scale * n.x), of_y = (int32_t)(int_scale * n.y), of_z = (int32_t)(int_scale * n.z);
    float pix = __int_as_float(__float_as_int(p.x) + ((p.x < 0) ? -of_x : of_x));
    float piy = __int_as_float(__float_as_int(p.y) + ((p.y < 0) ? -of_y : of_y));
    float piz = __int_as_float(__float_as_int(p.z) + ((p.z < 0) ? -of_z : of_z));
    return f3(fabsf(p.x) < origin ? p.x + float_scale * n.x : pix,
              fabsf(p.y) < origin ? p.y + float_scale * n.y : piy,
              fabsf(p.z) < origin ? p.z + float_scale * n.z : piz);
}
// Square.hh:40-58
TRC_DEV void square_sample(const SceneRef& S, uint32_t index, F2 u, F3 pos, LightSample& lsr) {
    const uint32_t* sq = S.small_base + S.off_squares + index * kSquareDwords;
    const float4 rg = ld4(sq);
    const float4 kx = ld4(sq + 4);
    const uint32_t axes = __float_as_uint(kx.z);
    const uint32_t axis_i = axes & 3u, axis_j = (axes >> 2) & 3u, axis_k = (axes >> 4) & 3u;
    F3 p = f3(0);
    set_comp(p, axis_k, kx.x);
    set_comp(p, axis_i, rg.x + u.x * (rg.y - rg.x));
    set_comp(p, axis_j, rg.z + u.y * (rg.w - rg.z));
    F3 n = f3(0);
    set_comp(n, axis_k, 1.0f);
    F3 w = normalize(pos - p);
    set_comp(n, axis_k, copysignf(1.0f, dot(w, n)));
    lsr.n = n;
    lsr.p = offset_ray(p, n);
    lsr.areaPDF = kx.y;
    lsr.material = __float_as_uint(kx.w);
}

template <bool STATS, bool VOL>
TRC_DEV bool cube_hit_test(const SceneRef& S, uint32_t index, const Ray& ray, float& ry, HitRec& rec, TravCounters& cnt) {
    const uint32_t* cb = S.small_base + S.off_cubes + index * kCubeDwords;
    // inverse matrix columns c0..c3 (xyz each)
    const float4 i0 = ld4(cb), i1 = ld4(cb + 4), i2 = ld4(cb + 8);
    const F3 ic0 = f3(i0.x, i0.y, i0.z), ic1 = f3(i0.w, i1.x, i1.y), ic2 = f3(i1.z, i1.w, i2.x), ic3 = f3(i2.y, i2.z, i2.w);
    F3 origin = ((ic0 * ray.o.x + ic1 * ray.o.y) + ic2 * ray.o.z) + ic3;
    F3 direction = (ic0 * ray.d.x + ic1 * ray.d.y) + ic2 * ray.d.z;
    direction = normalize(direction);                       // Ray(origin.xyz, direction.xyz), Cube.hh:23
    // layout: [0,12) inverse, [12,24) model, [24,33) normal c0..c2, [33,39) box min/max, [39] material
    const F3 mini = f3(__uint_as_float(cb[33]), __uint_as_float(cb[34]), __uint_as_float(cb[35]));
    const F3 maxi = f3(__uint_as_float(cb[36]), __uint_as_float(cb[37]), __uint_as_float(cb[38]));
    float t_obj; F3 gn_obj, p_obj; F2 uv;
    if (!box_hit_record(mini, maxi, origin, direction, ry, t_obj, gn_obj, p_obj, uv)) return false;
    if (STATS) cnt.hit_cube++;
    const float4 m0 = ld4(cb + 12), m1 = ld4(cb + 16), m2 = ld4(cb + 20);
    const F3 mc0 = f3(m0.x, m0.y, m0.z), mc1 = f3(m0.w, m1.x, m1.y), mc2 = f3(m1.z, m1.w, m2.x), mc3 = f3(m2.y, m2.z, m2.w);
    F3 p = ((mc0 * p_obj.x + mc1 * p_obj.y) + mc2 * p_obj.z) + mc3;
    float t = length(ray.o - p);                            // distance(ray.origin, p), Cube.hh:32
    if (t >= ry) return false;
    ry = t;
    if (VOL) { rec.vol_o = origin; rec.vol_d = direction; rec.vol_t = t_obj; rec.vol_cube = index; }
    const float4 n0 = ld4(cb + 24), n1 = ld4(cb + 28);
    const F3 nc0 = f3(n0.x, n0.y, n0.z), nc1 = f3(n0.w, n1.x, n1.y), nc2 = f3(n1.z, n1.w, __uint_as_float(cb[32]));
    rec.t = t;
    rec.p = p;
    rec.gn = normalize((nc0 * gn_obj.x + nc1 * gn_obj.y) + nc2 * gn_obj.z);
    check_face(rec, ray);
    rec.uv = uv;
    rec.material = cb[39];
    // rec.PDF keeps its previous value: the reference copies an uninitialised field here (Cube.hh:21,45)
    return true;
}

struct TriPos { float4 a, b, c; };       // the 48-byte position record of one triangle
TRC_DEV TriPos load_tripos(const SceneRef& S, uint32_t index) {
    const uint32_t* tp = S.blob + S.off_tripos + (size_t)index * kTriPosDwords;
    TriPos t; t.a = ld4_global(tp); t.b = ld4_global(tp + 4); t.c = ld4_global(tp + 8);
    return t;
}
template <bool STATS>
TRC_DEV bool triangle_hit_test(const SceneRef& S, uint32_t index, const TriPos& pos, const Ray& ray, float rx, float& ry, HitRec& rec, TravCounters& cnt) {
    const float4 a4 = pos.a, b4 = pos.b, c4 = pos.c;
    const F3 v0 = f3(a4.x, a4.y, a4.z), v1 = f3(b4.x, b4.y, b4.z), v2 = f3(c4.x, c4.y, c4.z);
    F3 v0v1 = v1 - v0;
    F3 v0v2 = v2 - v0;
    F3 pvec = cross(ray.d, v0v2);
    float det = dot(v0v1, pvec);
    if (fabsf(det) < FLT_EPSILON) return false;
    float invDet = 1 / det;
    F3 tvec = ray.o - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    F3 qvec = cross(tvec, v0v1);
    float v = dot(ray.d, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float w = 1.0f - u - v;
    float t = dot(v0v2, qvec) * invDet;
    if (t > ry || t < rx) return false;
    if (STATS) cnt.hit_triangle++;
    rec.p = (u * v1 + v * v2) + w * v0;
    ry = t;
    rec.t = t;
    const uint32_t* ta = S.blob + S.off_triattr + (size_t)index * kTriAttrDwords;
    const float4 q0 = ld4(ta), q1 = ld4(ta + 4), q2 = ld4(ta + 8), q3 = ld4(ta + 12);
    const F3 n0 = f3(q0.x, q0.y, q0.z), n1 = f3(q0.w, q1.x, q1.y), n2 = f3(q1.z, q1.w, q2.x);
    rec.gn = (u * n1 + v * n2) + w * n0;                    // unnormalised (B-5)
    rec.uv.x = (u * q2.w + v * q3.y) + w * q2.y;            // uv0 = q2.yz, uv1 = q2.w q3.x, uv2 = q3.yz
    rec.uv.y = (u * q3.x + v * q3.z) + w * q2.z;
    check_face(rec, ray);
    rec.material = 19;                                      // hard-coded, Triangle.hh:82
    return true;
}

// The record of a triangle test that is KNOWN to be accepted (the replay of a walk's winner, trav_build_record): the same
// expressions as triangle_hit_test above in the same order -- same bits -- without the reject exits, and with the position
// and the attribute record requested together: one memory round trip where test-then-fetch takes two.
TRC_DEV void triangle_record(const SceneRef& S, uint32_t index, const Ray& ray, HitRec& rec) {
    const uint32_t* tp = S.blob + S.off_tripos + (size_t)index * kTriPosDwords;
    const uint32_t* ta = S.blob + S.off_triattr + (size_t)index * kTriAttrDwords;
    const float4 a4 = ld4_global(tp), b4 = ld4_global(tp + 4), c4 = ld4_global(tp + 8);
    const float4 q0 = ld4_global(ta), q1 = ld4_global(ta + 4), q2 = ld4_global(ta + 8), q3 = ld4_global(ta + 12);
    const F3 v0 = f3(a4.x, a4.y, a4.z), v1 = f3(b4.x, b4.y, b4.z), v2 = f3(c4.x, c4.y, c4.z);
    F3 v0v1 = v1 - v0;
    F3 v0v2 = v2 - v0;
    F3 pvec = cross(ray.d, v0v2);
    float det = dot(v0v1, pvec);
    float invDet = 1 / det;
    F3 tvec = ray.o - v0;
    float u = dot(tvec, pvec) * invDet;
    F3 qvec = cross(tvec, v0v1);
    float v = dot(ray.d, qvec) * invDet;
    float w = 1.0f - u - v;
    float t = dot(v0v2, qvec) * invDet;
    rec.p = (u * v1 + v * v2) + w * v0;
    rec.t = t;
    const F3 n0 = f3(q0.x, q0.y, q0.z), n1 = f3(q0.w, q1.x, q1.y), n2 = f3(q1.z, q1.w, q2.x);
    rec.gn = (u * n1 + v * n2) + w * n0;                    // unnormalised (B-5)
    rec.uv.x = (u * q2.w + v * q3.y) + w * q2.y;
    rec.uv.y = (u * q3.x + v * q3.z) + w * q2.z;
    check_face(rec, ray);
    rec.material = 19;                                      // hard-coded, Triangle.hh:82
}

// ---------------------------------------------------------------- Scene::hit
// Resumable traversal: the walk of one ray is a small state (Trav) advanced by trav_iter(), one "descend until a
// leaf, test the leaf" round per call, so a kernel may interleave the rounds of its lanes with other work
// (k_render_sched).  scene_hit() below runs it to completion.
// `stack` is this lane's column of the workgroup stack (entry e at stack[e * kBlock]); `lvstack`
// (STATS only) holds the level of the node that deferred each entry.
struct Trav {
    uint32_t tag, sp;
    float ry;                 // closest accepted t so far (test_t while nothing was hit)
    int32_t level;            // STATS only: level of the interior node being expanded / parent level of a leaf
    // "the walk is over" is tag == kTagNone, not a flag of its own: a lane-divergent bool carried around the traversal loop costs
    // three scalar mask instructions at every join of the loop's control flow (a third of the loop's instructions were those)
    TRC_DEV bool is_done() const { return tag == kTagNone; }
    TRC_DEV void finish() { tag = kTagNone; }
    // DEFER (below): the last accepted test -- its tag and the range_t.y it was run against.  Together with `ry` (= its t)
    // that is all a walk has to carry: the record is a function of (ray, primitive, range_t.y) and is built after the walk.
    uint32_t win_tag;
    float win_ry;
};

// pops the next deferred sibling, or ends the traversal; `ret_start` = level at which the reference's first
// upward ("came from child") iteration would run -- only counted, never executed
// A lane's stack.  A ray rarely has more than a dozen siblings pending, but the stack must hold the tree's depth.  HYB (the render
// kernels on trees read from memory, whose registers allow more wavefronts than whole stacks in LDS would): the first S.stack_lds
// entries live in LDS, deeper ones in the wavefront's rows in global memory.  The other kernels keep the whole stack in LDS.
// Almost every access of a wavefront is below stack_lds in all of its lanes: ONE wave-uniform test (a ballot) keeps the per-lane
// if / else -- six scalar mask instructions and two branches per access -- out of the box-step loop.
// (Round 6 tried the LDS entries as the TOP of the stack instead -- a ring of 2^k slots that moves its oldest entry to the global row
// when it is full: profiles/r06/exp_stack_ring.patch.  Nothing at 8 entries, -14 % at 4, and traceMIS loses 5 %: a lane that is deep
// keeps pushing and popping around the ring's edge, and every such step then costs an eviction AND a refill.)
template <bool HYB>
TRC_DEV void stack_push(const SceneRef& S, uint32_t* stack, uint32_t& sp, uint32_t v) {
    if (!HYB || __builtin_expect(__builtin_amdgcn_ballot_w64(sp >= S.stack_lds) == 0ull, 1) || sp < S.stack_lds) stack[sp * kBlock] = v;
    else st1_global(S.ovf + (sp - S.stack_lds) * kBlock, v);
    sp++;
}
template <bool HYB>
TRC_DEV uint32_t stack_pop(const SceneRef& S, const uint32_t* stack, uint32_t& sp) {      // sp > 0
    sp--;
    if (!HYB || __builtin_expect(__builtin_amdgcn_ballot_w64(sp >= S.stack_lds) == 0ull, 1) || sp < S.stack_lds) return stack[sp * kBlock];
    return ld1_global(S.ovf + (sp - S.stack_lds) * kBlock);
}

template <bool HYB, bool STATS>
TRC_DEV void trav_pop_or_finish(const SceneRef& S, Trav& tv, int32_t ret_start, const uint32_t* stack, const uint32_t* lvstack, TravCounters& cnt) {
    if (tv.sp == 0) {
        if (STATS) cnt.n_return += (uint32_t)(ret_start + 1);
        tv.finish();
        return;
    }
    tv.tag = stack_pop<HYB>(S, stack, tv.sp);
    if (STATS) {
        const int32_t ls = (int32_t)lvstack[tv.sp * kBlock];
        cnt.n_return += (uint32_t)(ret_start - ls + 1);
        tv.level = ls;
        if ((tv.tag >> kTagIndexBits) == kTagInterior) tv.level += 1;
    }
}

// Render.hh:135-153: counts the ray, rejects non-finite rays and rays that miss the root box.  Returns false
// (the walk over: tv.is_done()) when the traversal is over before it starts.
template <bool STATS>
TRC_DEV bool trav_begin(const F3 root_min, const F3 root_max, const Ray& ray, const float test_t, Trav& tv, TravCounters& cnt) {
    if (STATS) cnt.rays++;
    tv.tag = kTagNone;
    tv.sp = 0;
    tv.level = 0;
    tv.ry = test_t;
    tv.win_tag = kTagNone;
    tv.win_ry = test_t;
    // a ray with a NaN / infinite component misses the scene (same rule as oracle/oracle.cpp Scene::hit: B-4/B-10);
    // without it NaN-ignoring min/max make such a ray "hit" every box of the tree
    const float finite_probe = fabsf(ray.o.x) + fabsf(ray.o.y) + fabsf(ray.o.z) + fabsf(ray.d.x) + fabsf(ray.d.y) + fabsf(ray.d.z);
    if (!(finite_probe < __builtin_inff())) return false;
    if (!box_hit(root_min, root_max, ray, FLT_MIN, test_t)) return false;
    tv.tag = kTagInterior << kTagIndexBits;   // root
    return true;
}

// leaf test of one primitive tag; returns true when the hit was accepted (tv.ry lowered).
//
// DEFER (production closest-hit walks): "test now, build the record for the winner afterwards".  The reference writes
// the whole HitRecord at every accepted test (15 dwords that then stay live across the latency-bound loop: the mesh
// kernels spilled them).  Every field of an accepted test's record is a function of (ray, primitive, the range_t.y the
// test ran against), and the next accepted test overwrites all of them -- with two exceptions that are kept live: PDF
// is written by squares only (Square.hh:110; a cube copies an uninitialised one, Cube.hh:21,45), so a later sphere / cube /
// triangle hit still carries the last accepted square's, and the object-space fields of traceVolume are written by cubes
// only (Cube.hh:30-31,39).  So the walk carries t (tv.ry), the winner's tag and the range it saw, writes through those two
// groups when their primitive type is accepted, and trav_build_record() replays the winner once after the walk: same
// function, same inputs, same bits.
// DEFER = 2: everything but the cube, whose test is the most expensive one to replay (the LDS-resident kernels, where
// instructions and not registers are what is short): an accepted cube writes the record at once as before, and the replay
// after the walk leaves such a winner's record alone.
template <bool STATS, bool EAGER_UV, bool VOL, int DEFER = 0>
TRC_DEV bool trav_test_leaf(const SceneRef& S, const Ray& ray, HitRec& rec, Trav& tv, uint32_t tag, TravCounters& cnt) {
    const float rx = FLT_MIN;
    const uint32_t type = tag >> kTagIndexBits, index = tag & kTagIndexMask;
    bool ok;
    HitRec scratch;                          // DEFER: what the tests write here is read by nobody (dead code), except ...
    HitRec& out = DEFER ? scratch : rec;
    const float ry_seen = tv.ry;
    if (type == 1u) {
        if (STATS) cnt.leaf[1]++;
        ProfScope<STATS> scope(cnt, kProfSquare);
        ok = square_hit_test<EAGER_UV>(S, index, ray, rx, tv.ry, out);
        if (DEFER && ok) rec.PDF = scratch.PDF;                                   // ... the fields only ONE type writes
    } else if (type == 0u) {
        if (STATS) cnt.leaf[0]++;
        ProfScope<STATS> scope(cnt, kProfSphere);
        ok = sphere_hit_test<EAGER_UV>(S, index, ray, rx, tv.ry, out);
    } else if (type == 2u) {
        if (STATS) cnt.leaf[2]++;
        ProfScope<STATS> scope(cnt, kProfCube);
        ok = cube_hit_test<STATS, VOL>(S, index, ray, tv.ry, DEFER == 2 ? rec : out, cnt);
        if (DEFER == 1 && VOL && ok) { rec.vol_o = scratch.vol_o; rec.vol_d = scratch.vol_d; rec.vol_t = scratch.vol_t; rec.vol_cube = scratch.vol_cube; }
    } else {
        if (STATS) cnt.leaf[3]++;
        ProfScope<STATS> scope(cnt, kProfTriangle);
        ok = triangle_hit_test<STATS>(S, index, load_tripos(S, index), ray, rx, tv.ry, out, cnt);
    }
    if (ok) {
        if (DEFER) { tv.win_tag = tag; tv.win_ry = ry_seen; }
        else rec.tag = tag;
    }
    return ok;
}
// DEFER: the record of the walk's winner, once.  A sphere under the lazy-uv rule (the render kernels) gets uv = 0 where the
// eager walk leaves a stale value: hit_color() never reads a sphere's uv (it derives it from gn), so nobody can tell, and
// the two dwords are not carried through the walk.
template <bool EAGER_UV, bool VOL, int DEFER>
TRC_DEV void trav_build_record(const SceneRef& S, const Ray& ray, HitRec& rec, const Trav& tv) {
    const uint32_t type = tv.win_tag >> kTagIndexBits, index = tv.win_tag & kTagIndexMask;
    TravCounters nocount;
    float ry = tv.win_ry;
    if (type == 3u) triangle_record(S, index, ray, rec);
    else if (type == 1u) { square_hit_test<EAGER_UV>(S, index, ray, FLT_MIN, ry, rec); if (!EAGER_UV) { rec.uv.x = 0; rec.uv.y = 0; } }
    else if (type == 0u) { sphere_hit_test<EAGER_UV>(S, index, ray, FLT_MIN, ry, rec); if (!EAGER_UV) { rec.uv.x = 0; rec.uv.y = 0; } }
    else if (DEFER != 2) cube_hit_test<false, VOL>(S, index, ray, ry, rec, nocount);
    rec.tag = tv.win_tag;
}

// one round: (1) expand interior nodes until this lane holds a leaf (or runs out of work) -- the whole wavefront
// does box tests here; (2) test the leaf.  Lanes whose walk is over idle through the call.
//
// Every lane performs exactly ITS OWN sequence of the reference's steps (box tests against its own running closest
// hit, near child first, far child deferred without a re-test, primitive tests in that order); what a production
// kernel is free to choose is the INTERLEAVING across the lanes of a wavefront:
//   * the instrumented kernels descend until every lane holds a leaf (and take the exact counters);
//   * production kernels on a tree read from global memory leave the descent as soon as fewer than
//     TRC_DESCEND_MIN_GLOBAL lanes are still descending while others wait with a leaf: a few lanes of a mesh ray
//     batch walk 50-300 nodes while their neighbours need 10, and a dependent L2 round trip per step makes waiting
//     for them the dominant cost (config 3: 77.8 -> 70.7 ms, config 4: 40.9 -> 35.7 ms with 8; 2 / 4 / 6 / 12 / 16 /
//     24 / 40 measured: tools/ab_bench.py, docs/HISTORY.md 4.1).  The stragglers resume in the next round;
//   * on an LDS-resident tree the plain round is the fastest (22.7 ms; thresholds 2..40: 24.0-25.5 ms).
// Round 1 shipped a SPECULATIVE round instead (Aila & Laine 2009: a lane that reaches a leaf early postpones it and keeps
// descending against its not-yet-updated closest hit).  That visits a superset of the reference's boxes and can test a
// primitive the reference culls -- harmless only if a primitive's computed t is never below its box's computed entry
// t, which does not hold (MakeSphere makes spheres LARGER than their boxes, Tracer.mm:165-172;
// tests/test_gpu_traversal.py constructs the case and the speculative round fails it).  A checked form (snapshot +
// rollback when the postponed test accepts) is exact but slower than the plain round (measured: 23.9 / 75.5 / 39.3 ms
// against 22.6 / 77.8 / 40.9 plain and 22.7 / 70.8 / 35.7 for the threshold round); the unchecked one survives only as
// profiles/r05/exp_removed_variants.patch, the build that must FAIL the adversarial test.
template <bool ALL_LDS, bool STATS, bool ANY, bool EAGER_UV, bool VOL = false, bool HYB = false, int DEFER = 0>
TRC_DEV void trav_iter(const SceneRef& S, const Ray& ray, HitRec& rec, const float test_t, Trav& tv,
                       uint32_t* stack, uint32_t* lvstack, TravCounters& cnt) {
    const float rx = FLT_MIN;
#ifndef TRC_DESCEND_MIN_LDS
#define TRC_DESCEND_MIN_LDS 1
#endif
    // LDS-resident trees: the plain round (a compile-time 1); trees read from memory: the scene's threshold, wave-uniform
    const uint32_t kDescendMin = ALL_LDS ? (uint32_t)TRC_DESCEND_MIN_LDS : S.descend_min;
    for (;;) {
        const bool interior = (tv.tag >> kTagIndexBits) == kTagInterior;
        if (!STATS && (!ALL_LDS || TRC_DESCEND_MIN_LDS > 1)) {
            const unsigned long long m = __ballot(interior);
            if (m == 0ull) break;
            if ((uint32_t)__popcll(m) < kDescendMin && __ballot((tv.tag >> kTagIndexBits) < kTagInterior) != 0ull) break;      // somebody waits with a leaf
            if (!interior) continue;
        } else if (!interior) break;
        float4 q0, q1, q2, q3;
        load_node<ALL_LDS>(S, tv.tag & kTagIndexMask, q0, q1, q2, q3);
        if (STATS) cnt.n_descend++;
        ProfScope<STATS> scope(cnt, kProfDescend);
        float t_left = tv.ry, t_right = tv.ry;
        const bool left_test = box_hit_t(f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), ray, rx, tv.ry, t_left);
        const bool right_test = box_hit_t(f3(q1.z, q1.w, q2.x), f3(q2.y, q2.z, q2.w), ray, rx, tv.ry, t_right);
        if (left_test || right_test) {
            const uint32_t tagL = __float_as_uint(q3.z), tagR = __float_as_uint(q3.w);
            const bool left_first = t_left < t_right;            // Render.hh:174 (literal, also when only one hit)
            if (left_test && right_test) {
                if (STATS) lvstack[tv.sp * kBlock] = (uint32_t)tv.level;
                stack_push<HYB>(S, stack, tv.sp, left_first ? tagR : tagL);    // Render.hh:171-172: visit the other one later
            }
            tv.tag = left_first ? tagL : tagR;
            if (STATS && (tv.tag >> kTagIndexBits) == kTagInterior) tv.level += 1;
        } else {
            trav_pop_or_finish<HYB, STATS>(S, tv, tv.level - 1, stack, lvstack, cnt);
        }
    }
    if ((tv.tag >> kTagIndexBits) < kTagInterior) {
        trav_test_leaf<STATS, EAGER_UV, VOL, DEFER>(S, ray, rec, tv, tv.tag, cnt);
        if (ANY && tv.ry < test_t) tv.finish();                       // Render.hh:244
        else trav_pop_or_finish<HYB, STATS>(S, tv, tv.level, stack, lvstack, cnt);
    }
}

template <bool ALL_LDS, bool STATS, bool ANY, bool EAGER_UV, bool VOL = false, bool HYB = false, int DEFER = 0>
TRC_DEV bool scene_hit(const SceneRef& S, const F3 root_min, const F3 root_max, const Ray& ray, HitRec& rec, const float test_t,
                       uint32_t* stack, uint32_t* lvstack, TravCounters& cnt) {
    static_assert(!DEFER || (!STATS && !ANY), "the deferred record is the production closest-hit walk's");
    Trav tv;
    if (!trav_begin<STATS>(root_min, root_max, ray, test_t, tv, cnt)) return false;
    while (!tv.is_done()) trav_iter<ALL_LDS, STATS, ANY, EAGER_UV, VOL, HYB, DEFER>(S, ray, rec, test_t, tv, stack, lvstack, cnt);
    const bool hit = tv.ry < test_t;
    // a miss ends the path (the record is re-initialised before anybody reads it), so only a hit is materialised
    if (DEFER && hit) trav_build_record<EAGER_UV, VOL, DEFER>(S, ray, rec, tv);
    return hit;
}

// ---------------------------------------------------------------- any-hit, order-free (production kernels)
// A shadow ray only asks WHETHER something lies in (FLT_MIN, test_t): Scene::hit returns at the first accepted hit
// (Render.hh:244) and nothing lowers range_t.y before that, so every box and primitive of an any-hit walk is tested
// against the same fixed range -- the set of boxes that pass and the answer do not depend on the visiting order (which
// primitive is found first does; no caller of the any-hit form reads the record: Render.metal:335-337) -- with one
// exception that is kept, the pick between the children when only one box passes (below).  The
// production kernels therefore walk shadow rays in whatever order hides latency best: left-first, so that the shadow rays of
// a wavefront walk together; no hit_t, no near / far ordering, no HitRecord.  (Two subtrees in hand per lane -- two independent
// node fetches in flight, half the dependent round trips -- cost the MIS kernels 31 spilled registers and measured slower; nearest
// child first measured the same: profiles/r05/exp_removed_variants.patch.)  The instrumented kernels keep the reference's walk
// (their counters are defined on it), and tests/test_gpu_traversal.py compares the two on the adversarial batches.
// Boxes that pass beyond the one in hand go to the lane's stack, one entry per level at most -- the depth it is sized for.
// Two corners of the reference's arithmetic are kept.  A square or a triangle met at EXACTLY t == test_t is "accepted" by
// its hit_test (Square.hh / Triangle.hh reject only t > range_t.y) but leaves range_t.y where it was, so Scene::hit does
// not report it (range_t.y < test_t, Render.hh:244,250): not an occluder here either.  And an accepted test whose t is
// NaN (overflowing coordinates) makes range_t.y NaN, after which every later test of the reference's walk passes and
// the answer does depend on the order: such a ray is handed back to the reference's walk (return value 2).
enum : uint32_t { kOccludedNo = 0u, kOccludedYes = 1u, kOccludedAskReference = 2u };
template <bool ALL_LDS, bool VOL, bool HYB>
TRC_DEV uint32_t scene_occluded_free(const SceneRef& S, const F3 root_min, const F3 root_max, const Ray& ray, const float test_t,
                                     uint32_t* stack, const uint32_t stack_cap) {
    Trav tb;
    TravCounters nocount;
    if (!trav_begin<false>(root_min, root_max, ray, test_t, tb, nocount)) return kOccludedNo;
    const float rx = FLT_MIN;
    constexpr uint32_t kRoot = kTagInterior << kTagIndexBits;
    uint32_t cur0 = kRoot, sp = 0;
    bool found = false, ask_reference = false;
    HitRec rec;                                   // written by the primitive tests, read by nobody
    hit_init(rec);
    auto is_interior = [](uint32_t tag) { return tag != kTagNone && (tag >> kTagIndexBits) == kTagInterior; };
    auto is_leaf = [](uint32_t tag) { return tag != kTagNone && (tag >> kTagIndexBits) != kTagInterior; };
    bool restarted = false;                       // this step ran out of stack: what it still finds belongs to the abandoned walk
    auto give = [&](uint32_t tag) {               // a box that passed: into the hand, else onto the stack
        if (restarted) return;
        if (cur0 == kTagNone) cur0 = tag;
        else if (sp < stack_cap) stack_push<HYB>(S, stack, sp, tag);
        // cannot happen on a tree of the depth the stack is sized for (one pending entry per level at most); if it does -- a tree
        // uploaded with a wrong depth -- an identical rewalk would fill the stack again and spin: the reference's walk answers
        else { restarted = true; sp = 0; cur0 = kTagNone; found = true; ask_reference = true; }
    };
#ifndef TRC_OCCL_DESCEND_MIN
#define TRC_OCCL_DESCEND_MIN 1      // plain round: shadow rays of a wavefront walk left-first, i.e. together (1 / 4 / 8 / 16 / 32: 50.9 / 50.9 / 51.3 / 52.1 / 53.1 ms on config 3)
#endif
    constexpr int kDescendMin = ALL_LDS ? TRC_DESCEND_MIN_LDS : TRC_OCCL_DESCEND_MIN;
    for (;;) {
        // ---- box steps: every lane with an interior node in hand expands it
        for (;;) {
            const bool i0 = is_interior(cur0);
            const unsigned long long m = __ballot(i0);
            if (m == 0ull) break;
            if (__popcll(m) < kDescendMin && __ballot(is_leaf(cur0)) != 0ull) break;
            if (!i0) continue;
            float4 a0, a1, a2, a3;
            const uint32_t t0 = cur0;
            restarted = false;
            load_node<ALL_LDS>(S, t0 & kTagIndexMask, a0, a1, a2, a3);
            cur0 = kTagNone;
            // both children pass: both are walked, in any order.  ONE passes: the reference still picks by
            // (t_left < t_right) with the other side's t left at range_t.y (Render.hh:161-174) -- a lone child entered at
            // exactly t == range_t.y loses to its sibling, whose box did NOT pass, and is never visited.  Kept literally.
            auto expand = [&](const float4& q0, const float4& q1, const float4& q2, const float4& q3) {
                float t_left = test_t, t_right = test_t;
                const bool l = box_hit_t(f3(q0.x, q0.y, q0.z), f3(q0.w, q1.x, q1.y), ray, rx, test_t, t_left);
                const bool r = box_hit_t(f3(q1.z, q1.w, q2.x), f3(q2.y, q2.z, q2.w), ray, rx, test_t, t_right);
                const uint32_t tagL = __float_as_uint(q3.z), tagR = __float_as_uint(q3.w);
                if (l && r) { give(tagL); give(tagR); }
                else if (l || r) give(t_left < t_right ? tagL : tagR);
            };
            expand(a0, a1, a2, a3);
            // refill the hand from the stack (deepest pending subtree first)
            if (restarted) continue;
            if (cur0 == kTagNone && sp > 0u) cur0 = stack_pop<HYB>(S, stack, sp);
        }
        // ---- primitive test: the leaf in hand
        if (!found && is_leaf(cur0)) {
            Trav tv;
            tv.ry = test_t;
            if (trav_test_leaf<false, false, VOL>(S, ray, rec, tv, cur0, nocount)) {
                if (tv.ry < test_t) found = true;                 // Render.hh:244
                else if (!(tv.ry == test_t)) { found = true; ask_reference = true; }      // NaN: the order matters from here on
            }
            cur0 = kTagNone;
        }
        if (found) { cur0 = kTagNone; sp = 0; }
        if (cur0 == kTagNone && sp > 0u) cur0 = stack_pop<HYB>(S, stack, sp);
        if (__ballot(cur0 != kTagNone) == 0ull) break;
    }
    return ask_reference ? kOccludedAskReference : (found ? kOccludedYes : kOccludedNo);
}
// the any-hit Scene::hit of the production kernels: order-free, the reference's walk for the rays that need it
template <bool ALL_LDS, bool VOL, bool HYB>
TRC_DEV bool scene_occluded(const SceneRef& S, const F3 root_min, const F3 root_max, const Ray& ray, const float test_t,
                            uint32_t* stack, const uint32_t stack_cap) {
    const uint32_t r = scene_occluded_free<ALL_LDS, VOL, HYB>(S, root_min, root_max, ray, test_t, stack, stack_cap);
    if (r != kOccludedAskReference) return r == kOccludedYes;
    HitRec rec;
    hit_init(rec);
    TravCounters nocount;
    return scene_hit<ALL_LDS, false, true, false, VOL, HYB>(S, root_min, root_max, ray, rec, test_t, stack, nullptr, nocount);
}

}  // namespace trcdev

// trc_sppm.hip -- the SPPM photon pass on gfx950 (BASELINE config 5).
//
// Reference: RT_Metal/Metal/Photon.metal:3-623, Photon.hh:12-89; host sequencing
// RT_Metal/Tracer/AAPLRenderer.mm:860-1086 (photon: = [photonPrepare on frame 0] + photonWork).
//   kernelCameraRecording   :96-167   -> k_sppm_camera   (+ the AABB of the valid records via ordered-key atomics,
//   kernelCameraReducing    :169-218     replacing the multi-pass ping-pong tree reduce; min/max are exact)
//   kernelPhotonParams      :357-372  -> k_sppm_params   (1 lane, like the reference)
//   kernelPhotonRadius      :374-384  -> k_sppm_radius
//   kernelPhotonRecording   :286-355  -> k_sppm_photon
//   kernelPhotonHashing + PhotonMarkVS/FS point raster :386-456 -> k_sppm_hash: the raster pass becomes
//       atomicMax(photon index) for the mark (Metal resolves same-pixel writes in primitive order: the LAST
//       point = highest index wins) and atomicAdd for the additively blended count
//   kernelPhotonSumming     :458-496  -> k_sppm_table   (the frame's photon sum + the per-cell gather table)
//   kernelPhotonRefine      :498-623  -> k_sppm_refine  (one 16-byte table read per hash cell, four cells in flight; 32 more for a photon in range)
// All launches go to the context stream in order; nothing synchronises with the host.
#include "trc_ctx.hpp"

#include <cstdlib>
#include <cstring>
#include <new>

namespace {

constexpr uint32_t kHashN = TRC_PHOTON_HASHN;
constexpr uint32_t kCellsB = kHashN * kHashN + 1u;      // gather table: float4 index of array B (after the kHashN^2 + 1 entries of A)

struct DComplex {           // device-side Complex (Camera.hh:27-55) + the bound keys of the camera reduce
    float box_min[3], box_max[3], box_size[3];
    float initial_radius, hash_scale, total_photon_sum;
    uint32_t frame_photon_sum;
    uint32_t key_min[3], key_max[3];
};

// The visible points (CameraRecord, Photon.hh:30-53: 112-byte records in the reference) as planes, one float4 / uint per
// pixel: a wavefront's access to one field is 8 rows x 128 contiguous bytes instead of 64 x 12 bytes scattered over 7 KB,
// and a pass touches only the planes it needs -- the refine reads three or one and rewrites flux | radius and the count.
// The four planes the camera pass writes exist TWICE: the pass of frame k + 1 reads copy A and writes copy B (a field it
// does not assign is carried over) while the refine of frame k still reads A, so it can run beside frame k on its own
// stream (trc_sppm_frames).  trc_sppm_download packs the current copy back into the reference's records.
struct VisiblePoints {
    float4* ratio_valid;     // ratio.xyz | valid (bits of a uint: 0 / 1)
    float4* position;
    float4* direction;
    float4* alternative;
    float4* flux_radius;     // flux.xyz | radius
    uint32_t* count;         // photonCount
};

struct KSppm {
    KScene ks;
    DCamera cam;
    float ambient[3];
    const float* env_rgb; uint32_t env_w, env_h;
    uint32_t W, H, frame_count, photon_first;   // photon_first: first photon index of this rank's range
    const uint32_t* tiles;
    uint32_t* canvas_rng;
    float* accum;
    uint32_t* photon_rng;
    VisiblePoints vp;        // the visible points the pass writes (camera) / works on (refine)
    VisiblePoints vp_prev;   // camera pass: the copy it takes unassigned fields from (== vp on the first frame)
    trc_PhotonRecord* pho_rec;
    uint32_t* mark;          // winning photon index + 1 per cell, 0 = empty
    uint32_t* count;
    const float4* cells;     // gather table of k_sppm_table: array A (1 float4 per hash cell + 1 for out-of-range reads), then array B (2 each)
    DComplex* cx;
    unsigned long long* stats;   // ctx counters: rays += Scene::hit calls of the camera and photon passes
};

constexpr uint32_t kPhotonSlack = 64u * kBlock;      // records past the 512 x 512 photons in the buffers the ranks all-gather into: an uneven split pads its last chunks
// order-preserving float <-> uint mapping for atomicMin/atomicMax
__device__ __forceinline__ uint32_t f2key(float f) { uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float key2f(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k); }

__device__ __forceinline__ F3 ld3(const trc_float3& v) { return f3(v.x, v.y, v.z); }
__device__ __forceinline__ void st3(trc_float3& d, F3 v) { d.x = v.x; d.y = v.y; d.z = v.z; d._pad = 0.0f; }

// Render.hh:96-120 (same word swap as kernelPathTracing, B-1)
__device__ __forceinline__ Pcg to_rng(uint4 t) { Pcg r; r.state = ((uint64_t)t.z << 32) | t.w; r.inc = ((uint64_t)t.x << 32) | t.y; return r; }
__device__ __forceinline__ uint4 ex_rng(const Pcg& r) {
    uint4 t; t.x = (uint32_t)(r.state >> 32); t.y = (uint32_t)r.state; t.z = (uint32_t)(r.inc >> 32); t.w = (uint32_t)r.inc; return t;
}

// a / b for a divisor that many quotients share (the hash constants, the frame's hash scale).  The compiler's binary32
// division is v_div_scale x 2, v_rcp_f32, a Newton step on the reciprocal, a * y, two residual corrections,
// v_div_fixup: 10-11 instructions.  The reciprocal and its Newton step depend on b alone, so they are taken once
// (DivBy) and a quotient costs the multiply and the SAME two fused corrections: 5 instructions, the same bits.  What is
// left out is only the operand scaling / special-case fix-up, which matters for denormal or near-overflow operands,
// infinities and NaNs; the hash works on cell indices (< 2^24), moduli (< 2^23) and the hash scale (finite, > 0).
struct DivBy { float b, y; };
__device__ __forceinline__ DivBy div_by(float b) {
    DivBy d; d.b = b;
    const float y0 = __builtin_amdgcn_rcpf(b);
    d.y = __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
    return d;
}
__device__ __forceinline__ float operator/(float a, const DivBy d) {
    float q = a * d.y;
    q = __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.y, q);
    return __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.y, q);
}
__device__ __forceinline__ F3 operator/(F3 a, const DivBy d) { return f3(a.x / d, a.y / d, a.z / d); }
struct HashDiv { DivBy scale, q[4], m[4]; };           // the divisors of ph_hash
__device__ __forceinline__ HashDiv hash_div(float HashScale) {
    HashDiv h;
    h.scale = div_by(HashScale);
    const float q[4] = {1225.0f, 1585.0f, 2457.0f, 2098.0f}, m[4] = {4194287.0f, 4194277.0f, 4194191.0f, 4194167.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) { h.q[k] = div_by(q[k]); h.m[k] = div_by(m[k]); }
    return h;
}

// Photon.hh:57-89
__device__ __forceinline__ float ph_mod(float x, float y) { return x - y * floorf(x / y); }
__device__ __forceinline__ float ph_hash(const F3 idx, const HashDiv& hd, const float BufInfo) {
    const float HashNum = BufInfo * BufInfo;
    const float n[4] = {idx.x, idx.y, idx.z, idx.x + idx.y - idx.z};
    const float q[4] = {1225.0f, 1585.0f, 2457.0f, 2098.0f};
    const float r[4] = {1112.0f, 367.0f, 92.0f, 265.0f};
    const float a[4] = {3423.0f, 2646.0f, 1707.0f, 1999.0f};
    const float m[4] = {4194287.0f, 4194277.0f, 4194191.0f, 4194167.0f};
    float nm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float nk = n[k] * 4194304.0f / hd.scale;
        float beta = floorf(nk / hd.q[k]);
        float pk = a[k] * (nk - beta * q[k]) - beta * r[k];
        float sgn = (-pk > 0.0f) ? 1.0f : ((-pk < 0.0f) ? -1.0f : 0.0f);
        beta = (sgn + 1.0f) * 0.5f * m[k];
        nk = pk + beta;
        nm[k] = nk / hd.m[k];
    }
    float d = ((nm[0] * 1.0f + nm[1] * -1.0f) + nm[2] * 1.0f) + nm[3] * -1.0f;
    float fr = d - floorf(d);
    return floorf(fr * HashNum);
}
__device__ __forceinline__ F3 uniform_sample_hemisphere(F2 u) {     // Sampling.hh:55-60
    float z = u.x;
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float phi = 2 * kPi * u.y;
    float s, c;
    dm_sincosf(phi, &s, &c);
    return f3(r * c, r * s, z);
}

struct SppmCtx {
    SceneRef S;
    F3 root_min, root_max;
    Shade sh;
    F3 ambient;
    EnvMap env;
    uint32_t* stack;
    uint32_t* lvstack;
};
__device__ __forceinline__ SppmCtx make_sppm_ctx(const KSppm& kp, const uint32_t* small_base) {
    const DScene& sc = kp.ks.sc;
    SppmCtx cx;
    cx.S = make_scene_ref(sc, small_base);
    cx.root_min = f3(kp.ks.root_box[0], kp.ks.root_box[1], kp.ks.root_box[2]);
    cx.root_max = f3(kp.ks.root_box[3], kp.ks.root_box[4], kp.ks.root_box[5]);
    cx.sh.mats = small_base + sc.off_materials;
    cx.ambient = f3(kp.ambient[0], kp.ambient[1], kp.ambient[2]);
    cx.env.rgb = kp.env_rgb; cx.env.w = kp.env_w; cx.env.h = kp.env_h;
    cx.stack = lane_stack(sc);
    cx.lvstack = lane_lvstack(sc);
    return cx;
}
template <bool ALL_LDS>
__device__ __forceinline__ bool sppm_hit(const SppmCtx& cx, const Ray& ray, HitRec& rec, uint32_t& n_rays) {
    n_rays++;
    TravCounters cnt;    // dead in non-instrumented instantiations
    return scene_hit<ALL_LDS, false, false, false>(cx.S, cx.root_min, cx.root_max, ray, rec, FLT_MAX, cx.stack, cx.lvstack, cnt);
}

// kernelCameraRecording, Photon.metal:96-167 + traceCameraRecord :3-92
#ifndef TRC_SPPM_CAMERA_WAVES
#define TRC_SPPM_CAMERA_WAVES 4
#endif
#ifndef TRC_SPPM_PHOTON_WAVES
#define TRC_SPPM_PHOTON_WAVES 5
#endif
#ifndef TRC_SPPM_REFINE_WAVES
#define TRC_SPPM_REFINE_WAVES 5
#endif
template <bool ALL_LDS>
__global__ void __launch_bounds__(kBlock, TRC_SPPM_CAMERA_WAVES) k_sppm_camera(const KSppm kp) {
    const uint32_t* small_base = stage_scene(kp.ks.sc);
    const SppmCtx cx = make_sppm_ctx(kp, small_base);
    const uint32_t tile = kp.tiles[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t px = (tile & 0xFFFFu) * 8u + (lane & 7u);
    const uint32_t py = (tile >> 16) * 8u + (lane >> 3);
    uint32_t n_rays = 0;
    if (px < kp.W && py < kp.H) {
    const size_t pix = (size_t)py * kp.W + px;

    Pcg rng = to_rng(reinterpret_cast<const uint4*>(kp.canvas_rng)[pix]);
    const float u = (float)px / (float)kp.W, v = (float)py / (float)kp.H;
    Ray ray = cast_ray(kp.cam, u, v, rng);

    // cr.reset() on the first frame (Photon.hh:42-52; `alternative` is NOT reset); a field of the record the pass does not
    // assign keeps its value: it is simply not stored
    const bool first = kp.frame_count == 0;
    int depth = first ? 3 : 8;
    bool valid = false, alt_set = false;
    F3 cr_ratio = f3(1), cr_position = f3(0), cr_direction = f3(0), cr_alternative = f3(0);
    {
        HitRec rec;
        hit_init(rec);
        F3 ratio = f3(1.0f);
        bool hitted = sppm_hit<ALL_LDS>(cx, ray, rec, n_rays);
        bool finished = false;
        do {
            if (!hitted) { cr_alternative = ratio * env_radiance(cx.env, cx.ambient, ray.d); alt_set = true; finished = true; break; }
            const int mtype = mat_type(cx.sh, rec.material);
            if (mtype == kMatDiffuse) {
                F3 le = mat_albedo(cx.sh, rec.material);
                float w = dot(-ray.d, -rec.gn);
                cr_alternative = ratio * le * fabsf(w); alt_set = true;
                finished = true; break;
            }
            if (!mat_specular(cx.sh, rec.material)) {
                valid = true; cr_ratio = ratio; cr_position = rec.p; cr_direction = ray.d;
                finished = true; break;
            }
            F3 nx, ny;
            coordinate_system(rec.sn, nx, ny);
            F3 wi = f3(0); float bxPDF = 0;
            F3 minus_d = -ray.d;
            F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
            F2 uu; uu.x = pcg_float(rng); uu.y = pcg_float(rng);
            F3 attenuation = material_S_F(mtype, hit_color(cx.S, cx.sh, rec), wo, wi, uu, bxPDF);
            if (bxPDF <= 0) break;
            F3 pn = rec.sn * copysignf(1.0f, wi.z);
            F3 _origin = offset_ray(rec.p, pn);
            ray = make_ray(_origin, (nx * wi.x + ny * wi.y) + rec.sn * wi.z);
            ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
            if (is_inf(ratio.x) || is_inf(ratio.y) || is_inf(ratio.z) || is_nan(ratio.x) || is_nan(ratio.y) || is_nan(ratio.z)) ratio = f3(1.0f);
            hitted = sppm_hit<ALL_LDS>(cx, ray, rec, n_rays);
        } while ((--depth) > 0);
        if (!finished) { cr_alternative = f3(0); alt_set = true; }
    }
    if (valid || first) {
        kp.vp.ratio_valid[pix] = make_float4(cr_ratio.x, cr_ratio.y, cr_ratio.z, __uint_as_float(valid ? 1u : 0u));
        kp.vp.position[pix] = make_float4(cr_position.x, cr_position.y, cr_position.z, 0.0f);
        kp.vp.direction[pix] = make_float4(cr_direction.x, cr_direction.y, cr_direction.z, 0.0f);
    } else {                                    // ratio, position, direction keep their values; valid = 0
        float4 rv = kp.vp_prev.ratio_valid[pix];
        rv.w = __uint_as_float(0u);
        kp.vp.ratio_valid[pix] = rv;
        kp.vp.position[pix] = kp.vp_prev.position[pix];
        kp.vp.direction[pix] = kp.vp_prev.direction[pix];
    }
    kp.vp.alternative[pix] = alt_set ? make_float4(cr_alternative.x, cr_alternative.y, cr_alternative.z, 0.0f) : kp.vp_prev.alternative[pix];
    if (first) { kp.vp.flux_radius[pix] = make_float4(0, 0, 0, 0); kp.vp.count[pix] = 0u; }
    reinterpret_cast<uint4*>(kp.canvas_rng)[pix] = ex_rng(rng);

    // cameraAABB + kernelCameraReducing (Photon.metal:157-161,169-218): exact min / max of the valid positions over the {FLT_MAX, -FLT_MAX}
    // every other pixel contributes -- the keys start there (trc_sppm_init), so a frame without ANY visible point ends like the reference's
    // (box size -inf, radius -inf, hash scale -0), and a NaN component is skipped as min / max skip it
    if (kp.frame_count == 0 && valid) {
        const float c[3] = {cr_position.x, cr_position.y, cr_position.z};
        for (int k = 0; k < 3; ++k)
            if (c[k] == c[k]) { atomicMin(&kp.cx->key_min[k], f2key(c[k])); atomicMax(&kp.cx->key_max[k], f2key(c[k])); }
    }
    }
    const uint32_t r = wave_sum(n_rays);
    if (lane == 0 && r) atomicAdd(&stat_row(kp.stats, blockIdx.x)[kStatRays], (unsigned long long)r);
}

// kernelPhotonParams, Photon.metal:357-372
__global__ void k_sppm_params(DComplex* x) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    F3 lo = f3(key2f(x->key_min[0]), key2f(x->key_min[1]), key2f(x->key_min[2]));
    F3 hi = f3(key2f(x->key_max[0]), key2f(x->key_max[1]), key2f(x->key_max[2]));
    F3 size = hi - lo;
    float radius = dot(size, f3(1.0f / 3.0f));
    radius *= 2.5f / (1 << 12);
    lo = lo - f3(radius);
    hi = hi + f3(radius);
    x->box_min[0] = lo.x; x->box_min[1] = lo.y; x->box_min[2] = lo.z;
    x->box_max[0] = hi.x; x->box_max[1] = hi.y; x->box_max[2] = hi.z;
    x->box_size[0] = size.x; x->box_size[1] = size.y; x->box_size[2] = size.z;
    x->initial_radius = radius;
    x->hash_scale = 1.0f / (radius * 1.5f);
}

// kernelPhotonRadius, Photon.metal:374-384
__global__ void __launch_bounds__(256) k_sppm_radius(float4* flux_radius, uint32_t n, const DComplex* cx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flux_radius[i].w = cx->initial_radius;
}

// kernelPhotonRecording, Photon.metal:286-355 + tracePhotonRecord :220-285
template <bool ALL_LDS>
__global__ void __launch_bounds__(kBlock, TRC_SPPM_PHOTON_WAVES) k_sppm_photon(const KSppm kp) {
    const uint32_t* small_base = stage_scene(kp.ks.sc);
    const SppmCtx cx = make_sppm_ctx(kp, small_base);
    const uint32_t idx = kp.photon_first + blockIdx.x * kBlock + threadIdx.x;     // grid covers exactly this rank's photon range
    uint32_t n_rays = 0;
    trc_PhotonRecord& slot = kp.pho_rec[idx];
    F3 flux = ld3(slot.flux), normal = ld3(slot.normal), position = ld3(slot.position), direction = ld3(slot.direction);
    uint32_t step = slot.step;
    bool active = slot.active != 0;
    Pcg rng = to_rng(reinterpret_cast<const uint4*>(kp.photon_rng)[idx]);

    Ray ray;
    const bool check = (kp.frame_count == 0) || (step == 0) || (step == 8);
    if (check) {                                    // a new photon from the light (squareList[5]; `random() < 1`)
        flux = f3(1); step = 0; active = false;     // reset()
        LightSample lsr;
        F2 uu; uu.x = pcg_float(rng); uu.y = pcg_float(rng);
        const F3 _origin = f3(450, 250, 250);
        if (pcg_float(rng) < 1) square_sample(cx.S, 5, uu, _origin, lsr);
        else square_sample(cx.S, 6, uu, _origin, lsr);
        flux = mat_albedo(cx.sh, lsr.material) * 100000.0f;
        F3 nx, ny;
        coordinate_system(lsr.n, nx, ny);
        uu.x = pcg_float(rng); uu.y = pcg_float(rng);
        F3 h = uniform_sample_hemisphere(uu);
        ray = make_ray(lsr.p, (nx * h.x + ny * h.y) + lsr.n * h.z);
    } else {
        ray = make_ray(position, direction);
    }
    {   // tracePhotonRecord
        HitRec rec;
        hit_init(rec);
        F3 ratio = f3(1.0f);
        const bool hitted = sppm_hit<ALL_LDS>(cx, ray, rec, n_rays);
        const int mtype = mat_type(cx.sh, rec.material);
        bool alive = hitted && mtype != kMatDiffuse;
        F3 nx = f3(0), ny = f3(0), wi = f3(0);
        if (alive) {
            coordinate_system(rec.sn, nx, ny);
            float bxPDF = 0;
            F3 minus_d = -ray.d;
            F3 wo = f3(dot(nx, minus_d), dot(ny, minus_d), dot(rec.sn, minus_d));
            F2 uu; uu.x = pcg_float(rng); uu.y = pcg_float(rng);
            F3 attenuation = material_S_F(mtype, hit_color(cx.S, cx.sh, rec), wo, wi, uu, bxPDF);
            if (bxPDF <= 0) alive = false;
            else {
                ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
                float p = rgb_to_y(ratio);
                if (pcg_float(rng) > p) alive = false;
                else ratio = ratio * (1.0f / p);
            }
        }
        if (!alive) { flux = f3(1); step = 0; active = false; }      // reset()
        else {
            // a NaN wi (degenerate BSDF sample) counts as positive and is stored canonical: the sign of a NaN is platform business
            // (oracle.cpp tracePhotonRecord)
            F3 pn = rec.sn * (is_nan(wi.z) ? 1.0f : copysignf(1.0f, wi.z));
            position = offset_ray(rec.p, pn);
            normal = pn;
            direction = canon_nan((nx * wi.x + ny * wi.y) + rec.sn * wi.z);
            flux = canon_nan(flux * ratio);
            step = (step + 1) & 0xFFu;
            active = !mat_specular(cx.sh, rec.material);
        }
    }
    st3(slot.flux, flux); st3(slot.normal, normal); st3(slot.position, position); st3(slot.direction, direction);
    slot.step = (uint8_t)step; slot.active = active ? 1 : 0;
    reinterpret_cast<uint4*>(kp.photon_rng)[idx] = ex_rng(rng);
    const uint32_t r = wave_sum(n_rays);
    if ((threadIdx.x & 63u) == 0 && r) atomicAdd(&stat_row(kp.stats, blockIdx.x)[kStatRays], (unsigned long long)r);
}

// What the passes AFTER the photon pass read of a photon: position, direction, flux and whether it is active -- 40 of the
// record's 80 bytes (normal and step belong to the photon's own bounce chain, i.e. to the rank that owns it).  With several
// ranks that is what crosses the links every frame: rank r packs its own photon range into `wire`, ONE in-place all-gather
// moves 10.5 MB instead of 21 MB, and the hash / table passes read the wire records; the full records are gathered only
// when a host asks for them (trc_sppm_download).  One rank: no wire, the passes read the records themselves.
struct PhotonWire { float position[3], direction[3], flux[3]; uint32_t active; };
static_assert(sizeof(PhotonWire) == 40, "wire record");
__global__ void __launch_bounds__(256) k_sppm_pack_wire(const trc_PhotonRecord* pho, PhotonWire* wire, uint32_t first, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const trc_PhotonRecord& r = pho[first + i];
    PhotonWire w;
    w.position[0] = r.position.x; w.position[1] = r.position.y; w.position[2] = r.position.z;
    w.direction[0] = r.direction.x; w.direction[1] = r.direction.y; w.direction[2] = r.direction.z;
    w.flux[0] = r.flux.x; w.flux[1] = r.flux.y; w.flux[2] = r.flux.z;
    w.active = r.active ? 1u : 0u;
    wire[first + i] = w;
}
struct PhotonView {                      // one of the two is null
    const trc_PhotonRecord* pho; const PhotonWire* wire;
    __device__ __forceinline__ bool active(uint32_t i) const { return wire ? wire[i].active != 0u : pho[i].active != 0; }
    __device__ __forceinline__ F3 position(uint32_t i) const { return wire ? f3(wire[i].position[0], wire[i].position[1], wire[i].position[2]) : ld3(pho[i].position); }
    __device__ __forceinline__ F3 direction(uint32_t i) const { return wire ? f3(wire[i].direction[0], wire[i].direction[1], wire[i].direction[2]) : ld3(pho[i].direction); }
    __device__ __forceinline__ F3 flux(uint32_t i) const { return wire ? f3(wire[i].flux[0], wire[i].flux[1], wire[i].flux[2]) : ld3(pho[i].flux); }
};

// kernelPhotonHashing + point raster (PhotonMarkVS/FS), Photon.metal:386-456
__global__ void __launch_bounds__(256) k_sppm_hash(const PhotonView pv, uint32_t* mark, uint32_t* count, const DComplex* cx) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kHashN * kHashN) return;
    if (!pv.active(idx)) return;            // z = -1: clipped
    const F3 position = pv.position(idx);
    const float scale = cx->hash_scale;
    F3 hi = (position - f3(cx->box_min[0], cx->box_min[1], cx->box_min[2])) * scale;
    hi = f3(floorf(hi.x), floorf(hi.y), floorf(hi.z));
    const float hashed = ph_hash(hi, hash_div(scale), (float)kHashN);
    const float tx = ph_mod(hashed, (float)kHashN) - 1.0f, ty = floorf(hashed / (float)kHashN) - 1.0f;
    if (!(tx >= 0.0f && tx < (float)kHashN && ty >= 0.0f && ty < (float)kHashN)) return;
    const uint32_t cell = (uint32_t)ty * kHashN + (uint32_t)tx;
    atomicMax(&mark[cell], idx + 1u);
    atomicAdd(&count[cell], 1u);
}

// kernelPhotonSumming (Photon.metal:458-496) + the gather table of the refine pass.
//
// The reference's refine reads, per hash cell a pixel looks at, the mark texture (which photon won the cell), the count
// texture (how many fell into it) and then three fields of that photon's 80-byte record: four dependent gathers from
// three arrays.  Which photon a cell shows and with what weight is the same for every pixel of the frame, so it is
// resolved ONCE per cell here (262 144 cells, a streaming pass over the two grids the sum reads anyway) into
//    q0 = photon position, Correction (= count; -1: empty cell)     -- array A, 16 bytes per cell: what EVERY visited
//                                                                      cell costs; 4.2 MB, i.e. about one XCD's L2
//    q1 = photon direction, q2 = photon flux                         -- array B, 32 bytes per cell, read only for a
//                                                                      photon that passed the distance test
// Record 512*512 stands for the reference's out-of-range texture read (returns 0: photon (0,0), count 0,
// Photon.metal:556-558).  (grid-stride: 128 workgroups, so the single counter sees 512 atomics per frame, not 4096)
__global__ void __launch_bounds__(256) k_sppm_table(const uint32_t* mark, const uint32_t* count, const PhotonView pv,
                                                    float4* cells, DComplex* cx) {
    uint32_t v = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= kHashN * kHashN; i += gridDim.x * blockDim.x) {
        const bool oob = i == kHashN * kHashN;
        const uint32_t c = oob ? 0u : count[i], mk = oob ? 1u : mark[i];
        v += c > 0 ? (c > 1u ? c : 1u) : 0u;
        float4 q0 = make_float4(0, 0, 0, -1.0f), q1 = make_float4(0, 0, 0, 0), q2 = q1;
        if (mk != 0u) {
            const F3 pp = pv.position(mk - 1u), pd = pv.direction(mk - 1u), pf = pv.flux(mk - 1u);
            q0 = make_float4(pp.x, pp.y, pp.z, (float)c);
            q1 = make_float4(pd.x, pd.y, pd.z, 0.0f);
            q2 = make_float4(pf.x, pf.y, pf.z, 0.0f);
        }
        cells[i] = q0; cells[kCellsB + 2u * i] = q1; cells[kCellsB + 2u * i + 1u] = q2;
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&cx->frame_photon_sum, v);
}

// kernelPhotonRefine, Photon.metal:498-623: progressive radius / flux update of every visible point from the photons
// of the hash cells its query sphere touches, then the running mean of the frame.
//
// One lane per pixel, one wavefront per 8x8 block.  At 1080p neighbouring pixels are ~1.5 hash cells apart (cell edge
// = 1.5 x the initial radius = 0.5 scene units, pixel footprint 0.77), so lanes share no cells and there is nothing to
// stage for the wavefront; what the pass was waiting for (70 % of its wavefront cycles, rocprofv3 SQ_WAIT_ANY) is the
// CHAIN of gathers per cell, one cell after the other.  Here a lane walks its cells -- in the reference's z, y, x
// order, which fixes the floating-point summation order -- four at a time: four hashes, then four independent table
// reads in flight together, then the four contributions added in order; direction and flux are fetched only for a
// photon that passed the distance test.
#ifndef TRC_REFINE_CHUNK
#define TRC_REFINE_CHUNK 4
#endif
constexpr int kRefineChunk = TRC_REFINE_CHUNK;
__global__ void __launch_bounds__(kBlock, TRC_SPPM_REFINE_WAVES) k_sppm_refine(const KSppm kp) {
    const uint32_t tile = kp.tiles[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t qx = (tile & 0xFFFFu) * 8u + (lane & 7u);
    const uint32_t qy = (tile >> 16) * 8u + (lane >> 3);
    if (qx >= kp.W || qy >= kp.H) return;
    const uint32_t pixel = qy * kp.W + qx;
    const float4 rv = kp.vp.ratio_valid[pixel];                          // the pixel's visible point
    float4* px = reinterpret_cast<float4*>(kp.accum) + pixel;
    const float4 shown = *px;
    const F3 mean = f3(shown.x, shown.y, shown.z);
    const float frame = (float)kp.frame_count, frame1 = (float)(kp.frame_count + 1);
    if (__float_as_uint(rv.w) == 0u) {                                   // nothing diffuse was seen: the stored radiance
        const float4 alt = kp.vp.alternative[pixel];
        const F3 result = (mean * frame + f3(alt.x, alt.y, alt.z)) / frame1;
        *px = make_float4(result.x, result.y, result.z, 1.0f);
        return;
    }
    const DComplex& cx = *kp.cx;
    const float4 vpos = kp.vp.position[pixel], vdir = kp.vp.direction[pixel], fr = kp.vp.flux_radius[pixel];
    const F3 at = f3(vpos.x, vpos.y, vpos.z), facing = f3(vdir.x, vdir.y, vdir.z), reflectance = f3(rv.x, rv.y, rv.z);
    F3 flux = f3(fr.x, fr.y, fr.z);
    float radius = fr.w;
    uint32_t n_photons = kp.vp.count[pixel];
    const F3 box_min = f3(cx.box_min[0], cx.box_min[1], cx.box_min[2]);
    const float scale = cx.hash_scale, fN = (float)kHashN;
    const HashDiv hd = hash_div(scale);
    // cells touched by the query sphere's box, as the reference derives them (abs() included, :533-536)
    const F3 lo = at - f3(radius) - box_min, hi = at + f3(radius) - box_min;
    const F3 flo = f3(fabsf(lo.x), fabsf(lo.y), fabsf(lo.z)) * scale, fhi = f3(fabsf(hi.x), fabsf(hi.y), fabsf(hi.z)) * scale;
    const int x0 = (int)flo.x, x1 = (int)fhi.x, y0 = (int)flo.y, y1 = (int)fhi.y, z0 = (int)flo.z, z1 = (int)fhi.z;
    F3 gathered = f3(0);
    uint32_t gathered_n = 0;
    int ix = x0, iy = y0, iz = z0;
    bool more = x0 <= x1 && y0 <= y1 && z0 <= z1;
    while (more) {
        F3 cell[kRefineChunk];
        uint32_t rec[kRefineChunk];
        bool use[kRefineChunk];
#pragma unroll
        for (int j = 0; j < kRefineChunk; ++j) {                         // the next (up to) four cells: hash -> table index
            use[j] = more;
            cell[j] = f3((float)ix, (float)iy, (float)iz);
            rec[j] = kHashN * kHashN;                                    // out-of-range read
            if (more) {
                const float hashed = ph_hash(cell[j], hd, fN);
                const float hx = ph_mod(hashed, fN) - 1.0f, hy = floorf(hashed / fN) - 1.0f;
                if (hx >= 0.0f && hx < fN && hy >= 0.0f && hy < fN) rec[j] = (uint32_t)hy * kHashN + (uint32_t)hx;
                if (++ix > x1) { ix = x0; if (++iy > y1) { iy = y0; if (++iz > z1) more = false; } }
            }
        }
        float4 q0[kRefineChunk];
#pragma unroll
        for (int j = 0; j < kRefineChunk; ++j) q0[j] = use[j] ? kp.cells[rec[j]] : make_float4(0, 0, 0, -1.0f);
#pragma unroll
        for (int j = 0; j < kRefineChunk; ++j) {
            const float weight = q0[j].w;                                // Correction; -1 marks an empty cell (:560)
            if (!use[j] || weight < 0.0f) continue;
            const F3 p = f3(q0[j].x, q0[j].y, q0[j].z);
            const F3 cmin = cell[j] / hd.scale + box_min, cmax = (cell[j] + f3(1.0f)) / hd.scale + box_min;
            if (!((cmin.x < p.x) && (p.x < cmax.x) && (cmin.y < p.y) && (p.y < cmax.y) && (cmin.z < p.z) && (p.z < cmax.z))) continue;
            const float d = length(p - at);
            if (!(d < radius)) continue;
            const float4 q1 = kp.cells[kCellsB + 2u * rec[j]];
            if (!(-dot(facing, f3(q1.x, q1.y, q1.z)) > 0.001f)) continue;
            const float4 q2 = kp.cells[kCellsB + 2u * rec[j] + 1u];
            gathered = gathered + f3(q2.x, q2.y, q2.z) * weight;
            gathered_n = (uint32_t)((float)gathered_n + weight);
        }
    }
    gathered = gathered * (reflectance / 3.141592f);                     // Lambertian BRDF
    const float alpha = 0.8f;                                            // progressive photon mapping (:596-603)
    const float g = fminf(((float)n_photons + (float)gathered_n * alpha) / (float)(n_photons + gathered_n), 1.0f);
    radius = radius * sqrtf(g);
    n_photons = (uint32_t)((float)n_photons + (float)gathered_n * alpha);
    flux = (flux + gathered) * g;
    kp.vp.flux_radius[pixel] = make_float4(flux.x, flux.y, flux.z, radius);
    kp.vp.count[pixel] = n_photons;
    float emitted = cx.total_photon_sum;
    emitted += (float)cx.frame_photon_sum;
    const F3 radiance = flux / (radius * radius * 3.141592f * emitted);
    F3 result = (mean * frame + radiance) / frame1;
    if (is_nan(result.x) || is_nan(result.y) || is_nan(result.z)) result = f3(0);
    *px = make_float4(result.x, result.y, result.z, 1.0f);
}

// the planes back into the reference's 112-byte records (trc_sppm_download)
__global__ void __launch_bounds__(256) k_sppm_pack_records(const VisiblePoints vp, trc_CameraRecord* out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 rv = vp.ratio_valid[i], po = vp.position[i], di = vp.direction[i], al = vp.alternative[i], fr = vp.flux_radius[i];
    trc_CameraRecord r;
    memset(&r, 0, sizeof r);
    r.ratio.x = rv.x; r.ratio.y = rv.y; r.ratio.z = rv.z;
    r.position.x = po.x; r.position.y = po.y; r.position.z = po.z;
    r.direction.x = di.x; r.direction.y = di.y; r.direction.z = di.z;
    r.valid = __float_as_uint(rv.w) ? 1 : 0;
    r.alternative.x = al.x; r.alternative.y = al.y; r.alternative.z = al.z;
    r.flux.x = fr.x; r.flux.y = fr.y; r.flux.z = fr.z;
    r.radius = fr.w;
    r.photonCount = vp.count[i];
    out[i] = r;
}

#ifdef TRC_TEST_HOOKS      // libtracer_amd_hooks.so only (include/tracer_test_hooks.h)
// trc_sppm_hash_cells
__global__ void __launch_bounds__(256) k_sppm_hash_cells(const float* cells, uint32_t n, float scale, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = ph_hash(f3(cells[3 * i], cells[3 * i + 1], cells[3 * i + 2]), hash_div(scale), (float)kHashN);
}
#endif

// completion handler, AAPLRenderer.mm:1031-1036
__global__ void k_sppm_end_frame(DComplex* x) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    x->total_photon_sum += (float)x->frame_photon_sum;
    x->frame_photon_sum = 0;
}

}  // namespace

struct SppmState {
    uint32_t W = 0, H = 0, frame_count = 0;
    uint32_t* d_photon_rng = nullptr;
    void* d_vp = nullptr;            // the visible-point planes, one allocation
    VisiblePoints vp[2]{};           // [cur] = current values; the two share flux_radius and count
    int cur = 0;
    hipStream_t cam_stream = nullptr;    // the camera pass of the next odd frame runs here, beside the current frame
    hipEvent_t ev_main = nullptr, ev_cam = nullptr;
    uint32_t cam_ahead = 0;          // frame whose camera pass is already in flight on cam_stream (0 = none)
    bool poisoned = false;           // a frame failed half-way (collective / HIP error): the pass state is undefined
    trc_PhotonRecord* d_pho = nullptr;
    PhotonWire* d_wire = nullptr;        // grouped runs: what the other ranks see of a photon (allocated on first use)
    bool pho_partial = false;            // grouped runs: d_pho holds only this rank's records up to date
    uint32_t* d_mark = nullptr;
    uint32_t* d_count = nullptr;
    float4* d_cells = nullptr;       // gather table of k_sppm_table
    DComplex* d_cx = nullptr;
};

// a camera pass running ahead on its own stream reads and advances the canvas RNG texels: whatever the caller queues next
// on the context stream that touches them (trc_seed, trc_upload_rng) is ordered after it
void trc_sppm_order_after_camera(trc_ctx* ctx) {
    if (!ctx || !ctx->sppm || ctx->sppm->cam_ahead == 0) return;
    (void)hipStreamWaitEvent(ctx->stream, ctx->sppm->ev_cam, 0);
}

void trc_sppm_release(trc_ctx* ctx) {
    if (!ctx || !ctx->sppm) return;
    SppmState* s = ctx->sppm;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (s->cam_stream) { (void)hipStreamSynchronize(s->cam_stream); (void)hipStreamDestroy(s->cam_stream); }
    if (s->ev_main) (void)hipEventDestroy(s->ev_main);
    if (s->ev_cam) (void)hipEventDestroy(s->ev_cam);
    (void)hipFree(s->d_photon_rng); (void)hipFree(s->d_vp); (void)hipFree(s->d_pho); (void)hipFree(s->d_wire);
    (void)hipFree(s->d_mark); (void)hipFree(s->d_count); (void)hipFree(s->d_cells); (void)hipFree(s->d_cx);
    delete s;
    ctx->sppm = nullptr;
}

__global__ void __launch_bounds__(256) k_sppm_seed(uint32_t* rng, uint32_t n, uint64_t seed) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    Pcg r; r.state = 0; r.inc = ((uint64_t)p << 1u) | 1u;
    pcg_next(r); r.state += seed; pcg_next(r);
    uint4 out; out.x = pcg_next(r); out.y = pcg_next(r); out.z = pcg_next(r); out.w = pcg_next(r);
    reinterpret_cast<uint4*>(rng)[p] = out;
}

extern "C" {

trc_status trc_sppm_init(trc_ctx* ctx, uint64_t photon_seed) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_accum) return trc_fail(ctx, TRC_ERR_NO_FRAME, "trc_sppm_init before trc_resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    trc_sppm_release(ctx);
    SppmState* s = new (std::nothrow) SppmState();
    if (!s) return TRC_ERR_OOM;
    ctx->sppm = s;
    s->W = ctx->width; s->H = ctx->height;
    const size_t np = (size_t)s->W * s->H, nph = (size_t)kHashN * kHashN;
    HIP_TRY(ctx, hipMalloc((void**)&s->d_photon_rng, nph * 16));
    const size_t vp_bytes = np * (9 * sizeof(float4) + sizeof(uint32_t));
    HIP_TRY(ctx, hipMalloc(&s->d_vp, vp_bytes));
    {
        float4* q = static_cast<float4*>(s->d_vp);
        for (int c = 0; c < 2; ++c) {
            VisiblePoints& v = s->vp[c];
            v.ratio_valid = q + (4 * c) * np; v.position = q + (4 * c + 1) * np; v.direction = q + (4 * c + 2) * np;
            v.alternative = q + (4 * c + 3) * np;
            v.flux_radius = q + 8 * np; v.count = reinterpret_cast<uint32_t*>(q + 9 * np);
        }
    }
    {   // lowest priority: the pass fills what the frame beside it leaves idle
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        HIP_TRY(ctx, hipStreamCreateWithPriority(&s->cam_stream, hipStreamNonBlocking, least));
    }
    HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&s->ev_cam, hipEventDisableTiming));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_pho, (nph + kPhotonSlack) * sizeof(trc_PhotonRecord)));      // + the padding of an uneven split over ranks
    HIP_TRY(ctx, hipMalloc((void**)&s->d_mark, nph * 4));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_count, nph * 4));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_cells, (nph + 1) * 3 * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_cx, sizeof(DComplex)));
    HIP_TRY(ctx, hipMemsetAsync(s->d_vp, 0, vp_bytes, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_pho, 0, (nph + kPhotonSlack) * sizeof(trc_PhotonRecord), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_mark, 0, nph * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_count, 0, nph * 4, ctx->stream));
    DComplex h;
    std::memset(&h, 0, sizeof h);
    for (int k = 0; k < 3; ++k) { h.key_min[k] = 0xFF7FFFFFu; h.key_max[k] = 0x00800000u; }      // f2key(FLT_MAX), f2key(-FLT_MAX): k_sppm_camera
    HIP_TRY(ctx, hipMemcpyAsync(s->d_cx, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // `h` is a stack temporary
    hipLaunchKernelGGL(k_sppm_seed, dim3((unsigned)((nph + 255) / 256)), dim3(256), 0, ctx->stream, s->d_photon_rng,
                       (uint32_t)nph, photon_seed);
    HIP_TRY(ctx, hipGetLastError());
    return TRC_OK;
}

trc_status trc_sppm_frames(trc_ctx* ctx, uint32_t n_frames) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    SppmState* s = ctx->sppm;
    if (!s) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_sppm_frames before trc_sppm_init");
    if (!ctx->has_scene) return trc_fail(ctx, TRC_ERR_NO_SCENE, "trc_sppm_frames before trc_upload_scene");
    if (!ctx->has_camera) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_sppm_frames before trc_set_camera");
    if (ctx->ks.sc.n_squares < 7) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "SPPM emits photons from squareList[5] (Photon.metal:316-320)");
    if (s->W != ctx->width || s->H != ctx->height) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "frame resized after trc_sppm_init");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Multi-GPU: with a communicator (trc_group_init) rank r traces the camera records and refines
    // the pixels of ITS tiles, and bounces ITS photon index range; the camera-record AABB is all-reduced
    // (min/max on order-preserving keys: exact) and the photon records are all-gathered every frame so that
    // every rank hashes the full photon set.  All of it is deterministic: N ranks == 1 rank, bit for bit.
    if (s->poisoned) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "an earlier trc_sppm_frames failed half-way through a frame: call trc_sppm_init again");
    const bool grouped = ctx->grouped();
    const uint32_t nranks = grouped ? (uint32_t)ctx->nranks : 1u, rank = grouped ? (uint32_t)ctx->rank : 0u;
    const uint32_t np = s->W * s->H, nph = kHashN * kHashN;
    // rank r bounces photons [r * chunk, r * chunk + mine): chunks of whole wavefronts, the last non-empty one shorter when the ranks do not
    // divide the 4 096 wavefronts of photons; the all-gathers move `chunk` records per rank into buffers padded by kPhotonSlack records
    if (nranks > kPhotonSlack / kBlock) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "the grouped SPPM pass splits its photons over at most 64 ranks");
    const uint32_t chunk = (nph / kBlock + nranks - 1) / nranks * kBlock;
    const uint32_t first = std::min(nph, rank * chunk), mine = std::min(chunk, nph - first);
    { trc_status ts = trc_ensure_tiles(ctx, nranks, rank); if (ts != TRC_OK) return ts; }
    // A rank that owns no tile of a small frame still bounces ITS photon range and takes part in every collective (it returned early
    // here up to round 5 and left the others waiting in the all-reduce: tests/campaigns/fuzz_ranks.sh); only its camera and refine launches are empty.
    if (ctx->n_tiles == 0 && !grouped) return TRC_OK;
    const bool has_tiles = ctx->n_tiles != 0;

    KSppm kp{};
    kp.ks = ctx->ks; kp.cam = ctx->cam;
    kp.ambient[0] = ctx->ambient[0]; kp.ambient[1] = ctx->ambient[1]; kp.ambient[2] = ctx->ambient[2];
    kp.env_rgb = ctx->d_envmap; kp.env_w = ctx->env_w; kp.env_h = ctx->env_h;
    kp.W = s->W; kp.H = s->H;
    kp.photon_first = first;
    kp.tiles = ctx->d_tiles;
    kp.canvas_rng = ctx->d_rng; kp.accum = ctx->d_accum; kp.photon_rng = s->d_photon_rng;
    kp.vp = kp.vp_prev = s->vp[s->cur]; kp.pho_rec = s->d_pho; kp.mark = s->d_mark; kp.count = s->d_count; kp.cells = s->d_cells; kp.cx = s->d_cx;
    kp.stats = ctx->d_stats;
    const size_t lds = trc_dyn_lds_bytes(ctx, false);
    const bool all_lds = ctx->lds_scene;
    // a frame that fails after its first launch leaves RNG texels advanced and a camera pass possibly in flight: no
    // retry can reproduce the frame, so the state is marked and the caller has to start over (trc_sppm_init)
    auto frame_failed = [&](trc_status st) {
        s->poisoned = true;
        (void)hipStreamSynchronize(s->cam_stream);
        return st;
    };
    auto camera_launch = [&](const KSppm& k, hipStream_t st) {
        if (!has_tiles) return;
        if (all_lds) hipLaunchKernelGGL((k_sppm_camera<true>), dim3(ctx->n_tiles), dim3(kBlock), lds, st, k);
        else hipLaunchKernelGGL((k_sppm_camera<false>), dim3(ctx->n_tiles), dim3(kBlock), lds, st, k);
    };
    // The camera pass of an odd frame depends on nothing the photon / refine passes produce (its RNG texels, the scene,
    // the copy of the visible points it reads), and it lasts as long as its slowest wavefront -- an 8-bounce specular
    // chain, 0.4 ms at 1080p with the GPU 90 % idle (docs/HISTORY.md section 9).  So it runs on its own stream, reading copy
    // `cur` of the visible points and writing copy `cur ^ 1`, after everything queued so far on the context stream (the
    // refines that still read the copy it overwrites); the refine of ITS frame waits for it and switches copies.  Within
    // one call it is started a whole frame early, at the top of the even frame before.
    auto camera_beside = [&](uint32_t frame) -> trc_status {
        KSppm kc = kp;
        kc.frame_count = frame; kc.vp_prev = s->vp[s->cur]; kc.vp = s->vp[s->cur ^ 1];
        HIP_TRY(ctx, hipEventRecord(s->ev_main, ctx->stream));
        HIP_TRY(ctx, hipStreamWaitEvent(s->cam_stream, s->ev_main, 0));
        camera_launch(kc, s->cam_stream);
        HIP_TRY(ctx, hipEventRecord(s->ev_cam, s->cam_stream));
        s->cam_ahead = frame;
        return TRC_OK;
    };
    const bool serial_camera = ctx->knobs.sppm_serial_camera != 0;                          // A/B knob: no frame of lead
    for (uint32_t f = 0; f < n_frames; ++f) {
        kp.frame_count = s->frame_count;
        kp.vp = kp.vp_prev = s->vp[s->cur];
        if (s->frame_count == 0) {                      // photonPrepare, AAPLRenderer.mm:860-947
            camera_launch(kp, ctx->stream);
            if (grouped) {
                DComplex* cx = s->d_cx;              // the reference's ping-pong tree over the records, Photon.metal:169-218
                trc_status cs = trc_coll_allreduce(ctx, cx->key_min, 3, kNcclUint32, kNcclMin, ctx->stream, "allreduce(min) of the bound keys");
                if (cs == TRC_OK) cs = trc_coll_allreduce(ctx, cx->key_max, 3, kNcclUint32, kNcclMax, ctx->stream, "allreduce(max) of the bound keys");
                if (cs != TRC_OK) return frame_failed(cs);
            }
            hipLaunchKernelGGL(k_sppm_params, dim3(1), dim3(64), 0, ctx->stream, s->d_cx);
            hipLaunchKernelGGL(k_sppm_radius, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, s->vp[0].flux_radius, np, s->d_cx);
        }
        const bool camera_frame = (s->frame_count % 2) != 0;       // photonWork re-runs the camera pass on odd frames, :953-955
        if (camera_frame && s->cam_ahead != s->frame_count) { trc_status ts = camera_beside(s->frame_count); if (ts != TRC_OK) return frame_failed(ts); }
        if (!camera_frame && f + 1 < n_frames && !serial_camera && s->cam_ahead != s->frame_count + 1) {
            trc_status ts = camera_beside(s->frame_count + 1);
            if (ts != TRC_OK) return frame_failed(ts);
        }
        // knob "sppm_timing": the frame's two segments on the context stream -- [photon pass] and [hash, table,
        // refine], i.e. everything but the collectives -- timed with event pairs into trc_stats.kernel_ms / launches
        hipEvent_t seg[4] = {nullptr, nullptr, nullptr, nullptr};
        const bool timing = ctx->knobs.sppm_timing != 0;
        if (timing) {
            for (hipEvent_t& e : seg) e = trc_get_event(ctx);
            if (!seg[0] || !seg[1] || !seg[2] || !seg[3]) return frame_failed(trc_fail(ctx, TRC_ERR_HIP, "hipEventCreate failed"));
            ctx->pending.emplace_back(seg[0], seg[1]); ctx->pending.emplace_back(seg[2], seg[3]);
            ctx->launches++;
            (void)hipEventRecord(seg[0], ctx->stream);
        }
        if (mine == 0) {}
        else if (all_lds) hipLaunchKernelGGL((k_sppm_photon<true>), dim3(mine / kBlock), dim3(kBlock), lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_sppm_photon<false>), dim3(mine / kBlock), dim3(kBlock), lds, ctx->stream, kp);
        if (timing) (void)hipEventRecord(seg[1], ctx->stream);
        PhotonView pv{s->d_pho, nullptr};
        if (grouped) {                                  // every rank needs every photon for hashing + refine: 40 bytes of each
            if (!s->d_wire && hipMalloc((void**)&s->d_wire, (size_t)(nph + kPhotonSlack) * sizeof(PhotonWire)) != hipSuccess)
                return frame_failed(trc_fail(ctx, TRC_ERR_OOM, "hipMalloc photon wire records"));
            if (mine) hipLaunchKernelGGL(k_sppm_pack_wire, dim3((mine + 255) / 256), dim3(256), 0, ctx->stream, s->d_pho, s->d_wire, kp.photon_first, mine);
            trc_status cs = trc_coll_allgather(ctx, s->d_wire, (size_t)chunk * sizeof(PhotonWire), ctx->stream, "allgather of the photon wire records");
            if (cs != TRC_OK) return frame_failed(cs);
            pv = PhotonView{nullptr, s->d_wire};
            s->pho_partial = nranks > 1;
        }
        if (timing) (void)hipEventRecord(seg[2], ctx->stream);
        if (hipMemsetAsync(s->d_mark, 0, (size_t)nph * 4, ctx->stream) != hipSuccess ||     // loadAction clear, :785-790
            hipMemsetAsync(s->d_count, 0, (size_t)nph * 4, ctx->stream) != hipSuccess) return frame_failed(trc_fail(ctx, TRC_ERR_HIP, "SPPM grid clear"));
        hipLaunchKernelGGL(k_sppm_hash, dim3(nph / 256), dim3(256), 0, ctx->stream, pv, s->d_mark, s->d_count, s->d_cx);
        hipLaunchKernelGGL(k_sppm_table, dim3(128), dim3(256), 0, ctx->stream, s->d_mark, s->d_count, pv, s->d_cells, s->d_cx);
        if (camera_frame) {                             // this frame's visible points: wait for them, switch copies
            if (hipStreamWaitEvent(ctx->stream, s->ev_cam, 0) != hipSuccess) return frame_failed(trc_fail(ctx, TRC_ERR_HIP, "SPPM camera event"));
            s->cur ^= 1; s->cam_ahead = 0;
            kp.vp = kp.vp_prev = s->vp[s->cur];
        }
        if (has_tiles) hipLaunchKernelGGL(k_sppm_refine, dim3(ctx->n_tiles), dim3(kBlock), 0, ctx->stream, kp);
        hipLaunchKernelGGL(k_sppm_end_frame, dim3(1), dim3(64), 0, ctx->stream, s->d_cx);
        if (timing) (void)hipEventRecord(seg[3], ctx->stream);
        { const hipError_t le = hipGetLastError(); if (le != hipSuccess) return frame_failed(trc_fail(ctx, TRC_ERR_HIP, std::string("SPPM frame: ") + hipGetErrorString(le))); }
        s->frame_count += 1;
    }
    return TRC_OK;
}

#ifdef TRC_TEST_HOOKS
trc_status trc_sppm_hash_cells(trc_ctx* ctx, const float* cells, size_t n, float hash_scale, float* out) {
    if (!ctx || (n && (!cells || !out))) return TRC_ERR_INVALID_ARG;
    if (n == 0) return TRC_OK;
    if (n > 0x7FFFFFFFu / 3u) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_sppm_hash_cells: too many cells in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_in = nullptr, *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_in, n * 12));
    if (hipMalloc((void**)&d_out, n * 4) != hipSuccess) { (void)hipFree(d_in); return trc_fail(ctx, TRC_ERR_OOM, "hipMalloc"); }
    trc_status ts = trc_copy_to_device(ctx, d_in, cells, n * 12, ctx->stream);
    if (ts == TRC_OK) {
        hipLaunchKernelGGL(k_sppm_hash_cells, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_in, (uint32_t)n, hash_scale, d_out);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) ts = trc_fail(ctx, TRC_ERR_HIP, std::string("trc_sppm_hash_cells: ") + hipGetErrorString(e));
    }
    if (ts == TRC_OK) ts = trc_copy_to_host(ctx, out, d_out, n * 4, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_in); (void)hipFree(d_out);
    return ts;
}
#endif  // TRC_TEST_HOOKS

trc_status trc_sppm_download(trc_ctx* ctx, trc_CameraRecord* cam, trc_PhotonRecord* pho, float* mark, float* count, trc_Complex* cx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    SppmState* s = ctx->sppm;
    if (!s) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_sppm_download before trc_sppm_init");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)s->W * s->H, nph = (size_t)kHashN * kHashN;
    trc_CameraRecord* d_packed = nullptr;
    if (cam) {
        HIP_TRY(ctx, hipMalloc((void**)&d_packed, np * sizeof(trc_CameraRecord)));
        hipLaunchKernelGGL(k_sppm_pack_records, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, ctx->stream, s->vp[s->cur], d_packed, (uint32_t)np);
        const trc_status cs = trc_copy_to_host(ctx, cam, d_packed, np * sizeof(trc_CameraRecord), ctx->stream);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_packed);
        if (cs != TRC_OK) return cs;
    }
    if (pho) {
        if (s->pho_partial && ctx->grouped()) {           // the per-frame gather moves 40 bytes per photon: the whole records, now (collective)
            const size_t chunk = (nph / kBlock + (size_t)ctx->nranks - 1) / (size_t)ctx->nranks * kBlock;      // trc_sppm_frames' split
            trc_status cs = trc_coll_allgather(ctx, s->d_pho, chunk * sizeof(trc_PhotonRecord), ctx->stream, "allgather of the photon records (download)");
            if (cs != TRC_OK) return cs;
            s->pho_partial = false;
        }
        { const trc_status cs = trc_copy_to_host(ctx, pho, s->d_pho, nph * sizeof(trc_PhotonRecord), ctx->stream); if (cs != TRC_OK) return cs; }
    }
    std::vector<uint32_t> hm, hc;
    if (mark) { hm.resize(nph); const trc_status cs = trc_copy_to_host(ctx, hm.data(), s->d_mark, nph * 4, ctx->stream); if (cs != TRC_OK) return cs; }
    if (count) { hc.resize(nph); const trc_status cs = trc_copy_to_host(ctx, hc.data(), s->d_count, nph * 4, ctx->stream); if (cs != TRC_OK) return cs; }
    DComplex h;
    HIP_TRY(ctx, hipMemcpyAsync(&h, s->d_cx, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (mark) for (size_t c = 0; c < nph; ++c) {
        if (hm[c] == 0) { mark[4 * c] = mark[4 * c + 1] = mark[4 * c + 2] = mark[4 * c + 3] = -1.0f; }
        else {
            const uint32_t w = hm[c] - 1;
            mark[4 * c] = (float)(w % kHashN); mark[4 * c + 1] = (float)(w / kHashN);
            mark[4 * c + 2] = (float)(c % kHashN); mark[4 * c + 3] = (float)(c / kHashN);
        }
    }
    if (count) for (size_t c = 0; c < nph; ++c) count[c] = (float)hc[c];
    if (cx) {
        std::memset(cx, 0, sizeof *cx);
        cx->frame_count = s->frame_count;
        cx->tex_size.x = cx->view_size.x = (float)s->W; cx->tex_size.y = cx->view_size.y = (float)s->H;
        cx->photonBox.mini.x = h.box_min[0]; cx->photonBox.mini.y = h.box_min[1]; cx->photonBox.mini.z = h.box_min[2];
        cx->photonBox.maxi.x = h.box_max[0]; cx->photonBox.maxi.y = h.box_max[1]; cx->photonBox.maxi.z = h.box_max[2];
        cx->photonBoxSize.x = h.box_size[0]; cx->photonBoxSize.y = h.box_size[1]; cx->photonBoxSize.z = h.box_size[2];
        cx->photonInitialRadius = h.initial_radius; cx->photonHashScale = h.hash_scale;
        cx->totalPhotonSum = h.total_photon_sum; cx->framePhotonSum = h.frame_photon_sum;
    }
    return TRC_OK;
}

}  // extern "C"

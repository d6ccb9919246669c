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
//   kernelPhotonSumming     :458-496  -> k_sppm_sum
//   kernelPhotonRefine      :498-623  -> k_sppm_refine
// All launches go to the context stream in order; nothing synchronises with the host.
#include "trc_ctx.hpp"

#include <cstring>
#include <new>

namespace {

constexpr uint32_t kHashN = TRC_PHOTON_HASHN;

struct DComplex {           // device-side Complex (Camera.hh:27-55) + the bound keys of the camera reduce
    float box_min[3], box_max[3], box_size[3];
    float initial_radius, hash_scale, total_photon_sum;
    uint32_t frame_photon_sum;
    uint32_t key_min[3], key_max[3];
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
    trc_CameraRecord* cam_rec;
    trc_PhotonRecord* pho_rec;
    uint32_t* mark;          // winning photon index + 1 per cell, 0 = empty
    uint32_t* count;
    DComplex* cx;
    unsigned long long* stats;   // ctx counters: rays += Scene::hit calls of the camera and photon passes
};

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

// Photon.hh:57-89
__device__ __forceinline__ float ph_mod(float x, float y) { return x - y * floorf(x / y); }
__device__ __forceinline__ float ph_hash(const F3 idx, const float HashScale, const float BufInfo) {
    const float HashNum = BufInfo * BufInfo;
    const float n[4] = {idx.x, idx.y, idx.z, idx.x + idx.y - idx.z};
    const float q[4] = {1225.0f, 1585.0f, 2457.0f, 2098.0f};
    const float r[4] = {1112.0f, 367.0f, 92.0f, 265.0f};
    const float a[4] = {3423.0f, 2646.0f, 1707.0f, 1999.0f};
    const float m[4] = {4194287.0f, 4194277.0f, 4194191.0f, 4194167.0f};
    float nm[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float nk = n[k] * 4194304.0f / HashScale;
        float beta = floorf(nk / q[k]);
        float pk = a[k] * (nk - beta * q[k]) - beta * r[k];
        float sgn = (-pk > 0.0f) ? 1.0f : ((-pk < 0.0f) ? -1.0f : 0.0f);
        beta = (sgn + 1.0f) * 0.5f * m[k];
        nk = pk + beta;
        nm[k] = nk / m[k];
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
template <bool ALL_LDS>
__global__ void __launch_bounds__(kBlock) k_sppm_camera(const KSppm kp) {
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

    trc_CameraRecord& slot = kp.cam_rec[pix];
    F3 cr_ratio = ld3(slot.ratio), cr_position = ld3(slot.position), cr_direction = ld3(slot.direction);
    F3 cr_alternative = ld3(slot.alternative), cr_flux = ld3(slot.flux);
    float cr_radius = slot.radius;
    uint32_t cr_count = slot.photonCount;
    int depth = 8;
    if (kp.frame_count == 0) {      // cr.reset(), Photon.hh:42-52 (alternative is NOT reset)
        cr_ratio = f3(1); cr_position = f3(0); cr_direction = f3(0); cr_flux = f3(0); cr_radius = 0; cr_count = 0;
        depth = 3;
    }
    bool valid = false;
    {
        HitRec rec;
        hit_init(rec);
        F3 ratio = f3(1.0f);
        bool hitted = sppm_hit<ALL_LDS>(cx, ray, rec, n_rays);
        bool finished = false;
        do {
            if (!hitted) { cr_alternative = ratio * env_radiance(cx.env, cx.ambient, ray.d); finished = true; break; }
            const int mtype = mat_type(cx.sh, rec.material);
            if (mtype == kMatDiffuse) {
                F3 le = mat_albedo(cx.sh, rec.material);
                float w = dot(-ray.d, -rec.gn);
                cr_alternative = ratio * le * fabsf(w);
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
            F3 attenuation = material_S_F(mtype, hit_color(cx.sh, rec), wo, wi, uu, bxPDF);
            if (bxPDF <= 0) break;
            F3 pn = rec.sn * copysignf(1.0f, wi.z);
            F3 _origin = offset_ray(rec.p, pn);
            ray = make_ray(_origin, (nx * wi.x + ny * wi.y) + rec.sn * wi.z);
            ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
            if (is_inf(ratio.x) || is_inf(ratio.y) || is_inf(ratio.z) || is_nan(ratio.x) || is_nan(ratio.y) || is_nan(ratio.z)) ratio = f3(1.0f);
            hitted = sppm_hit<ALL_LDS>(cx, ray, rec, n_rays);
        } while ((--depth) > 0);
        if (!finished) cr_alternative = f3(0);
    }
    st3(slot.ratio, cr_ratio); st3(slot.position, cr_position); st3(slot.direction, cr_direction);
    slot.valid = valid ? 1 : 0;
    st3(slot.alternative, cr_alternative); st3(slot.flux, cr_flux);
    slot.radius = cr_radius; slot.photonCount = cr_count;
    reinterpret_cast<uint4*>(kp.canvas_rng)[pix] = ex_rng(rng);

    if (kp.frame_count == 0 && valid) {         // cameraAABB + kernelCameraReducing: exact min/max of the valid positions
        atomicMin(&kp.cx->key_min[0], f2key(cr_position.x)); atomicMax(&kp.cx->key_max[0], f2key(cr_position.x));
        atomicMin(&kp.cx->key_min[1], f2key(cr_position.y)); atomicMax(&kp.cx->key_max[1], f2key(cr_position.y));
        atomicMin(&kp.cx->key_min[2], f2key(cr_position.z)); atomicMax(&kp.cx->key_max[2], f2key(cr_position.z));
    }
    }
    const uint32_t r = wave_sum(n_rays);
    if (lane == 0 && r) atomicAdd(&kp.stats[kStatRays], (unsigned long long)r);
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
__global__ void __launch_bounds__(256) k_sppm_radius(trc_CameraRecord* cam_rec, uint32_t n, const DComplex* cx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) cam_rec[i].radius = cx->initial_radius;
}

// kernelPhotonRecording, Photon.metal:286-355 + tracePhotonRecord :220-285
template <bool ALL_LDS>
__global__ void __launch_bounds__(kBlock) k_sppm_photon(const KSppm kp) {
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
            F3 attenuation = material_S_F(mtype, hit_color(cx.sh, rec), wo, wi, uu, bxPDF);
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
            F3 pn = rec.sn * copysignf(1.0f, wi.z);
            position = offset_ray(rec.p, pn);
            normal = pn;
            direction = (nx * wi.x + ny * wi.y) + rec.sn * wi.z;
            flux = flux * ratio;
            step = (step + 1) & 0xFFu;
            active = !mat_specular(cx.sh, rec.material);
        }
    }
    st3(slot.flux, flux); st3(slot.normal, normal); st3(slot.position, position); st3(slot.direction, direction);
    slot.step = (uint8_t)step; slot.active = active ? 1 : 0;
    reinterpret_cast<uint4*>(kp.photon_rng)[idx] = ex_rng(rng);
    const uint32_t r = wave_sum(n_rays);
    if ((threadIdx.x & 63u) == 0 && r) atomicAdd(&kp.stats[kStatRays], (unsigned long long)r);
}

// kernelPhotonHashing + point raster (PhotonMarkVS/FS), Photon.metal:386-456
__global__ void __launch_bounds__(256) k_sppm_hash(const trc_PhotonRecord* pho, uint32_t* mark, uint32_t* count, const DComplex* cx) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kHashN * kHashN) return;
    if (!pho[idx].active) return;            // z = -1: clipped
    const F3 position = ld3(pho[idx].position);
    const float scale = cx->hash_scale;
    F3 hi = (position - f3(cx->box_min[0], cx->box_min[1], cx->box_min[2])) * scale;
    hi = f3(floorf(hi.x), floorf(hi.y), floorf(hi.z));
    const float hashed = ph_hash(hi, scale, (float)kHashN);
    const float tx = ph_mod(hashed, (float)kHashN) - 1.0f, ty = floorf(hashed / (float)kHashN) - 1.0f;
    if (!(tx >= 0.0f && tx < (float)kHashN && ty >= 0.0f && ty < (float)kHashN)) return;
    const uint32_t cell = (uint32_t)ty * kHashN + (uint32_t)tx;
    atomicMax(&mark[cell], idx + 1u);
    atomicAdd(&count[cell], 1u);
}

// kernelPhotonSumming, Photon.metal:458-496
// (grid-stride: 128 workgroups, so the single counter sees 512 atomics per frame instead of 4096 -- they serialise)
__global__ void __launch_bounds__(256) k_sppm_sum(const uint32_t* count, DComplex* cx) {
    uint32_t v = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < kHashN * kHashN; i += gridDim.x * blockDim.x) {
        const uint32_t c = count[i];
        v += c > 0 ? (c > 1u ? c : 1u) : 0u;
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(&cx->frame_photon_sum, v);
}

// kernelPhotonRefine, Photon.metal:498-623
__global__ void __launch_bounds__(kBlock) k_sppm_refine(const KSppm kp) {
    // one wavefront per 8x8 pixel block of this rank (same block list as the camera pass)
    const uint32_t tile = kp.tiles[blockIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t qx = (tile & 0xFFFFu) * 8u + (lane & 7u);
    const uint32_t qy = (tile >> 16) * 8u + (lane >> 3);
    if (qx >= kp.W || qy >= kp.H) return;
    const uint32_t i = qy * kp.W + qx;
    trc_CameraRecord& c = kp.cam_rec[i];
    float4* px = reinterpret_cast<float4*>(kp.accum) + i;
    const float4 cached = *px;
    const F3 cache = f3(cached.x, cached.y, cached.z);
    const float frame = (float)kp.frame_count, frame1 = (float)(kp.frame_count + 1);
    if (!c.valid) {
        F3 result = (cache * frame + ld3(c.alternative)) / frame1;
        float4 o; o.x = result.x; o.y = result.y; o.z = result.z; o.w = 1.0f;
        *px = o;
        return;
    }
    const DComplex& cx = *kp.cx;
    const F3 QueryPosition = ld3(c.position), QueryDirection = ld3(c.direction), QueryReflectance = ld3(c.ratio);
    F3 QueryFlux = ld3(c.flux);
    float QueryRadius = c.radius;
    uint32_t QueryPhotonCount = c.photonCount;
    const F3 BBoxMin = f3(cx.box_min[0], cx.box_min[1], cx.box_min[2]);
    const float HashScale = cx.hash_scale;
    const float fN = (float)kHashN;
    F3 rmin = QueryPosition - f3(QueryRadius) - BBoxMin, rmax = QueryPosition + f3(QueryRadius) - BBoxMin;
    const F3 RangeMin = f3(fabsf(rmin.x), fabsf(rmin.y), fabsf(rmin.z)) * HashScale;
    const F3 RangeMax = f3(fabsf(rmax.x), fabsf(rmax.y), fabsf(rmax.z)) * HashScale;
    F3 _Flux = f3(0);
    uint32_t _PhotonCount = 0;
    for (int iz = (int)RangeMin.z; iz <= (int)RangeMax.z; iz++)
        for (int iy = (int)RangeMin.y; iy <= (int)RangeMax.y; iy++)
            for (int ix = (int)RangeMin.x; ix <= (int)RangeMax.x; ix++) {
                const F3 hashIndex = f3((float)ix, (float)iy, (float)iz);
                const float hashed = ph_hash(hashIndex, HashScale, fN);
                const float hx = ph_mod(hashed, fN) - 1.0f, hy = floorf(hashed / fN) - 1.0f;
                uint32_t winner; float Correction;
                if (hx >= 0.0f && hx < fN && hy >= 0.0f && hy < fN) {
                    const uint32_t cell = (uint32_t)hy * kHashN + (uint32_t)hx;
                    const uint32_t mk = kp.mark[cell];
                    if (mk == 0u) continue;                           // PhotonIndex2D.x < 0: empty cell
                    winner = mk - 1u;
                    Correction = (float)kp.count[cell];
                } else { winner = 0u; Correction = 0.0f; }           // out-of-range texture read returns 0
                const trc_PhotonRecord& ph = kp.pho_rec[winner];
                const F3 PhotonPosition = ld3(ph.position);
                const F3 _RangeMin = hashIndex / HashScale + BBoxMin;
                const F3 _RangeMax = (hashIndex + f3(1.0f)) / HashScale + BBoxMin;
                if ((_RangeMin.x < PhotonPosition.x) && (PhotonPosition.x < _RangeMax.x) &&
                    (_RangeMin.y < PhotonPosition.y) && (PhotonPosition.y < _RangeMax.y) &&
                    (_RangeMin.z < PhotonPosition.z) && (PhotonPosition.z < _RangeMax.z)) {
                    const float d = length(PhotonPosition - QueryPosition);
                    if ((d < QueryRadius) && (-dot(QueryDirection, ld3(ph.direction)) > 0.001f)) {
                        _Flux = _Flux + ld3(ph.flux) * Correction;
                        _PhotonCount = (uint32_t)((float)_PhotonCount + Correction);
                    }
                }
            }
    _Flux = _Flux * (QueryReflectance / 3.141592f);                    // BRDF (Lambertian)
    const float alpha = 0.8f;                                          // progressive refinement
    float g = fminf(((float)QueryPhotonCount + (float)_PhotonCount * alpha) / (float)(QueryPhotonCount + _PhotonCount), 1.0f);
    QueryRadius = QueryRadius * sqrtf(g);
    QueryPhotonCount = (uint32_t)((float)QueryPhotonCount + (float)_PhotonCount * alpha);
    QueryFlux = (QueryFlux + _Flux) * g;
    st3(c.flux, QueryFlux);
    c.radius = QueryRadius;
    c.photonCount = QueryPhotonCount;
    float TotalPhotonNum = cx.total_photon_sum;
    TotalPhotonNum += (float)cx.frame_photon_sum;
    F3 color = QueryFlux / (QueryRadius * QueryRadius * 3.141592f * TotalPhotonNum);
    F3 result = (cache * frame + color) / frame1;
    if (is_nan(result.x) || is_nan(result.y) || is_nan(result.z)) result = f3(0);
    float4 o; o.x = result.x; o.y = result.y; o.z = result.z; o.w = 1.0f;
    *px = o;
}

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
    trc_CameraRecord* d_cam = nullptr;
    trc_PhotonRecord* d_pho = nullptr;
    uint32_t* d_mark = nullptr;
    uint32_t* d_count = nullptr;
    DComplex* d_cx = nullptr;
};

void trc_sppm_release(trc_ctx* ctx) {
    if (!ctx || !ctx->sppm) return;
    SppmState* s = ctx->sppm;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(s->d_photon_rng); (void)hipFree(s->d_cam); (void)hipFree(s->d_pho);
    (void)hipFree(s->d_mark); (void)hipFree(s->d_count); (void)hipFree(s->d_cx);
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
    HIP_TRY(ctx, hipMalloc((void**)&s->d_cam, np * sizeof(trc_CameraRecord)));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_pho, nph * sizeof(trc_PhotonRecord)));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_mark, nph * 4));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_count, nph * 4));
    HIP_TRY(ctx, hipMalloc((void**)&s->d_cx, sizeof(DComplex)));
    HIP_TRY(ctx, hipMemsetAsync(s->d_cam, 0, np * sizeof(trc_CameraRecord), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_pho, 0, nph * sizeof(trc_PhotonRecord), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_mark, 0, nph * 4, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(s->d_count, 0, nph * 4, ctx->stream));
    DComplex h;
    std::memset(&h, 0, sizeof h);
    for (int k = 0; k < 3; ++k) { h.key_min[k] = 0xFFFFFFFFu; h.key_max[k] = 0u; }
    HIP_TRY(ctx, hipMemcpyAsync(s->d_cx, &h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));      // `h` is a stack temporary
    hipLaunchKernelGGL(k_sppm_seed, dim3((unsigned)((nph + 255) / 256)), dim3(256), 0, ctx->stream, s->d_photon_rng,
                       (uint32_t)nph, photon_seed);
    HIP_TRY(ctx, hipGetLastError());
    return TRC_OK;
}

trc_status trc_sppm_frames(trc_ctx* ctx, uint32_t n_frames) {
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
    const bool grouped = ctx->comm != nullptr;
    const uint32_t nranks = grouped ? (uint32_t)ctx->nranks : 1u, rank = grouped ? (uint32_t)ctx->rank : 0u;
    const uint32_t np = s->W * s->H, nph = kHashN * kHashN;
    if (nph % (nranks * kBlock)) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "512*512 photons must split evenly over the ranks");
    const uint32_t chunk = nph / nranks;
    { trc_status ts = trc_ensure_tiles(ctx, nranks, rank); if (ts != TRC_OK) return ts; }
    if (ctx->n_tiles == 0) return TRC_OK;

    KSppm kp{};
    kp.ks = ctx->ks; kp.cam = ctx->cam;
    kp.ambient[0] = ctx->ambient[0]; kp.ambient[1] = ctx->ambient[1]; kp.ambient[2] = ctx->ambient[2];
    kp.env_rgb = ctx->d_envmap; kp.env_w = ctx->env_w; kp.env_h = ctx->env_h;
    kp.W = s->W; kp.H = s->H;
    kp.photon_first = rank * chunk;
    kp.tiles = ctx->d_tiles;
    kp.canvas_rng = ctx->d_rng; kp.accum = ctx->d_accum; kp.photon_rng = s->d_photon_rng;
    kp.cam_rec = s->d_cam; kp.pho_rec = s->d_pho; kp.mark = s->d_mark; kp.count = s->d_count; kp.cx = s->d_cx;
    kp.stats = ctx->d_stats;
    const size_t lds = trc_dyn_lds_bytes(ctx, false);
    const bool all_lds = ctx->lds_scene;
    auto rccl_fail = [&](const char* what, int rc) {
        return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
    };
    auto camera_pass = [&]() {
        if (all_lds) hipLaunchKernelGGL((k_sppm_camera<true>), dim3(ctx->n_tiles), dim3(kBlock), lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_sppm_camera<false>), dim3(ctx->n_tiles), dim3(kBlock), lds, ctx->stream, kp);
    };
    for (uint32_t f = 0; f < n_frames; ++f) {
        kp.frame_count = s->frame_count;
        if (s->frame_count == 0) {                      // photonPrepare, AAPLRenderer.mm:860-947
            camera_pass();
            if (grouped) {
                DComplex* cx = s->d_cx;
                int rc = g_rccl.AllReduce(cx->key_min, cx->key_min, 3, kNcclUint32, kNcclMin, ctx->comm, ctx->stream);
                if (rc) return rccl_fail("ncclAllReduce(min)", rc);
                rc = g_rccl.AllReduce(cx->key_max, cx->key_max, 3, kNcclUint32, kNcclMax, ctx->comm, ctx->stream);
                if (rc) return rccl_fail("ncclAllReduce(max)", rc);
            }
            hipLaunchKernelGGL(k_sppm_params, dim3(1), dim3(64), 0, ctx->stream, s->d_cx);
            hipLaunchKernelGGL(k_sppm_radius, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, s->d_cam, np, s->d_cx);
        }
        if (s->frame_count % 2) camera_pass();          // photonWork re-runs the camera pass on odd frames, :953-955
        if (all_lds) hipLaunchKernelGGL((k_sppm_photon<true>), dim3(chunk / kBlock), dim3(kBlock), lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_sppm_photon<false>), dim3(chunk / kBlock), dim3(kBlock), lds, ctx->stream, kp);
        if (grouped) {                                  // every rank needs every photon for hashing + refine
            int rc = g_rccl.AllGather(s->d_pho + (size_t)rank * chunk, s->d_pho, (size_t)chunk * sizeof(trc_PhotonRecord),
                                      kNcclUint8, ctx->comm, ctx->stream);
            if (rc) return rccl_fail("ncclAllGather(photons)", rc);
        }
        HIP_TRY(ctx, hipMemsetAsync(s->d_mark, 0, (size_t)nph * 4, ctx->stream));     // loadAction clear, :785-790
        HIP_TRY(ctx, hipMemsetAsync(s->d_count, 0, (size_t)nph * 4, ctx->stream));
        hipLaunchKernelGGL(k_sppm_hash, dim3(nph / 256), dim3(256), 0, ctx->stream, s->d_pho, s->d_mark, s->d_count, s->d_cx);
        hipLaunchKernelGGL(k_sppm_sum, dim3(128), dim3(256), 0, ctx->stream, s->d_count, s->d_cx);
        hipLaunchKernelGGL(k_sppm_refine, dim3(ctx->n_tiles), dim3(kBlock), 0, ctx->stream, kp);
        hipLaunchKernelGGL(k_sppm_end_frame, dim3(1), dim3(64), 0, ctx->stream, s->d_cx);
        HIP_TRY(ctx, hipGetLastError());
        s->frame_count += 1;
    }
    return TRC_OK;
}

trc_status trc_sppm_download(trc_ctx* ctx, trc_CameraRecord* cam, trc_PhotonRecord* pho, float* mark, float* count, trc_Complex* cx) {
    if (!ctx) return TRC_ERR_INVALID_ARG;
    SppmState* s = ctx->sppm;
    if (!s) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_sppm_download before trc_sppm_init");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t np = (size_t)s->W * s->H, nph = (size_t)kHashN * kHashN;
    if (cam) HIP_TRY(ctx, hipMemcpyAsync(cam, s->d_cam, np * sizeof(trc_CameraRecord), hipMemcpyDeviceToHost, ctx->stream));
    if (pho) HIP_TRY(ctx, hipMemcpyAsync(pho, s->d_pho, nph * sizeof(trc_PhotonRecord), hipMemcpyDeviceToHost, ctx->stream));
    std::vector<uint32_t> hm, hc;
    if (mark) { hm.resize(nph); HIP_TRY(ctx, hipMemcpyAsync(hm.data(), s->d_mark, nph * 4, hipMemcpyDeviceToHost, ctx->stream)); }
    if (count) { hc.resize(nph); HIP_TRY(ctx, hipMemcpyAsync(hc.data(), s->d_count, nph * 4, hipMemcpyDeviceToHost, ctx->stream)); }
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

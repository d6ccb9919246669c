// trc_abi.hip -- kernels and the C ABI of libtracer_amd.so (gfx950 only).
//
// Replaces, for the path-tracing hot path, the reference's Metal host glue
// (-[AAPLRenderer render:] AAPLRenderer.mm:1134-1196) and kernelPathTracing
// (RT_Metal/Metal/Render.metal:495-558).  See include/tracer_abi.h for the boundary and
// DESIGN.md for the data layout and kernel design.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "trc_ctx.hpp"
#include "trc_render_config.hpp"
#include "trc_scene_prep.hpp"

// deterministic stand-in for fillRNG (AAPLRenderer.mm:296-344): texel p = 4 outputs of
// pcg32_srandom_r(seed, p)
__global__ void __launch_bounds__(256) k_seed(uint32_t* rng, uint32_t n_pixels, uint64_t seed) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pixels) return;
    Pcg r;
    r.state = 0;
    r.inc = ((uint64_t)p << 1u) | 1u;
    pcg_next(r);
    r.state += seed;
    pcg_next(r);
    uint4 out;
    out.x = pcg_next(r); out.y = pcg_next(r); out.z = pcg_next(r); out.w = pcg_next(r);
    reinterpret_cast<uint4*>(rng)[p] = out;
}

// blob triangle records from the caller's arrays (trc_scene_prep.hpp): one thread per triangle, 3 gathered 32-byte
// vertices in, 7 float4 out
__global__ void __launch_bounds__(256) k_repack_triangles(const trc_TriangleVertex* __restrict__ verts, const uint32_t* __restrict__ idx,
                                                          uint32_t n_tri, float4* __restrict__ tripos, float4* __restrict__ triattr) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tri) return;
    const trc_TriangleVertex a = verts[idx[3 * t]], b = verts[idx[3 * t + 1]], c = verts[idx[3 * t + 2]];
    tripos[3 * (size_t)t] = make_float4(a.v[0], a.v[1], a.v[2], 0.0f);
    tripos[3 * (size_t)t + 1] = make_float4(b.v[0], b.v[1], b.v[2], 0.0f);
    tripos[3 * (size_t)t + 2] = make_float4(c.v[0], c.v[1], c.v[2], 0.0f);
    triattr[4 * (size_t)t] = make_float4(a.n[0], a.n[1], a.n[2], b.n[0]);
    triattr[4 * (size_t)t + 1] = make_float4(b.n[1], b.n[2], c.n[0], c.n[1]);
    triattr[4 * (size_t)t + 2] = make_float4(c.n[2], a.uv[0], a.uv[1], b.uv[0]);
    triattr[4 * (size_t)t + 3] = make_float4(b.uv[1], c.uv[0], c.uv[1], 0.0f);
}

// BVH::buildNode for the triangles (AAPLRenderer.mm:575-589 + BVH.hh:273-314): the box of the three vertices -- std::max({a, b, c}) /
// std::min({a, b, c}) keep the first of equals -- taken corner by corner through the identity matrix (column sums in the reference's
// order, so a -0 comes out as the host's arithmetic leaves it) into fmin / fmax from +-FLT_MAX (the second operand on a tie, as the
// host's minss / maxss).  One thread per triangle, one 64-byte leaf record out.
__global__ void __launch_bounds__(256) k_triangle_leaves(const trc_TriangleVertex* __restrict__ verts, const uint32_t* __restrict__ idx,
                                                         uint32_t n_tri, trc_BVH* __restrict__ leaves) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= n_tri) return;
    const trc_TriangleVertex a = verts[idx[3 * t]], b = verts[idx[3 * t + 1]], c = verts[idx[3 * t + 2]];
    float ele[2][3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float hi = a.v[k]; if (hi < b.v[k]) hi = b.v[k]; if (hi < c.v[k]) hi = c.v[k];
        float lo = a.v[k]; if (b.v[k] < lo) lo = b.v[k]; if (c.v[k] < lo) lo = c.v[k];
        ele[0][k] = lo; ele[1][k] = hi;
    }
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float x = ele[i][0], y = ele[j][1], z = ele[k][2];
                const float w[3] = {1.0f * x + 0.0f * y + 0.0f * z + 0.0f * 1.0f, 0.0f * x + 1.0f * y + 0.0f * z + 0.0f * 1.0f,
                                    0.0f * x + 0.0f * y + 1.0f * z + 0.0f * 1.0f};
#pragma unroll
                for (int q = 0; q < 3; ++q) { mn[q] = mn[q] < w[q] ? mn[q] : w[q]; mx[q] = mx[q] > w[q] ? mx[q] : w[q]; }
            }
    trc_BVH r;
    memset(&r, 0, sizeof r);
    r.pType = TRC_PRIM_TRIANGLE; r.pIndex = t;
    r.bBOX.mini.x = mn[0]; r.bBOX.mini.y = mn[1]; r.bBOX.mini.z = mn[2];
    r.bBOX.maxi.x = mx[0]; r.bBOX.maxi.y = mx[1]; r.bBOX.maxi.z = mx[2];
    leaves[t] = r;
}

trc_status trc_repack_triangles(trc_ctx* ctx, const trc_scene* s, const DScene& sc, uint32_t* d_blob, trc_BVH* d_tri_leaves) {
    const uint32_t n_tri = s->n_index / 3;
    if (n_tri == 0) return TRC_OK;
    trc_TriangleVertex* d_verts = nullptr;
    uint32_t* d_idx = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_verts, (size_t)s->n_vertex * sizeof(trc_TriangleVertex)));
    if (hipMalloc((void**)&d_idx, (size_t)s->n_index * 4) != hipSuccess) { (void)hipFree(d_verts); return trc_fail(ctx, TRC_ERR_OOM, "hipMalloc triangle indices"); }
    trc_status st = TRC_OK;
    do {
        if (trc_copy_to_device(ctx, d_verts, s->triList, (size_t)s->n_vertex * sizeof(trc_TriangleVertex), ctx->stream) != TRC_OK ||
            trc_copy_to_device(ctx, d_idx, s->idxList, (size_t)s->n_index * 4, ctx->stream) != TRC_OK) { st = trc_fail(ctx, TRC_ERR_HIP, "H2D triangles"); break; }
        hipLaunchKernelGGL(k_repack_triangles, dim3((n_tri + 255) / 256), dim3(256), 0, ctx->stream, d_verts, d_idx, n_tri,
                           reinterpret_cast<float4*>(d_blob + sc.off_tripos), reinterpret_cast<float4*>(d_blob + sc.off_triattr));
        if (d_tri_leaves) hipLaunchKernelGGL(k_triangle_leaves, dim3((n_tri + 255) / 256), dim3(256), 0, ctx->stream, d_verts, d_idx, n_tri, d_tri_leaves);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) { st = trc_fail(ctx, TRC_ERR_HIP, "k_repack_triangles"); break; }
    } while (0);
    (void)hipFree(d_verts); (void)hipFree(d_idx);
    return st;
}

// the counters' rows (stat_row) summed into one row
__global__ void __launch_bounds__(64) k_stats_sum(const unsigned long long* rows, unsigned long long* sum) {
    const uint32_t c = threadIdx.x;
    if (c >= kStatRowStride) return;
    unsigned long long v = 0;
    for (uint32_t r = 0; r < kStatRows; ++r) v += rows[(size_t)r * kStatRowStride + c];
    sum[c] = v;
}

// sort keys of the adaptive launch order: descending cost (shader clocks / 64, clamped to 24 bits), ties in list order.
// Lists that may be split (stride kCostSlots): the last launch may have run an 8x8 block as four quarters (split[i] != 0),
// and a quarter as four sixteenths (qsplit[4 i + q]).  A block's cost as ONE block is then what it measured when it last ran
// whole (whole[i], kept by k_build_launch), or -- a first launch made of quarters only -- an estimate from its slowest part
// on the high side.
constexpr float kQuarterCost = 0.85f;        // a quarter's duration relative to its 8x8 block's: what the plan assumes for
                                             // a block it has not split yet (measured: 0.8-0.9 for the blocks that matter)
constexpr float kQuarterEstimate = 0.65f;
constexpr float kSixteenthTier = 0.8f;       // quarters within this factor of the launch's longest part go on to 2x2 blocks
// qsplit[4 i + q]: bit 0 = quarter q of block i ran as four sixteenths in the last launch, bits 4..7 = sixteenth s of it ran as four
// single pixels (round 6).  for_each_part visits the cost slot of every part block i ran as (trc_ctx.hpp: slot = launch code - 1).
constexpr uint32_t kPixelBit = 4u;
template <class F>
__host__ __device__ __forceinline__ void for_each_part(const uint32_t* qsplit, uint32_t i, F&& f) {
    for (uint32_t q = 0; q < 4u; ++q) {
        const uint32_t m = qsplit[4u * i + q];
        if (!(m & 1u)) { f(q); continue; }
        for (uint32_t s4 = 0; s4 < 4u; ++s4) {
            if ((m >> (kPixelBit + s4)) & 1u) { for (uint32_t p4 = 0; p4 < 4u; ++p4) f(20u + 16u * q + 4u * s4 + p4); }
            else f(4u + 4u * q + s4);
        }
    }
}
// the slowest part of block i that the last launch ran (quarters, the sixteenths of the quarters that were split again, their pixels)
__device__ __forceinline__ uint32_t slowest_part(const uint32_t* cost, const uint32_t* qsplit, uint32_t i) {
    const uint32_t* c = cost + (size_t)i * kCostSlots;
    uint32_t m = 0u;
    for_each_part(qsplit, i, [&](uint32_t slot) { m = max(m, c[slot]); });
    return m;
}
// What a block's duration says about the block.  A SIMD issues from its oldest wavefronts first (tools/probe/age_probe.hip:
// of five wavefronts on a SIMD the first two run as fast as a lone one, the fifth takes 1.8x as long), so the duration a
// wavefront measures is its own work only if it started among the first of its SIMD; started later, the same block lasts up
// to twice as long.  Sorting by the raw durations therefore feeds back on itself: a heavy block that ran first looks light,
// starts late in the next launch, looks heavy again.  The order and the plan work on the SHORTEST duration seen lately
// instead (it grows by 1/64 per launch until a measurement undercuts it, so a scene that changes is followed): config 2
// 20.9 -> 20.4 ms, its shares of 2 / 4 / 8 ranks 12.3 -> 11.1, 8.5 -> 7.3, 5.95 -> 5.6 ms (knob no_cost_filter switches it off).
// One thread per block filters the slots its last launch wrote (the block, its quarters or their sixteenths).
__device__ __forceinline__ void filter_block_costs(const uint32_t* cost, const uint32_t* split, const uint32_t* qsplit, uint32_t stride, uint32_t i,
                                                   uint32_t* filt, uint32_t* whole, const bool fresh) {
    // fresh: what the filter holds are the durations of a cold HEAD (8 samples, row-major, trc_render) -- good enough to order
    // and plan the launch that followed, but no "shortest duration seen lately" of a settled launch: that launch's replace them
    auto slot = [&](uint32_t k) {
        const size_t at = (size_t)i * stride + k;
        const uint32_t f = filt[at], c = cost[at];
        filt[at] = (f && !fresh) ? min(f + (f >> 6) + 1u, c) : c;
    };
    if (stride != kCostSlots || !split[i]) { slot(0u); return; }
    for_each_part(qsplit, i, slot);
    // What the block cost when it last ran WHOLE ranks it for as long as it runs in parts (a value measured under the same
    // conditions as its unsplit neighbours': re-estimating it from the parts every launch made the plan settle elsewhere,
    // config 3 329 -> 344-366 ms).  It only follows the parts DOWN when they say the block is no longer what it was (a camera
    // or a scene that moved on): a quarter lasts 0.8-0.9 of its block, so parts below half of `whole` are another picture's.
    const uint32_t w = whole[i];
    if (w) whole[i] = max(1u, min(w, (uint32_t)((float)slowest_part(filt, qsplit, i) * 2.0f)));
}
// One thread per block: (1) filter the slots its last launch wrote (the block, its quarters or their sixteenths) into `filt`
// (skipped with the knob no_cost_filter: filt == cost then), (2) the block's sort key: its cost as ONE block, descending.
__global__ void __launch_bounds__(256) k_order_keys(const uint32_t* raw, uint32_t* cost, const uint32_t* split, uint32_t* whole, const uint32_t* qsplit,
                                                    uint32_t stride, uint32_t n, uint32_t* keys, uint32_t* vals, const bool filtered, const bool fresh) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (filtered) filter_block_costs(raw, split, qsplit, stride, i, cost, whole, fresh);
    uint32_t c = cost[(size_t)i * stride];
    if (stride == kCostSlots && split[i])
        c = whole[i] ? whole[i] : (uint32_t)((float)slowest_part(cost, qsplit, i) * (1.0f / kQuarterEstimate));
    keys[i] = 0xFFFFFFu - min(c, 0xFFFFFFu);
    vals[i] = i;
}

// Cost-adaptive block size.  A block's samples are a sequential chain, so a launch cannot end before its slowest
// wavefront; a rank that owns about as many 8x8 blocks as the GPU has wavefront slots (a strong-scaled share of a frame)
// lasts exactly that long, while most slots sit idle.  An 8x8 block run as four 4x4 quarters on 16 lanes each ends earlier
// (a quarter waits for 16 pixels' branches, not 64) but occupies four slots and issues ~3x the instructions -- so only
// the blocks that would otherwise decide the launch are split.  Input: the blocks in descending order of their cost as
// whole blocks (keys[r] = 0xFFFFFF - cost, vals[r] = block).  Model of a launch that splits the K most expensive blocks:
//     makespan(K) = max( cost[K], longest part, (sum + (4 * kQuarterCost - 1) * prefix(K)) / slots )
// -- the longest block left whole; the longest part: the slowest quarter / sixteenth MEASURED in the previous launch, and
// kQuarterCost x the most expensive block that launch ran whole if K reaches it; the work over the wavefront slots.  One
// workgroup picks the smallest K <= k_max that minimises it: a launch with many more blocks than slots gets K = 0 from
// the third term, an eighth of a 1080p frame splits the few blocks above the longest part.
// Second level: where wavefront slots are still idle after that (entries < slots), the quarters within kSixteenthTier of the
// launch's longest part -- the ones the launch now ends on -- run as four 2x2 sixteenths on 4 lanes each in the next launch;
// plan[3] = the threshold a quarter's duration must reach, plan[4] = how many do, plan[1] = the entries of the launch.
// per RANK r of the sorted order (block i = vals[r]), gathered by one thread each so that the one-workgroup planner below reads
// dense arrays: part[r] = the slowest part the last launch ran of it (0: it ran whole), rawv[r] = its measured duration when
// it ran whole (else -1), quart[4 r + q] = quarter q's cost when the block ran in parts and that quarter ran as ONE (else 0;
// 0xFFFFFFFF: it already ran as sixteenths)
__global__ void __launch_bounds__(256) k_plan_gather(const uint32_t* vals, const uint32_t* split, const uint32_t* cost, const uint32_t* qsplit,
                                                     const uint32_t* raw, uint32_t n, uint32_t* part, float* rawv, uint32_t* quart, uint32_t* sixt) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t i = vals[r];
    const bool sp = split[i] != 0u;
    part[r] = sp ? slowest_part(cost, qsplit, i) : 0u;
    rawv[r] = sp ? -1.0f : (float)raw[(size_t)i * kCostSlots];
#pragma unroll
    for (uint32_t q = 0; q < 4u; ++q) {
        const uint32_t m = sp ? qsplit[4u * i + q] : 0u;
        quart[4u * r + q] = !sp ? 0u : ((m & 1u) ? 0xFFFFFFFFu : cost[(size_t)i * kCostSlots + q]);
        // third level: sixteenth s of a quarter that ran as sixteenths -- its cost as ONE sixteenth (0xFFFFFFFF: it already ran as
        // pixels; 0: its quarter ran whole, nothing is known about it)
        for (uint32_t s4 = 0; s4 < 4u; ++s4)
            sixt[16u * r + 4u * q + s4] = !(m & 1u) ? 0u : (((m >> (kPixelBit + s4)) & 1u) ? 0xFFFFFFFFu : cost[(size_t)i * kCostSlots + 4u + 4u * q + s4]);
    }
}
// exclusive prefix sum over the 1024 threads of the workgroup (wave shuffles, then the 16 wave totals); returns the total
__device__ __forceinline__ double block_scan_1024(double v, double* s_wave /* [16] */, double& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    double inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const double o = __shfl_up(inc, off, 64); if ((int)lane >= off) inc += o; }
    if (lane == 63u) s_wave[wave] = inc;
    __syncthreads();
    double base = 0.0, tot = 0.0;
#pragma unroll
    for (uint32_t w = 0; w < 16u; ++w) { const double x = s_wave[w]; if (w < wave) base += x; tot += x; }
    __syncthreads();
    total = tot;
    return base + inc - v;
}
__global__ void __launch_bounds__(1024) k_plan_split(const uint32_t* keys, const uint32_t* part, const float* rawv, const uint32_t* quart, const uint32_t* sixt,
                                                     uint32_t n, uint32_t k_max, const uint32_t slots, const uint32_t max_entries, uint32_t* plan, uint32_t* launch) {
    __shared__ double s_wave[16];
    __shared__ float s_best[1024];
    __shared__ uint32_t s_k[1024], s_q[1024], s_first[1024];
    __shared__ float s_raw[1024], s_est[1024];
    const uint32_t t = threadIdx.x, per = (n + 1023u) / 1024u;
    const uint32_t lo = min(n, t * per), hi = min(n, lo + per);
    double local = 0.0;
    float raw_sum = 0.0f, est_sum = 0.0f;          // blocks the last launch ran whole: measured durations / filtered costs
    uint32_t q_max = 0u, first_whole = 0xFFFFFFFFu;
    for (uint32_t r = lo; r < hi; ++r) {
        const float c = (float)(0xFFFFFFu - keys[r]);
        local += (double)c;
        const uint32_t pr = part[r];
        if (pr) q_max = max(q_max, pr);
        else {
            if (first_whole == 0xFFFFFFFFu) first_whole = r;
            raw_sum += rawv[r]; est_sum += c;
        }
    }
    double total = 0.0;
    double prefix = block_scan_1024(local, s_wave, total);          // cost of the blocks before rank `lo`
    s_q[t] = q_max; s_first[t] = first_whole; s_raw[t] = raw_sum; s_est[t] = est_sum;
    __syncthreads();
    for (uint32_t off = 512u; off > 0u; off >>= 1) {
        if (t < off) { s_q[t] = max(s_q[t], s_q[t + off]); s_first[t] = min(s_first[t], s_first[t + off]); s_raw[t] += s_raw[t + off]; s_est[t] += s_est[t + off]; }
        __syncthreads();
    }
    // the costs are what a block takes when it starts first on its SIMD; a wavefront slot is held for the measured
    // duration: the work term scales by the ratio of the two over the blocks that ran whole
    const double held = s_est[0] > 0.0f ? (double)fminf(fmaxf(s_raw[0] / s_est[0], 1.0f), 4.0f) : 1.0;      // slot time per unit of cost
    const float part_seen = (float)s_q[0];
    const uint32_t r_whole = s_first[0];            // most expensive block the previous launch ran whole
    __syncthreads();
    if (t == 0) plan[2] = (uint32_t)min(held * total / (double)slots, 4294967295.0);   // diagnostic: work / slots of the unsplit launch
    const float quarter_new = r_whole < n ? kQuarterCost * (float)(0xFFFFFFu - keys[r_whole]) : 0.0f;
    const double extra = 4.0 * (double)kQuarterCost - 1.0;
    float best = 3.0e38f;
    uint32_t best_k = 0;
    auto candidate = [&](uint32_t k, float whole) {       // split ranks 0 .. k-1
        const float pt = k == 0u ? 0.0f : (k > r_whole ? fmaxf(part_seen, quarter_new) : part_seen);
        const float work = (float)(held * (total + extra * prefix) / (double)slots);
        const float m = fmaxf(fmaxf(whole, pt), work);
        if (m < best) { best = m; best_k = k; }
    };
    // fewer blocks than wavefront slots: the parts must not push the launch into a second round of wavefronts
    if (n < slots) k_max = min(k_max, (slots - n) / 3u);
    for (uint32_t k = lo; k < hi && k <= k_max; ++k) {
        candidate(k, (float)(0xFFFFFFu - keys[k]));
        prefix += (double)(0xFFFFFFu - keys[k]);
    }
    if (hi == n && lo < hi && n <= k_max) candidate(n, 0.0f);          // ... and "every block as quarters"
    s_best[t] = best; s_k[t] = best_k;
    __syncthreads();
    for (uint32_t off = 512u; off > 0u; off >>= 1) {
        if (t < off) {
            const float a = s_best[t], b = s_best[t + off];
            if (b < a || (b == a && s_k[t + off] < s_k[t])) { s_best[t] = b; s_k[t] = s_k[t + off]; }
        }
        __syncthreads();
    }
    // the plan feeds back on itself (parts that end earlier lower the bar for the next launch's split): K moves by at most
    // half of its previous value (+ 16) per launch, so the launch time settles instead of swinging
    const uint32_t k_prev = plan[6];               // 0xFFFFFFFF: the previous launch was not planned from measurements
    const uint32_t K = k_prev == 0xFFFFFFFFu ? s_k[0] : min(max(s_k[0], k_prev - k_prev / 2u), k_prev + k_prev / 2u + 16u);
    __syncthreads();
    // second level: the quarters of blocks that were quarters last launch too and lasted at least `tier`, as long as every
    // entry of the launch still gets a wavefront slot of its own
    // ... or, with more entries than slots, as long as the launch is bound by its longest part and not by its work
    const uint32_t entries1 = n + 3u * K;
    const float work_bound = (float)(held * total / (double)slots);    // the unsplit launch's work over the slots
    const bool tail_bound = entries1 < slots || work_bound < 0.7f * part_seen;
    const uint32_t tier = tail_bound && part_seen > 0.0f ? max(1u, (uint32_t)(kSixteenthTier * part_seen)) : 0xFFFFFFFFu;
    uint32_t have = 0u, want = 0u;                  // quarters that already run as sixteenths / that would join them
    for (uint32_t j = t; j < 4u * K; j += 1024u) {
        const uint32_t cq = quart[j];
        if (cq == 0u) continue;                                        // becomes quarters now: their durations are not known yet
        if (cq == 0xFFFFFFFFu) have++;
        else if (cq >= tier) want++;
    }
    s_k[t] = have; s_q[t] = want;
    __syncthreads();
    for (uint32_t off = 512u; off > 0u; off >>= 1) { if (t < off) { s_k[t] += s_k[t + off]; s_q[t] += s_q[t + off]; } __syncthreads(); }
    const uint32_t room = min(max_entries, max(slots, entries1 + slots / 8u)) - entries1;      // entries the sixteenths may add
    uint32_t K2 = s_k[0] + s_q[0], use_tier = tier, keep = 1u;
    if (12u * K2 > room) { K2 = s_k[0]; use_tier = 0xFFFFFFFFu; }                 // no new ones
    if (12u * K2 > room) { K2 = 0u; keep = 0u; }                                  // not even the old ones: back to quarters
    __syncthreads();
    // third level (round 6): the sixteenths of quarters that were sixteenths last launch too and lasted at least `tier` run as four
    // single pixels -- one lane, the floor of a pixel's sample chain -- under the same condition (the launch ends on its longest
    // part while wavefront slots are idle) and out of what room the second level left
    uint32_t have3 = 0u, want3 = 0u;
    if (keep) for (uint32_t j = t; j < 16u * K; j += 1024u) {
        const uint32_t cs = sixt[j];
        if (cs == 0u) continue;
        if (cs == 0xFFFFFFFFu) have3++;
        else if (cs >= tier) want3++;
    }
    s_k[t] = have3; s_q[t] = want3;
    __syncthreads();
    for (uint32_t off = 512u; off > 0u; off >>= 1) { if (t < off) { s_k[t] += s_k[t + off]; s_q[t] += s_q[t + off]; } __syncthreads(); }
    const uint32_t room3 = room - 12u * K2;
    uint32_t K3 = s_k[0] + s_q[0], tier3 = tier, keep3 = 1u;
    if (3u * K3 > room3) { K3 = s_k[0]; tier3 = 0xFFFFFFFFu; }
    if (3u * K3 > room3) { K3 = 0u; keep3 = 0u; }
    if (t == 0) {
        plan[0] = K; plan[3] = use_tier; plan[4] = K2; plan[7] = keep;
        plan[5] = 0u;                                                  // k_build_launch's cursor into the part region
        plan[6] = K;
        plan[8] = tier3; plan[9] = K3; plan[10] = keep3;
        plan[1] = entries1 + 12u * K2 + 3u * K3;
    }
    // the part region is sized for more parts than k_build_launch may make: entries it does not claim name no pixels
    for (uint32_t j = t; j < 4u * K + 12u * K2 + 3u * K3; j += 1024u) launch[j] = kLaunchIndexMask;
}
// the launch list of a plan: first the parts of ranks 0 .. K-1 (the longest blocks: quarters, or sixteenths of the quarters
// the second level picked), then the other blocks whole, longest first.  The parts take their places with an atomic cursor
// (any order will do among them: they all start in the first round of wavefronts); a part region sized for more sixteenths
// than were made is padded with entries that name no pixels.
__global__ void __launch_bounds__(256) k_build_launch(const uint32_t* keys, const uint32_t* vals, uint32_t* cost, uint32_t n, uint32_t* plan,
                                                      uint32_t* launch, uint32_t* split, uint32_t* whole, uint32_t* qsplit, uint32_t* qwhole, uint32_t* swhole, const bool filtered) {
    const uint32_t r = blockIdx.x * 256u + threadIdx.x;
    if (r >= n) return;
    const uint32_t K = plan[0], tier = plan[3], keep = plan[7], tier3 = plan[8], keep3 = plan[10], i = vals[r];
    const uint32_t region = 4u * K + 12u * plan[4] + 3u * plan[9];
    if (r < K) {
        const bool was_split = split[i] != 0u;
        uint32_t codes[64], n_parts = 0u;          // per quarter: 1 entry, or per sixteenth 1 or 4
        uint32_t* c = cost + (size_t)i * kCostSlots;
        for (uint32_t q = 0; q < 4u; ++q) {
            bool again = false;
            const uint32_t m = was_split ? qsplit[4u * i + q] : 0u;
            const bool was = (m & 1u) != 0u;
            if (was_split) {
                const uint32_t cq = was ? qwhole[4u * i + q] : c[q];
                again = was ? keep != 0u : cq >= tier;
                if (again && !was) {
                    qwhole[4u * i + q] = max(1u, cq);
                    if (filtered) for (uint32_t s4 = 0; s4 < 4u; ++s4) c[4u + 4u * q + s4] = 0u;   // nothing known yet
                } else if (!again && was && filtered) c[q] = qwhole[4u * i + q];    // back to one quarter: what it took as one
            }
            uint32_t mnew = again ? 1u : 0u;
            if (!again) { codes[n_parts++] = 1u + q; qsplit[4u * i + q] = 0u; continue; }
            for (uint32_t s4 = 0; s4 < 4u; ++s4) {
                // a sixteenth goes on to pixels only once it has been MEASURED as a sixteenth (its quarter ran as sixteenths before)
                const uint32_t at16 = 16u * i + 4u * q + s4;
                const bool was3 = was && ((m >> (kPixelBit + s4)) & 1u);
                bool again3 = false;
                if (was) {
                    const uint32_t cs = was3 ? swhole[at16] : c[4u + 4u * q + s4];
                    again3 = was3 ? keep3 != 0u : cs >= tier3;
                    if (again3 && !was3) {
                        swhole[at16] = max(1u, cs);
                        if (filtered) for (uint32_t p4 = 0; p4 < 4u; ++p4) c[20u + 16u * q + 4u * s4 + p4] = 0u;
                    } else if (!again3 && was3 && filtered) c[4u + 4u * q + s4] = swhole[at16];     // back to one sixteenth
                }
                if (again3) { mnew |= 1u << (kPixelBit + s4); for (uint32_t p4 = 0; p4 < 4u; ++p4) codes[n_parts++] = 21u + 16u * q + 4u * s4 + p4; }
                else codes[n_parts++] = 5u + 4u * q + s4;
            }
            qsplit[4u * i + q] = mnew;
        }
        const uint32_t at = atomicAdd(&plan[5], n_parts);
        for (uint32_t j = 0; j < n_parts; ++j) if (at + j < region) launch[at + j] = i | (codes[j] << kLaunchCodeShift);
        if (!was_split) {
            whole[i] = max(1u, 0xFFFFFFu - keys[r]);     // what it cost as one block, for as long as it runs in parts
            if (filtered) for (uint32_t k = 0; k < 4u; ++k) c[k] = 0u;               // the quarters: nothing known yet
        }
        split[i] = 1u;
    } else {
        launch[region + (r - K)] = i;
        if (filtered && split[i]) cost[(size_t)i * kCostSlots] = whole[i];                                       // back to one block: slot 0 was its first quarter's
        split[i] = 0u;
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) qsplit[4u * i + q] = 0u;
    }
}
// every block as four quarters (a first launch of few blocks: nothing is known about their costs yet)
__global__ void __launch_bounds__(256) k_build_launch_all_quarters(uint32_t n, uint32_t* plan, uint32_t* launch, uint32_t* split, uint32_t* whole, uint32_t* qsplit) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i == 0) { plan[0] = n; plan[1] = 4u * n; plan[2] = 0u; plan[3] = 0xFFFFFFFFu; plan[4] = 0u; plan[5] = 4u * n; plan[6] = 0xFFFFFFFFu; plan[7] = 1u;
                  plan[8] = 0xFFFFFFFFu; plan[9] = 0u; plan[10] = 1u; }
    if (i >= n) return;
#pragma unroll
    for (uint32_t j = 0; j < 4u; ++j) { launch[4u * i + j] = i | ((j + 1u) << kLaunchCodeShift); qsplit[4u * i + j] = 0u; }
    split[i] = 1u;
    whole[i] = 0u;            // never measured as one block: k_order_keys estimates it from the slowest quarter
}

// ---- output stage (fragmentShader, Render.metal:29-75): exposure sums, then ACES to 8 bit
__global__ void __launch_bounds__(256) k_tonemap_sum(const float4* accum, uint32_t n, unsigned long long* sums /* [3] */) {
    unsigned long long s[3] = {0, 0, 0};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const float4 px = accum[i];
        const float c[3] = {px.x, px.y, px.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float v = c[k];
            if (!(v > 0.0f)) v = 0.0f;
            if (v > 1048576.0f) v = 1048576.0f;
            s[k] += (unsigned long long)(v * 65536.0f + 0.5f);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_xor(s[k], off, 64);
        if ((threadIdx.x & 63u) == 0) atomicAdd(&sums[k], s[k]);
    }
}
__global__ void __launch_bounds__(256) k_tonemap(const float4* accum, uint32_t W, uint32_t H, float expose, uchar4* out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= W * H) return;
    const uint32_t y = i / W, x = i - y * W;
    const float4 px = accum[(size_t)(H - 1u - y) * W + x];
    const float A = 2.51f, B = 0.03f, Cc = 2.43f, D = 0.59f, E = 0.14f;
    const float c[3] = {px.x, px.y, px.z};
    uint32_t o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float col = c[k] * expose;
        float t = (col * (A * col + B)) / (col * (Cc * col + D) + E);        // ACESTone, Render.hh:78-89
        if (!(t > 0.0f)) t = 0.0f;
        if (t > 1.0f) t = 1.0f;
        o[k] = (uint32_t)(t * 255.0f + 0.5f);
    }
    out[i] = make_uchar4((unsigned char)o[0], (unsigned char)o[1], (unsigned char)o[2], 255);
}

// Scene::hit test hook: one lane per ray, full HitRecord + per-ray traversal counters.  STATS = true is the plain
// round with the exact counters; STATS = false is the traversal the render kernels run (speculative round,
// dev_intersect.hpp::trav_iter), counters left at zero.
template <bool LDS, bool ANY, bool STATS>
__global__ void __launch_bounds__(kBlock) k_trace(const KTrace kp) {
    const DScene& sc = kp.ks.sc;
    const uint32_t* small_base = stage_scene(sc);
    uint32_t* stack = lane_stack(sc);
    uint32_t* lvstack = lane_lvstack(sc);
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= kp.n) return;
    const trc_ray in = kp.rays[i];
    SceneRef S = make_scene_ref(sc, small_base);
    Ray ray = make_ray(f3(in.origin[0], in.origin[1], in.origin[2]), f3(in.direction[0], in.direction[1], in.direction[2]));
    HitRec rec;
    hit_init(rec);
    TravCounters cnt;
    counters_zero(cnt);
    constexpr bool kOrderFree = ANY && !STATS;      // the production kernels' shadow-ray walk: only the answer is defined
    const F3 root_min = f3(kp.ks.root_box[0], kp.ks.root_box[1], kp.ks.root_box[2]), root_max = f3(kp.ks.root_box[3], kp.ks.root_box[4], kp.ks.root_box[5]);
    bool h;
    if (kOrderFree) h = scene_occluded<LDS, false, false>(S, root_min, root_max, ray, in.tmax, stack, sc.stack_lds);
    else h = scene_hit<LDS, STATS, ANY, true, false, false, (STATS || ANY) ? 0 : (LDS ? TRC_DEFER_LDS : TRC_DEFER_GLOBAL)>(S, root_min, root_max, ray, rec, in.tmax, stack, lvstack, cnt);
    trc_hit o;
    memset(&o, 0, sizeof o);
    o.hit = h ? 1 : 0;
    o.pType = -1;
    if (h && !kOrderFree) {
        o.pType = (int32_t)(rec.tag >> kTagIndexBits);
        o.pIndex = rec.tag & kTagIndexMask;
        o.t = rec.t;
        o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
        o.gn[0] = rec.gn.x; o.gn[1] = rec.gn.y; o.gn[2] = rec.gn.z;
        o.sn[0] = rec.sn.x; o.sn[1] = rec.sn.y; o.sn[2] = rec.sn.z;
        o.uv[0] = rec.uv.x; o.uv[1] = rec.uv.y;
        o.material = rec.material;
        o.PDF = rec.PDF;
    }
    o.n_descend = cnt.n_descend;
    o.n_return = cnt.n_return;
    o.n_leaf = cnt.leaf[0] + cnt.leaf[1] + cnt.leaf[2] + cnt.leaf[3];
    kp.hits[i] = o;
}

#ifdef TRC_TEST_HOOKS      // libtracer_amd_hooks.so only (include/tracer_test_hooks.h)
// trc_div_by_test: a[i] / b[i] through the guarded shared-divisor path (three numerators a, -a, a * 0.75 on one divisor) and
// through the plain division
__global__ void __launch_bounds__(256) k_div_by_test(const float* a, const float* b, uint32_t n, float* fast, float* plain) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float av = a[i], bv = b[i];
    const GuardedDivBy d = guarded_div_by(bv);
    const F3 q = guarded_div(f3(av, -av, av * 0.75f), d);
    fast[3 * i] = q.x; fast[3 * i + 1] = q.y; fast[3 * i + 2] = q.z;
    plain[3 * i] = av / bv; plain[3 * i + 1] = -av / bv; plain[3 * i + 2] = (av * 0.75f) / bv;
}

// trc_unary_test: rcp_cr / sqrt_cr / rsqrt_cr (dev_vec.hpp) against the compiler's 1.0f / x, sqrtf(x), 1.0f / sqrtf(x) over a
// range of BIT PATTERNS; counts the operands whose results differ (NaN == NaN) and keeps the smallest one
__global__ void __launch_bounds__(256) k_unary_test(uint32_t op, uint32_t first, uint64_t count, unsigned long long* out) {
    unsigned long long bad = 0, first_bad = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < count; i += (uint64_t)gridDim.x * 256u) {
        const uint32_t bits = first + (uint32_t)i;
        const float x = __uint_as_float(bits);
        float a, b;
        if (op == 0u) { a = rcp_cr(x); b = 1.0f / x; }
        else if (op == 1u) { a = sqrt_cr(x); b = sqrtf(x); }
#if TRC_WAVE_GUARDS
        else if (op == 3u) { a = div_const(x, div_by_pi()); b = x / kPi; }
        else if (op == 4u) { a = div_const(x, div_by_sqr001()); b = x / (0.01f * 0.01f); }
        else if (op == 5u) { a = div_const(x, div_by_sqr002()); b = x / (0.02f * 0.02f); }
        else if (op == 6u) { a = div_const(x, div_by_sqr01()); b = x / (0.1f * 0.1f); }
#endif
        else { a = rsqrt_cr(x); b = 1.0f / sqrtf(x); }
        const bool same = __float_as_uint(a) == __float_as_uint(b) || (a != a && b != b);
        if (!same) { bad++; first_bad = min(first_bad, (unsigned long long)bits); }
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], first_bad); }
}
#endif  // TRC_TEST_HOOKS

// ======================================================================= host side
namespace {


}  // namespace

namespace {

constexpr bool kAutoSmallBlocks = true;       // decided by measurement (tools/small_blocks_bench.py, tile_balance.py); DESIGN.md section 5

inline trc_status fail(trc_ctx* ctx, trc_status st, const std::string& msg) { return trc_fail(ctx, st, msg); }

hipEvent_t get_event(trc_ctx* ctx) { return trc_get_event(ctx); }

// folds the per-launch event pairs that have already completed into kernel_ms without waiting (oldest first; the
// stream is in order, so the first unfinished pair ends the scan).  Called from trc_render, so a host that never
// synchronises through trc_synchronize / trc_get_stats (one launch + one download per frame, the reference's own
// pattern) keeps a bounded list.
void collect_finished_events(trc_ctx* ctx) {
    size_t done = 0;
    for (; done < ctx->pending.size(); ++done) {
        const hipError_t q = hipEventQuery(ctx->pending[done].second);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }      // "not ready" is reported through the sticky error too
        if (q != hipSuccess) break;                                        // a real error stays for the caller's next check
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->pending[done].first, ctx->pending[done].second) == hipSuccess) ctx->kernel_ms += ms;
        ctx->event_pool.push_back(ctx->pending[done].first);
        ctx->event_pool.push_back(ctx->pending[done].second);
    }
    ctx->pending.erase(ctx->pending.begin(), ctx->pending.begin() + (ptrdiff_t)done);
    for (done = 0; done < ctx->pending_sched.size(); ++done) {                 // the launch-list kernels' pairs: schedule_ms
        const hipError_t q = hipEventQuery(ctx->pending_sched[done].second);
        if (q == hipErrorNotReady) { (void)hipGetLastError(); break; }
        if (q != hipSuccess) break;
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, ctx->pending_sched[done].first, ctx->pending_sched[done].second) == hipSuccess) ctx->schedule_ms += ms;
        ctx->event_pool.push_back(ctx->pending_sched[done].first);
        ctx->event_pool.push_back(ctx->pending_sched[done].second);
    }
    ctx->pending_sched.erase(ctx->pending_sched.begin(), ctx->pending_sched.begin() + (ptrdiff_t)done);
}

// drains finished per-launch event pairs into kernel_ms (call after a stream sync)
void collect_events(trc_ctx* ctx) {
    for (auto& pr : ctx->pending) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->kernel_ms += ms;
        ctx->event_pool.push_back(pr.first);
        ctx->event_pool.push_back(pr.second);
    }
    ctx->pending.clear();
    for (auto& pr : ctx->pending_sched) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->schedule_ms += ms;
        ctx->event_pool.push_back(pr.first);
        ctx->event_pool.push_back(pr.second);
    }
    ctx->pending_sched.clear();
}


// Repack the reference arrays into the device layout (dev_scene.hpp) and validate the tree.
trc_status build_blob(trc_ctx* ctx, const trc_scene* s, std::vector<uint32_t>& blob, uint64_t& blob_total, KScene& ks) {
    if (!s || !s->bvhList || s->n_bvh < 3 || (s->n_bvh & 1u) == 0) return fail(ctx, TRC_ERR_INVALID_ARG, "scene: need >= 2 leaves (n_bvh odd, >= 3)");
    { trc_status st = validate_primitives(ctx, s); if (st != TRC_OK) return st; }
    const trc_BVH* nodes = s->bvhList;
    const uint32_t n = s->n_bvh;
    if (nodes[0].pType != TRC_PRIM_BVH) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: root is not an interior node");

    // BFS over interior nodes: compact ids, depth, validation
    std::vector<uint32_t> interior_id(n, 0xFFFFFFFFu), order, depth_of(n, 0);
    order.reserve(n / 2 + 1);
    order.push_back(0);
    interior_id[0] = 0;
    uint32_t visited = 1, max_leaf_depth = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const uint32_t i = order[h];
        const uint32_t kids[2] = {nodes[i].left, nodes[i].right};
        if (kids[0] == kids[1]) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: left == right");
        for (uint32_t c : kids) {
            if (c == 0 || c >= n) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: child index out of range");
            if (++visited > n) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: cycle");
            depth_of[c] = depth_of[i] + 1;
            if (nodes[c].pType == TRC_PRIM_BVH) {
                if (interior_id[c] != 0xFFFFFFFFu) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: node reached twice");
                interior_id[c] = (uint32_t)order.size();
                order.push_back(c);
            } else {
                max_leaf_depth = std::max(max_leaf_depth, depth_of[c]);
                trc_status st = validate_leaf(ctx, s, nodes[c]);
                if (st != TRC_OK) return st;
            }
        }
    }
    if (visited != n) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: unreachable nodes");
    if (max_leaf_depth > TRC_MAX_BVH_DEPTH) return fail(ctx, TRC_ERR_BVH_INVALID, "bvh: deeper than TRC_MAX_BVH_DEPTH");

    const uint32_t n_interior = (uint32_t)order.size();
    DScene sc{};
    uint64_t total = 0;
    { trc_status st = layout_scene(ctx, s, n_interior, sc, total); if (st != TRC_OK) return st; }
    plan_lds(sc, max_leaf_depth, true);
    blob.assign((size_t)sc.off_tripos, 0u);      // analytic primitives, materials, fat nodes; triangle records are made on the device
    blob_total = total;

    auto tag_of = [&](uint32_t c) -> uint32_t {
        if (nodes[c].pType == TRC_PRIM_BVH) return (kTagInterior << kTagIndexBits) | interior_id[c];
        return ((uint32_t)nodes[c].pType << kTagIndexBits) | nodes[c].pIndex;
    };
    prep_parallel_for(n_interior, [&](size_t kb, size_t ke) {
        for (size_t k = kb; k < ke; ++k) {
            const trc_BVH& nd = nodes[order[k]];
            const trc_AABB& L = nodes[nd.left].bBOX;
            const trc_AABB& R = nodes[nd.right].bBOX;
            uint32_t* q = &blob[sc.off_nodes + k * kNodeDwords];
            q[0] = f2u(L.mini.x); q[1] = f2u(L.mini.y); q[2] = f2u(L.mini.z); q[3] = f2u(L.maxi.x);
            q[4] = f2u(L.maxi.y); q[5] = f2u(L.maxi.z); q[6] = f2u(R.mini.x); q[7] = f2u(R.mini.y);
            q[8] = f2u(R.mini.z); q[9] = f2u(R.maxi.x); q[10] = f2u(R.maxi.y); q[11] = f2u(R.maxi.z);
            q[12] = 0; q[13] = 0; q[14] = tag_of(nd.left); q[15] = tag_of(nd.right);
        }
    });
    fill_primitives(s, sc, blob.data());
    ks.sc = sc;
    const trc_AABB& rb = nodes[0].bBOX;
    ks.root_box[0] = rb.mini.x; ks.root_box[1] = rb.mini.y; ks.root_box[2] = rb.mini.z;
    ks.root_box[3] = rb.maxi.x; ks.root_box[4] = rb.maxi.y; ks.root_box[5] = rb.maxi.z;
    return TRC_OK;
}

// tiles owned by `rank` of `nranks`, in row-major order.  Workgroups are dealt to the 8 XCDs round-robin
// (blockIdx % 8), so neighbouring tiles -- similar cost: the same object fills them -- land on different XCDs and
// every XCD receives the same mix.  Measured: handing each XCD a contiguous band of the image instead (the
// "L2-friendly" order) costs 22 % on the Cornell scene and 44 % on the 1 M-triangle scene, because the XCD whose
// band holds the glass / mesh pixels finishes long after the others; 8x8-tile blocks per XCD sit in between.
std::vector<uint32_t> make_tiles(uint32_t W, uint32_t H, uint32_t nranks, uint32_t rank, uint32_t view_height, uint32_t blk_shift) {
    // ownership is decided per TRC_TILE x TRC_TILE tile; the launch unit is the pixel block of one wavefront:
    // 8x8 (blk_shift 3) or, for launches with too few blocks to fill the GPU, 4x4 on 16 lanes (blk_shift 2)
    const uint32_t e = 1u << blk_shift;
    const uint32_t bw = (W + e - 1) / e, bh = (H + e - 1) / e;
    std::vector<uint32_t> mine;
    for (uint32_t by = 0; by < bh; ++by)
        for (uint32_t bx = 0; bx < bw; ++bx)
            if ((bx * e / TRC_TILE + by * e / TRC_TILE) % nranks == rank) mine.push_back(bx | (by << 16));
    if (view_height != 0 && view_height < H) {
        // stacked views: walk the rows of ALL views together (row within the view first), so the launch ends on the
        // last rows of every view like a single-view launch does.  View after view, the expensive blocks of the final
        // view would start a few ms before the end of the list and run on alone (measured 28-31 ms instead of 25).
        std::stable_sort(mine.begin(), mine.end(), [&](uint32_t a, uint32_t b) {
            const uint32_t ra = ((a >> 16) * e) % view_height / e, rb = ((b >> 16) * e) % view_height / e;
            return ra < rb;
        });
    }
    return mine;
}

}  // namespace

Rccl g_rccl;

hipEvent_t trc_get_event(trc_ctx* ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// the one piece of process-wide state: resolved once, under a lock (contexts may be created from several threads)
static std::mutex g_rccl_lock;
bool trc_load_rccl(std::string& err) {
    std::lock_guard<std::mutex> guard(g_rccl_lock);
    Rccl& r = g_rccl;
    if (r.ready) return true;
    if (r.handle) { dlclose(r.handle); r = Rccl{}; }        // an earlier attempt found the library but not every symbol
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (r.handle) break;
    }
    if (!r.handle) { err = std::string("dlopen(librccl) failed: ") + dlerror(); return false; }
    r.GetUniqueId = (int (*)(void*))dlsym(r.handle, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void**, int, IdBlob, int))dlsym(r.handle, "ncclCommInitRank");
    r.Reduce = (int (*)(const void*, void*, size_t, int, int, int, void*, hipStream_t))dlsym(r.handle, "ncclReduce");
    r.AllReduce = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))dlsym(r.handle, "ncclAllReduce");
    r.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(r.handle, "ncclAllGather");
    r.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(r.handle, "ncclSend");
    r.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(r.handle, "ncclRecv");
    r.GroupStart = (int (*)())dlsym(r.handle, "ncclGroupStart");
    r.GroupEnd = (int (*)())dlsym(r.handle, "ncclGroupEnd");
    r.CommDestroy = (int (*)(void*))dlsym(r.handle, "ncclCommDestroy");
    r.GetErrorString = (const char* (*)(int))dlsym(r.handle, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.Reduce || !r.AllReduce || !r.AllGather || !r.CommDestroy) {
        err = "librccl: missing symbols";
        dlclose(r.handle);
        r = Rccl{};
        return false;
    }
    r.ready = true;
    return true;
}

// ----------------------------------------------------------------------- transfers through pinned staging (trc_ctx.hpp)
static trc_status xfer_ready(trc_ctx* ctx) {
    if (ctx->h_xfer) return TRC_OK;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_xfer, 2 * kXferChunk, hipHostMallocDefault));
    for (hipEvent_t& e : ctx->ev_xfer) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return TRC_OK;
}
trc_status trc_copy_to_host(trc_ctx* ctx, void* host, const void* dev, size_t bytes, hipStream_t st) {
    if (bytes == 0) return TRC_OK;
    { const trc_status rs = xfer_ready(ctx); if (rs != TRC_OK) return rs; }
    size_t prev_off = 0, prev_n = 0;
    int slot = 0;
    for (size_t off = 0; off < bytes; off += kXferChunk, slot ^= 1) {
        const size_t n = std::min(kXferChunk, bytes - off);
        HIP_TRY(ctx, hipMemcpyAsync(ctx->h_xfer + slot * kXferChunk, static_cast<const char*>(dev) + off, n, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_xfer[slot], st));
        if (prev_n) {                                     // the chunk before, while this one is on its way
            HIP_TRY(ctx, hipEventSynchronize(ctx->ev_xfer[slot ^ 1]));
            std::memcpy(static_cast<char*>(host) + prev_off, ctx->h_xfer + (slot ^ 1) * kXferChunk, prev_n);
        }
        prev_off = off; prev_n = n;
    }
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_xfer[slot ^ 1]));
    std::memcpy(static_cast<char*>(host) + prev_off, ctx->h_xfer + (slot ^ 1) * kXferChunk, prev_n);
    return TRC_OK;
}
trc_status trc_copy_to_device(trc_ctx* ctx, void* dev, const void* host, size_t bytes, hipStream_t st) {
    if (bytes == 0) return TRC_OK;
    { const trc_status rs = xfer_ready(ctx); if (rs != TRC_OK) return rs; }
    int slot = 0;
    bool used[2] = {false, false};
    for (size_t off = 0; off < bytes; off += kXferChunk, slot ^= 1) {
        const size_t n = std::min(kXferChunk, bytes - off);
        if (used[slot]) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_xfer[slot]));       // the copy that last read this half has finished
        std::memcpy(ctx->h_xfer + slot * kXferChunk, static_cast<const char*>(host) + off, n);
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<char*>(dev) + off, ctx->h_xfer + slot * kXferChunk, n, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipEventRecord(ctx->ev_xfer[slot], st));
        used[slot] = true;
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return TRC_OK;
}

static size_t dyn_lds_bytes(const DScene& sc, bool stats) {
    size_t dwords = sc.lds_dwords + (size_t)sc.stack_lds * kBlock * (stats ? 2u : 1u);
    return dwords * 4;
}
size_t trc_dyn_lds_bytes(const trc_ctx* ctx, bool stats) { return dyn_lds_bytes(ctx->ks.sc, stats); }

// LDS plan of ONE production render launch on a tree that is read from memory.  A CU holds 4 x W one-wavefront
// workgroups (W = the waves per SIMD the kernel's registers allow) only if each fits 160 KB / (4 W) of LDS: the lane
// stacks plus the staged scene prefix.  plan_lds (upload time) assumes W = 4 and a stack as deep as the tree; here
//  * the stack keeps kStackLdsLevels entries per lane in LDS, deeper entries go to per-workgroup rows in global memory
//    (dev_intersect.hpp::stack_put) -- a ray rarely has more siblings pending, the tree depth is the worst case;
//  * the node prefix takes what is left of the workgroup's share (host trees: any prefix of the BFS order may be staged).
// Measured on the 1 M-triangle scene (depth 27: 6.9 KB of stack): the tracePath kernel (5 waves/SIMD by registers) was
// held at 4 by LDS; 34.9 -> 32.8 ms per 32-spp launch once it fits.
constexpr uint32_t kStackLdsLevels = 16;
// LDS entries of a two-level stack when `levels` are wanted, and the global rows behind them
static void set_hybrid_stack(DScene& sc, uint32_t levels) {
    sc.stack_lds = std::min(sc.stack_depth, std::max(1u, levels));
    sc.stack_ovf_rows = sc.stack_depth - sc.stack_lds;
}
static void plan_launch_lds(const trc_ctx* ctx, DScene& sc, uint32_t waves_per_simd, bool hybrid) {
    const uint32_t levels = ctx->knobs.stack_lds_levels > 0 ? (uint32_t)ctx->knobs.stack_lds_levels : kStackLdsLevels;
    if (hybrid) set_hybrid_stack(sc, ctx->knobs.no_lds_fit ? std::max(levels, sc.stack_depth) : levels);      // (knob: the whole stack in LDS)
    if (ctx->knobs.no_lds_fit) return;                                                 // A/B knobs (trc_debug_set)
    if (!ctx->lds_prefix_ok) return;                                                   // all or nothing was decided at upload
    const uint32_t per_wg = ((160u * 1024u / 4u) / (4u * waves_per_simd)) & ~127u;     // dwords; LDS is granted in 512-byte units
    const uint32_t stack = sc.stack_lds * kBlock;
    uint32_t room = std::max(per_wg > stack ? per_wg - stack : 0u, sc.off_nodes + kNodeDwords);
    room = std::min(room, kLdsSceneBytes / 4);
    sc.n_lds_nodes = std::min(sc.n_nodes, (room - sc.off_nodes) / kNodeDwords);
    sc.lds_dwords = sc.off_nodes + sc.n_lds_nodes * kNodeDwords;
}

// LDS plan of a persistent-workgroup launch (k_render_pwg): `waves` wavefronts share one staged prefix; the workgroup's
// share of the CU's 160 KB minus the wavefronts' stacks is all node prefix.  False when even one node does not fit.
static bool plan_pwg_lds(const trc_ctx* ctx, DScene& sc, uint32_t waves, uint32_t per_cu, bool hybrid, uint32_t default_levels, uint32_t park_rows) {
    uint32_t levels = ctx->knobs.stack_lds_levels > 0 ? (uint32_t)ctx->knobs.stack_lds_levels : default_levels;      // trc_render_config.hpp
    const uint32_t per_wg = ((160u * 1024u / 4u) / per_cu) & ~127u;
    for (;; --levels) {
        DScene t = sc;
        if (hybrid) set_hybrid_stack(t, levels);
        const uint32_t stacks = waves * (t.stack_lds + park_rows) * kBlock;       // per wavefront: its stack rows, then its park rows (k_render_pwg)
        if (per_wg >= stacks + t.off_nodes + kNodeDwords) {
            t.n_lds_nodes = std::min(t.n_nodes, (per_wg - stacks - t.off_nodes) / kNodeDwords);
            t.lds_dwords = t.off_nodes + t.n_lds_nodes * kNodeDwords;
            sc = t;
            return true;
        }
        // a scene with many analytic primitives / materials: fewer stack entries in LDS before giving the persistent workgroups up
        if (!hybrid || levels <= 6u || ctx->knobs.stack_lds_levels > 0) return false;
    }
}

trc_status trc_ensure_tiles(trc_ctx* ctx, uint32_t nranks, uint32_t rank, uint32_t view_height, uint32_t blk_shift) {
    if (ctx->d_tiles && ctx->d_block_cost && ctx->tiles_nranks == nranks && ctx->tiles_rank == rank &&
        ctx->tiles_view_height == view_height && ctx->tiles_blk_shift == blk_shift) return TRC_OK;
    std::vector<uint32_t> tiles = make_tiles(ctx->width, ctx->height, nranks, rank, view_height, blk_shift);
    if (tiles.size() > (size_t)kLaunchIndexMask) return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "frame too large: more pixel blocks than a launch-list entry can name");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    // the cache key (tiles_nranks ...) is written LAST: a failed allocation below leaves the list invalid, so the next
    // call rebuilds it instead of launching with a null block_cost / order buffer
    ctx->tiles_nranks = 0;
    (void)hipFree(ctx->d_tiles); ctx->d_tiles = nullptr;
    (void)hipFree(ctx->d_block_cost); ctx->d_block_cost = nullptr;
    for (int k = 0; k < 2; ++k) { (void)hipFree(ctx->d_order_keys[k]); (void)hipFree(ctx->d_order_vals[k]); ctx->d_order_keys[k] = ctx->d_order_vals[k] = nullptr; }
    (void)hipFree(ctx->d_order_hist); ctx->d_order_hist = nullptr;
    (void)hipFree(ctx->d_split); ctx->d_split = nullptr;
    (void)hipFree(ctx->d_whole); ctx->d_whole = nullptr;
    (void)hipFree(ctx->d_cost_est); ctx->d_cost_est = nullptr;
    (void)hipFree(ctx->d_qsplit); ctx->d_qsplit = nullptr;
    (void)hipFree(ctx->d_qwhole); ctx->d_qwhole = nullptr;
    (void)hipFree(ctx->d_swhole); ctx->d_swhole = nullptr;
    (void)hipFree(ctx->d_launch); ctx->d_launch = nullptr;
    (void)hipFree(ctx->d_plan_gather); ctx->d_plan_gather = nullptr;
    (void)hipFree(ctx->d_cost_scratch); ctx->d_cost_scratch = nullptr;
    ctx->plan_streak = 0;
    ctx->cost_valid = false; ctx->d_last_order = nullptr; ctx->d_stale_order = nullptr; ctx->cost_quarters = false; ctx->launch_cap = 0;
    ctx->n_tiles = (uint32_t)tiles.size();
    if (ctx->n_tiles) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_tiles, tiles.size() * 4));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_block_cost, tiles.size() * 4 * kCostSlots));      // per block: whole / quarters / sixteenths
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_split, tiles.size() * 4));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_whole, tiles.size() * 4));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cost_est, tiles.size() * 4 * kCostSlots));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_qsplit, tiles.size() * 16));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_qwhole, tiles.size() * 16));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_swhole, tiles.size() * 64));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_launch, tiles.size() * 4 * kCostSlots));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_plan_gather, tiles.size() * 4 * 22));     // k_plan_gather: part, raw, 4 quarters, 16 sixteenths per rank
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_cost_scratch, tiles.size() * 4 * kCostSlots));   // where instrumented launches leave their durations      // k_plan_gather: part, raw, 4 quarters per rank
        if (!ctx->d_plan) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_plan, kPlanWords * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_plan, 0, kPlanWords * sizeof(uint32_t), ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_split, 0, tiles.size() * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_qsplit, 0, tiles.size() * 16, ctx->stream));
        ctx->launch_cap = (uint32_t)tiles.size() * kCostSlots;
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_order_keys[k], tiles.size() * 4));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_order_vals[k], tiles.size() * 4));
        }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_order_hist, (trc_sort_hist_words(ctx->n_tiles) + 256) * 4));
        { const trc_status cs = trc_copy_to_device(ctx, ctx->d_tiles, tiles.data(), tiles.size() * 4, ctx->stream); if (cs != TRC_OK) return cs; }
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->tiles_nranks = nranks; ctx->tiles_rank = rank; ctx->tiles_view_height = view_height; ctx->tiles_blk_shift = blk_shift;
    return TRC_OK;
}

namespace {

// tables of pbrt::SobolSampler for a 2^m x 2^m pixel grid (include/trc_sobol.h), uploaded once per m
trc_status ensure_sobol_tables(trc_ctx* ctx, uint32_t m) {
    if (!ctx->d_sobol_vdc) {
        std::vector<uint32_t> m32(TRC_SOBOL_DIMS * TRC_SOBOL_MATRIX_SIZE);
        trc_sobol_matrices32(m32.data());
        if (!ctx->d_sobol32) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sobol32, m32.size() * sizeof(uint32_t)));
        { const trc_status cs = trc_copy_to_device(ctx, ctx->d_sobol32, m32.data(), m32.size() * sizeof(uint32_t), ctx->stream); if (cs != TRC_OK) return cs; }
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sobol_vdc, 2 * TRC_SOBOL_MATRIX_SIZE * sizeof(uint64_t)));
        ctx->sobol_m = ~0u;
    }
    if (ctx->sobol_m != m) {
        uint64_t tb[2 * TRC_SOBOL_MATRIX_SIZE] = {};
        if (m != 0 && trc_sobol_interval_tables(m, tb, tb + TRC_SOBOL_MATRIX_SIZE) != 0)
            return fail(ctx, TRC_ERR_UNSUPPORTED, "Sobol interval tables");
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));            // a launch in flight may still read the old tables
        HIP_TRY(ctx, hipMemcpy(ctx->d_sobol_vdc, tb, sizeof(tb), hipMemcpyHostToDevice));
        ctx->sobol_m = m;
    }
    return TRC_OK;
}

template <bool LDS>
void launch_render(trc_ctx* ctx, const KRender& kp, bool stats, uint32_t integrator, size_t lds, uint32_t n_workgroups, bool dense = false) {
    dim3 grid(n_workgroups), block(kBlock);
    if (dense) { hipLaunchKernelGGL(k_render_dense, grid, block, lds, ctx->stream, kp); return; }
    if (kp.strip > 1) {                      // few samples per pixel: a strip of blocks per wavefront (production kernels)
        dim3 sgrid((ctx->n_tiles + kp.strip - 1) / kp.strip);
        if (kp.sobol32) {
            if (integrator == TRC_INTEGRATOR_MIS) hipLaunchKernelGGL((k_render_strip<LDS, TRC_INTEGRATOR_MIS, true>), sgrid, block, lds, ctx->stream, kp);
            else hipLaunchKernelGGL((k_render_strip<LDS, TRC_INTEGRATOR_PATH, true>), sgrid, block, lds, ctx->stream, kp);
        } else if (integrator == TRC_INTEGRATOR_VOLUME) hipLaunchKernelGGL((k_render_strip<LDS, TRC_INTEGRATOR_VOLUME, false>), sgrid, block, lds, ctx->stream, kp);
        else if (integrator == TRC_INTEGRATOR_MIS) hipLaunchKernelGGL((k_render_strip<LDS, TRC_INTEGRATOR_MIS, false>), sgrid, block, lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_render_strip<LDS, TRC_INTEGRATOR_PATH, false>), sgrid, block, lds, ctx->stream, kp);
        return;
    }
    if (kp.sobol32) {                        // TRC_FLAG_SOBOL (production kernels of tracePath / traceMIS only)
        if (integrator == TRC_INTEGRATOR_MIS) hipLaunchKernelGGL((k_render<LDS, false, TRC_INTEGRATOR_MIS, true>), grid, block, lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_render<LDS, false, TRC_INTEGRATOR_PATH, true>), grid, block, lds, ctx->stream, kp);
    } else if (integrator == TRC_INTEGRATOR_VOLUME) {
        if (stats) hipLaunchKernelGGL((k_render<LDS, true, TRC_INTEGRATOR_VOLUME>), grid, block, lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_render<LDS, false, TRC_INTEGRATOR_VOLUME>), grid, block, lds, ctx->stream, kp);
    } else if (integrator == TRC_INTEGRATOR_MIS) {
        if (stats) hipLaunchKernelGGL((k_render<LDS, true, TRC_INTEGRATOR_MIS>), grid, block, lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_render<LDS, false, TRC_INTEGRATOR_MIS>), grid, block, lds, ctx->stream, kp);
    } else {
        if (stats) hipLaunchKernelGGL((k_render<LDS, true, TRC_INTEGRATOR_PATH>), grid, block, lds, ctx->stream, kp);
        else hipLaunchKernelGGL((k_render<LDS, false, TRC_INTEGRATOR_PATH>), grid, block, lds, ctx->stream, kp);
    }
}

// persistent workgroups (k_render_pwg): grid = workgroups the GPU holds at once, block = the workgroup's wavefronts
template <int INTEGRATOR, bool SOBOL>
hipError_t launch_pwg_one(trc_ctx* ctx, const KRender& kp, uint32_t grid, size_t lds) {
    // more than 64 KB of dynamic LDS has to be asked for once per kernel AND per device (the attribute is set on the
    // current device's copy of the function): remembered in the context, which is bound to one device
    const uint32_t bit = 1u << (INTEGRATOR * 2 + (SOBOL ? 1 : 0));
    if (lds > 64 * 1024 && !(ctx->pwg_lds_granted & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_pwg<INTEGRATOR, SOBOL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
        if (e != hipSuccess) return e;
        ctx->pwg_lds_granted |= bit;
    }
    hipLaunchKernelGGL((k_render_pwg<INTEGRATOR, SOBOL>), dim3(grid), dim3(64 * pwg_waves(INTEGRATOR)), lds, ctx->stream, kp);
    return hipSuccess;
}
hipError_t launch_render_pwg(trc_ctx* ctx, const KRender& kp, uint32_t integrator, uint32_t grid, size_t lds) {
    if (kp.sobol32) return integrator == TRC_INTEGRATOR_MIS ? launch_pwg_one<TRC_INTEGRATOR_MIS, true>(ctx, kp, grid, lds)
                                                            : launch_pwg_one<TRC_INTEGRATOR_PATH, true>(ctx, kp, grid, lds);
    if (integrator == TRC_INTEGRATOR_VOLUME) return launch_pwg_one<TRC_INTEGRATOR_VOLUME, false>(ctx, kp, grid, lds);
    if (integrator == TRC_INTEGRATOR_MIS) return launch_pwg_one<TRC_INTEGRATOR_MIS, false>(ctx, kp, grid, lds);
    return launch_pwg_one<TRC_INTEGRATOR_PATH, false>(ctx, kp, grid, lds);
}

}  // namespace

// ----------------------------------------------------------------------- collectives: RCCL or the caller's table
namespace {

size_t dtype_bytes(int dtype) { return dtype == kNcclUint8 ? 1 : 4; }

std::string rccl_error(const char* what, int rc) {
    return std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
}

// host-staged table call: wait for the producers on `st`, bring `bytes` at `buf` to pinned host memory, let the caller's
// function work on it, put the result back (only where the collective defines one)
template <typename Call>
trc_status staged(trc_ctx* ctx, void* buf, size_t bytes, bool copy_back, hipStream_t st, const char* what, Call&& call) {
    if (bytes > ctx->h_stage_bytes) {
        if (ctx->h_stage) { (void)hipHostFree(ctx->h_stage); ctx->h_stage = nullptr; ctx->h_stage_bytes = 0; }
        HIP_TRY(ctx, hipHostMalloc(&ctx->h_stage, bytes, hipHostMallocDefault));
        ctx->h_stage_bytes = bytes;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_stage, buf, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const int rc = call(ctx->h_stage);
    if (rc != 0) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's collective returned " + std::to_string(rc));
    if (copy_back) {
        HIP_TRY(ctx, hipMemcpyAsync(buf, ctx->h_stage, bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));          // the staging buffer is reused by the next collective
    }
    return TRC_OK;
}

}  // namespace

trc_status trc_coll_reduce(trc_ctx* ctx, void* buf, size_t count, int dtype, int op, int root, hipStream_t st, const char* what) {
    if (ctx->coll_active) {
        const trc_collectives& c = ctx->coll;
        if (!c.host_staged) {
            const int rc = c.reduce(c.user, buf, count, dtype, op, root, (void*)st);
            return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's collective returned " + std::to_string(rc));
        }
        return staged(ctx, buf, count * dtype_bytes(dtype), ctx->rank == root, st, what,
                      [&](void* h) { return c.reduce(c.user, h, count, dtype, op, root, nullptr); });
    }
    if (!ctx->comm) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + " before trc_group_init / trc_group_set_collectives");
    const int rc = g_rccl.Reduce(buf, buf, count, dtype, op, root, ctx->comm, st);      // in place on the root (sendbuff == recvbuff is allowed)
    return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, rccl_error(what, rc));
}

trc_status trc_coll_allreduce(trc_ctx* ctx, void* buf, size_t count, int dtype, int op, hipStream_t st, const char* what) {
    if (ctx->coll_active) {
        const trc_collectives& c = ctx->coll;
        if (!c.host_staged) {
            const int rc = c.allreduce(c.user, buf, count, dtype, op, (void*)st);
            return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's collective returned " + std::to_string(rc));
        }
        return staged(ctx, buf, count * dtype_bytes(dtype), true, st, what,
                      [&](void* h) { return c.allreduce(c.user, h, count, dtype, op, nullptr); });
    }
    if (!ctx->comm) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + " before trc_group_init / trc_group_set_collectives");
    const int rc = g_rccl.AllReduce(buf, buf, count, dtype, op, ctx->comm, st);
    return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, rccl_error(what, rc));
}

trc_status trc_coll_allgather(trc_ctx* ctx, void* buf, size_t bytes_per_rank, hipStream_t st, const char* what) {
    if (ctx->coll_active) {
        const trc_collectives& c = ctx->coll;
        if (!c.host_staged) {
            const int rc = c.allgather(c.user, buf, bytes_per_rank, (void*)st);
            return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's collective returned " + std::to_string(rc));
        }
        return staged(ctx, buf, bytes_per_rank * (size_t)ctx->nranks, true, st, what,
                      [&](void* h) { return c.allgather(c.user, h, bytes_per_rank, nullptr); });
    }
    if (!ctx->comm) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + " before trc_group_init / trc_group_set_collectives");
    const int rc = g_rccl.AllGather(static_cast<char*>(buf) + (size_t)ctx->rank * bytes_per_rank, buf, bytes_per_rank, kNcclUint8, ctx->comm, st);
    return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, rccl_error(what, rc));
}

// ======================================================================= C ABI
extern "C" {

uint32_t trc_abi_version(void) { return TRC_ABI_VERSION; }

const char* trc_build_flavor(void) {
#ifdef TRC_FAST_MATH
    return "fast-math";
#else
    return "exact";
#endif
}

int trc_has_test_hooks(void) {
#ifdef TRC_TEST_HOOKS
    return 1;
#else
    return 0;
#endif
}

const char* trc_status_string(trc_status s) {
    switch (s) {
        case TRC_OK: return "ok";
        case TRC_ERR_INVALID_ARG: return "invalid argument";
        case TRC_ERR_NO_DEVICE: return "no usable HIP device";
        case TRC_ERR_HIP: return "HIP runtime error";
        case TRC_ERR_NO_SCENE: return "no scene uploaded";
        case TRC_ERR_NO_FRAME: return "no frame allocated (trc_resize)";
        case TRC_ERR_BVH_INVALID: return "invalid BVH";
        case TRC_ERR_UNSUPPORTED: return "unsupported";
        case TRC_ERR_RCCL: return "RCCL error";
        case TRC_ERR_OOM: return "out of memory";
        default: return "unknown status";
    }
}

const char* trc_last_error(const trc_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

trc_status trc_create(int device, trc_ctx** out) {
    if (!out) return TRC_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return TRC_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return TRC_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TRC_ERR_NO_DEVICE;
    trc_ctx* ctx = new (std::nothrow) trc_ctx();
    if (!ctx) return TRC_ERR_OOM;
    ctx->device = device;
    { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) ctx->cu_count = cus; }
    if (ctx->cu_count <= 0) ctx->cu_count = 256;
    {
        auto env_int = [](const char* name, bool flag) {
            const char* v = std::getenv(name);
            return !v ? 0 : flag ? 1 : std::max(0, std::atoi(v));
        };
        ctx->knobs.no_lds_fit = env_int("TRC_NO_LDS_FIT", true);
        ctx->knobs.stack_lds_levels = env_int("TRC_STACK_LDS_LEVELS", false);
        ctx->knobs.strip_len = env_int("TRC_STRIP_LEN", false);
        ctx->knobs.no_pwg = env_int("TRC_NO_PWG", true);
        ctx->knobs.sppm_serial_camera = env_int("TRC_SPPM_SERIAL_CAMERA", true);
        ctx->knobs.no_cost_filter = env_int("TRC_NO_COST_FILTER", true);
        ctx->knobs.no_cold_probe = env_int("TRC_NO_COLD_PROBE", true);
        ctx->knobs.probe_spp = env_int("TRC_PROBE_SPP", false);
        ctx->knobs.no_plan_reuse = env_int("TRC_NO_PLAN_REUSE", true);
        ctx->knobs.no_coalesce = env_int("TRC_NO_COALESCE", true);
        ctx->knobs.no_dense = env_int("TRC_NO_DENSE", true);
        ctx->knobs.head_stages = env_int("TRC_HEAD_STAGES", false);
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc((void**)&ctx->d_stats, sizeof(unsigned long long) * kStatRows * kStatRowStride) != hipSuccess ||
        hipMalloc((void**)&ctx->d_stats_sum, sizeof(unsigned long long) * kStatRowStride) != hipSuccess ||
        hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * kStatRows * kStatRowStride, ctx->stream) != hipSuccess) {
        trc_destroy(ctx);
        return TRC_ERR_HIP;
    }
    *out = ctx;
    return TRC_OK;
}

void trc_destroy(trc_ctx* ctx) {
    if (!ctx) return;
    // nobody can read the frame of a kept launch after this call: it is dropped, not launched (trc_synchronize, a download or
    // trc_get_stats before trc_destroy launches it and reports its status)
    ctx->has_deferred = false;
    if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
    if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(ctx->comm);
    trc_sppm_release(ctx);
    collect_events(ctx);
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    (void)hipFree(ctx->d_blob); (void)hipFree(ctx->d_bvh_ref); (void)hipFree(ctx->d_density); (void)hipFree(ctx->d_occupancy); (void)hipFree(ctx->d_envmap); (void)hipFree(ctx->d_sobol32); (void)hipFree(ctx->d_sobol_vdc); (void)hipFree(ctx->d_rng); (void)hipFree(ctx->d_accum);
    (void)hipFree(ctx->d_tiles); (void)hipFree(ctx->d_stats); (void)hipFree(ctx->d_stats_sum); (void)hipFree(ctx->d_reduce_recv);
    (void)hipFree(ctx->d_block_cost); (void)hipFree(ctx->d_order_hist); (void)hipFree(ctx->d_split); (void)hipFree(ctx->d_whole); (void)hipFree(ctx->d_cost_est); (void)hipFree(ctx->d_qsplit); (void)hipFree(ctx->d_qwhole); (void)hipFree(ctx->d_swhole); (void)hipFree(ctx->d_launch); (void)hipFree(ctx->d_plan); (void)hipFree(ctx->d_plan_gather); (void)hipFree(ctx->d_cost_scratch);
    for (int k = 0; k < 2; ++k) { (void)hipFree(ctx->d_order_keys[k]); (void)hipFree(ctx->d_order_vals[k]); }
    (void)hipFree(ctx->d_accum_alt); (void)hipFree(ctx->d_stack_ovf); (void)hipFree(ctx->d_queue);
    (void)hipFree(ctx->d_shard_in); (void)hipFree(ctx->d_shard_out); (void)hipFree(ctx->d_shard_src);
    if (ctx->ev_snapshot_free) (void)hipEventDestroy(ctx->ev_snapshot_free);
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    if (ctx->h_xfer) (void)hipHostFree(ctx->h_xfer);
    for (hipEvent_t e : ctx->ev_xfer) if (e) (void)hipEventDestroy(e);
    if (ctx->h_readback) (void)hipHostFree(ctx->h_readback);
    for (hipEvent_t e : {ctx->ev_rendered, ctx->ev_busy, ctx->ev_busy_alt}) if (e) (void)hipEventDestroy(e);
    if (ctx->comm_stream) (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

trc_status trc_upload_scene(trc_ctx* ctx, const trc_scene* scene) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint32_t> blob;
    uint64_t blob_total = 0;
    KScene ks{};
    trc_status st = build_blob(ctx, scene, blob, blob_total, ks);
    if (st != TRC_OK) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_blob) { (void)hipFree(ctx->d_blob); ctx->d_blob = nullptr; }
    if (ctx->d_bvh_ref) { (void)hipFree(ctx->d_bvh_ref); ctx->d_bvh_ref = nullptr; }
    ctx->n_bvh_ref = 0;
    ctx->has_scene = false;
    ctx->blob_bytes = (size_t)blob_total * 4;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_blob, ctx->blob_bytes));
    { const trc_status cs = trc_copy_to_device(ctx, ctx->d_blob, blob.data(), blob.size() * 4, ctx->stream); if (cs != TRC_OK) return cs; }
    { trc_status rs = trc_repack_triangles(ctx, scene, ks.sc, ctx->d_blob, nullptr); if (rs != TRC_OK) return rs; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ks.sc.blob = ctx->d_blob;
    ctx->ks = ks;
    ctx->lds_scene = ks.sc.n_lds_nodes == ks.sc.n_nodes;      // whole tree staged in LDS
    ctx->lds_prefix_ok = true;
    ctx->has_scene = true;
    ctx->cost_valid = false; ctx->d_last_order = nullptr; ctx->d_stale_order = nullptr;      // another scene: the recorded block costs say nothing about it
    return TRC_OK;
}

trc_status trc_upload_density(trc_ctx* ctx, const trc_GridDensityInfo* info, const float* density) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_density); ctx->d_density = nullptr;
    (void)hipFree(ctx->d_occupancy); ctx->d_occupancy = nullptr;
    ctx->dinfo = trc_GridDensityInfo{};
    if (!info && !density) return TRC_OK;                                  // cleared
    if (!info || !density || info->nx == 0 || info->ny == 0 || info->nz == 0) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_upload_density: empty grid");
    const uint64_t count = (uint64_t)info->nx * info->ny * info->nz;
    if (count > (1ull << 31)) return fail(ctx, TRC_ERR_UNSUPPORTED, "trc_upload_density: more than 2^31 cells");
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_density, count * sizeof(float)));
    { const trc_status cs = trc_copy_to_device(ctx, ctx->d_density, density, count * sizeof(float), ctx->stream); if (cs != TRC_OK) return cs; }
    // occupancy of 4x4x4 bricks: brick b covers lookups whose base cell i = floor(p*n - 0.5) has (i + 1) >> 2 == b, i.e.
    // the cells 4b-1 .. 4b+3 and their +1 neighbours; nonzero = any of them holds a value other than +0
    const int nx = (int)info->nx, ny = (int)info->ny, nz = (int)info->nz;
    const int nbx = (nx + 4) >> 2, nby = (ny + 4) >> 2, nbz = (nz + 4) >> 2;
    std::vector<uint8_t> occ((size_t)nbx * nby * nbz, 0);
    for (int z = 0; z < nz; ++z)
        for (int y = 0; y < ny; ++y)
            for (int x = 0; x < nx; ++x) {
                const float v = density[((size_t)z * ny + y) * nx + x];
                uint32_t bits; std::memcpy(&bits, &v, 4);
                if (bits == 0u) continue;                                   // exactly +0: interpolates to +0
                // cell c is touched by base cells c-1 and c: bricks (c >> 2) and ((c + 1) >> 2)
                for (int bz = z >> 2; bz <= (z + 1) >> 2; ++bz)
                    for (int by = y >> 2; by <= (y + 1) >> 2; ++by)
                        for (int bx = x >> 2; bx <= (x + 1) >> 2; ++bx) occ[((size_t)bz * nby + by) * nbx + bx] = 1;
            }
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_occupancy, occ.size()));
    { const trc_status cs = trc_copy_to_device(ctx, ctx->d_occupancy, occ.data(), occ.size(), ctx->stream); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->dinfo = *info;
    return TRC_OK;
}

trc_status trc_set_camera(trc_ctx* ctx, const trc_Camera* c) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !c) return TRC_ERR_INVALID_ARG;
    const DCamera before = ctx->cam;
    DCamera& d = ctx->cam;
    const trc_float3* src[6] = {&c->lookFrom, &c->u, &c->v, &c->vertical, &c->horizontal, &c->cornerLowLeft};
    float* dst[6] = {d.lookFrom, d.u, d.v, d.vertical, d.horizontal, d.cornerLowLeft};
    for (int i = 0; i < 6; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    d.lenRadius = c->lenRadius;
    // another view: the blocks' recorded costs are another picture's (the next launch measures afresh: trc_render's head)
    // (the old view's launch order survives as a PRIOR for the head of the next launch: a camera usually moves a little)
    // Round 6 (tools/moving_camera.py, profiles/r06/moving_camera.txt): a camera that moves a LITTLE -- a drag, frame after frame -- keeps the
    // recorded costs: the next launch is ordered and planned by them as one pass, taking the last launch's RAW durations (the filter follows
    // a block that got heavier by 1.6 % per launch only).  Under a camera that moves 0.5 degrees per frame that costs 4-8 % over a settled
    // launch where forgetting the costs (head + rest on every view) costs 8-24 %.  A camera that JUMPS -- the view direction turns by more
    // than 5 degrees or the eye moves by more than 5 % of the scene's diagonal -- looks at another picture: the costs are forgotten as before.
    // Knob camera_policy: 0 = this rule, 1 = always forget, 2 = always keep (filtered costs), 3 = always keep (raw durations).
    if (!ctx->has_camera || std::memcmp(&before, &d, sizeof d) != 0) {
        auto centre = [](const DCamera& c, float out[3]) {      // direction of the view's centre ray
            float n = 0.0f;
            for (int k = 0; k < 3; ++k) { out[k] = c.cornerLowLeft[k] + 0.5f * c.horizontal[k] + 0.5f * c.vertical[k] - c.lookFrom[k]; n += out[k] * out[k]; }
            n = n > 0.0f ? 1.0f / std::sqrt(n) : 0.0f;
            for (int k = 0; k < 3; ++k) out[k] *= n;
        };
        bool small = false;
        if (ctx->has_camera && ctx->has_scene) {
            float a[3], b[3], cosang = 0.0f, shift = 0.0f, diag = 0.0f;
            centre(before, a); centre(d, b);
            for (int k = 0; k < 3; ++k) {
                cosang += a[k] * b[k];
                shift += (d.lookFrom[k] - before.lookFrom[k]) * (d.lookFrom[k] - before.lookFrom[k]);
                diag += (ctx->ks.root_box[3 + k] - ctx->ks.root_box[k]) * (ctx->ks.root_box[3 + k] - ctx->ks.root_box[k]);
            }
            small = cosang >= 0.9961947f /* cos 5 degrees */ && shift <= 0.0025f * diag;
        }
        const int policy = ctx->knobs.camera_policy;
        const bool keep = ctx->cost_valid && (policy == 0 ? small : policy >= 2);
        if (keep) { ctx->plan_streak = 0; ctx->cost_fresh_next = policy != 2; }
        else { if (ctx->cost_valid) ctx->d_stale_order = ctx->d_last_order; ctx->cost_valid = false; ctx->d_last_order = nullptr; }
    }
    ctx->has_camera = true;
    return TRC_OK;
}

trc_status trc_set_environment(trc_ctx* ctx, const float rgb[3]) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !rgb) return TRC_ERR_INVALID_ARG;
    ctx->ambient[0] = rgb[0]; ctx->ambient[1] = rgb[1]; ctx->ambient[2] = rgb[2];
    return TRC_OK;
}

trc_status trc_set_environment_map(trc_ctx* ctx, uint32_t w, uint32_t h, const float* rgb) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->d_envmap); ctx->d_envmap = nullptr; ctx->env_w = ctx->env_h = 0;
    if (!rgb) return TRC_OK;                                             // back to the constant environment
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_set_environment_map: bad size");
    const size_t bytes = (size_t)w * h * 3 * sizeof(float);
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_envmap, bytes));
    { const trc_status cs = trc_copy_to_device(ctx, ctx->d_envmap, rgb, bytes, ctx->stream); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->env_w = w; ctx->env_h = h;
    return TRC_OK;
}

trc_status trc_resize(trc_ctx* ctx, uint32_t width, uint32_t height) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || width == 0 || height == 0 || width > 65535u * 8u || height > 65535u * 8u) return TRC_ERR_INVALID_ARG;
    // pixel indices are 32-bit in the seed / tonemap / strip / SPPM kernels
    if ((uint64_t)width * height >= (1ull << 32)) return fail(ctx, TRC_ERR_UNSUPPORTED, "trc_resize: 2^32 pixels or more");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    (void)hipFree(ctx->d_rng); (void)hipFree(ctx->d_accum); (void)hipFree(ctx->d_accum_alt); (void)hipFree(ctx->d_tiles); (void)hipFree(ctx->d_reduce_recv);
    ctx->d_rng = nullptr; ctx->d_accum = nullptr; ctx->d_accum_alt = nullptr; ctx->d_composed = nullptr; ctx->d_tiles = nullptr; ctx->d_reduce_recv = nullptr;
    (void)hipFree(ctx->d_shard_in); (void)hipFree(ctx->d_shard_out); (void)hipFree(ctx->d_shard_src);
    ctx->d_shard_in = ctx->d_shard_out = ctx->d_shard_src = nullptr; ctx->shard_px = 0; ctx->shard_nranks = 0; ctx->snapshot_busy = false;
    ctx->busy = ctx->busy_alt = false;
    trc_sppm_release(ctx);          // per-pixel camera records depend on the frame size
    ctx->n_tiles = 0; ctx->tiles_nranks = 0;
    ctx->width = ctx->height = 0;
    const size_t n = (size_t)width * height;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_rng, n * 16));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_accum, n * 16));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_rng, 0, n * 16, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, n * 16, ctx->stream));
    ctx->width = width; ctx->height = height;
    return TRC_OK;
}

trc_status trc_seed(trc_ctx* ctx, uint64_t seed) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_rng) return fail(ctx, TRC_ERR_NO_FRAME, "trc_seed before trc_resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->width * ctx->height;
    trc_sppm_order_after_camera(ctx);
    hipLaunchKernelGGL(k_seed, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_rng, n, seed);
    HIP_TRY(ctx, hipGetLastError());
    return TRC_OK;
}

static trc_status copy_frame(trc_ctx* ctx, void* dev, void* host, bool to_device) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !host) return TRC_ERR_INVALID_ARG;
    if (!dev) return fail(ctx, TRC_ERR_NO_FRAME, "frame buffers not allocated (trc_resize)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)ctx->width * ctx->height * 16;
    trc_sppm_order_after_camera(ctx);
    { const trc_status cs = to_device ? trc_copy_to_device(ctx, dev, host, bytes, ctx->stream) : trc_copy_to_host(ctx, host, dev, bytes, ctx->stream); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    collect_events(ctx);
    return TRC_OK;
}
trc_status trc_upload_rng(trc_ctx* ctx, const uint32_t* rgba) { return copy_frame(ctx, ctx ? ctx->d_rng : nullptr, (void*)rgba, true); }
trc_status trc_download_rng(trc_ctx* ctx, uint32_t* rgba) { return copy_frame(ctx, ctx ? ctx->d_rng : nullptr, rgba, false); }
trc_status trc_upload_accum(trc_ctx* ctx, const float* rgba) { return copy_frame(ctx, ctx ? ctx->d_accum : nullptr, (void*)rgba, true); }
trc_status trc_download_accum(trc_ctx* ctx, float* rgba) { return copy_frame(ctx, ctx ? ctx->d_accum : nullptr, rgba, false); }

trc_status trc_clear_accum(trc_ctx* ctx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "trc_clear_accum before trc_resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, (size_t)ctx->width * ctx->height * 16, ctx->stream));
    return TRC_OK;
}

trc_status trc_tonemap(trc_ctx* ctx, uint8_t* rgba8, float* exposure_out) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !rgba8) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "trc_tonemap before trc_resize");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->width * ctx->height;
    unsigned long long* d_sums = nullptr;
    uchar4* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_sums, 3 * sizeof(unsigned long long)));
    if (hipMalloc((void**)&d_out, (size_t)n * 4) != hipSuccess) { (void)hipFree(d_sums); return fail(ctx, TRC_ERR_OOM, "hipMalloc tonemap"); }
    trc_status st = TRC_OK;
    do {
        unsigned long long sums[3];
        if (hipMemsetAsync(d_sums, 0, sizeof sums, ctx->stream) != hipSuccess) { st = fail(ctx, TRC_ERR_HIP, "tonemap memset"); break; }
        hipLaunchKernelGGL(k_tonemap_sum, dim3(std::min<uint32_t>((n + 255) / 256, 2048u)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float4*>(ctx->d_accum), n, d_sums);
        if (hipMemcpyAsync(sums, d_sums, sizeof sums, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { st = fail(ctx, TRC_ERR_HIP, "tonemap sums"); break; }
        // same binary32 / binary64 steps as oracle/oracle.cpp orc_tonemap (exp through trc_detmath.h)
        float mean[3];
        for (int c = 0; c < 3; ++c) mean[c] = (float)((double)sums[c] / 65536.0 / (double)n);
        const float luma = (mean[0] * 0.2126f + mean[1] * 0.7152f) + mean[2] * 0.0722f;
        float mapped = 1 - dm_expf(-1.0f * luma);
        mapped = std::fmin(std::fmax(mapped, 0.0f), 0.9999f);
        const float expose = 1.0f - mapped;
        if (exposure_out) *exposure_out = expose;
        hipLaunchKernelGGL(k_tonemap, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const float4*>(ctx->d_accum),
                           ctx->width, ctx->height, expose, d_out);
        if (hipGetLastError() != hipSuccess || trc_copy_to_host(ctx, rgba8, d_out, (size_t)n * 4, ctx->stream) != TRC_OK ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { st = fail(ctx, TRC_ERR_HIP, "tonemap kernel"); break; }
    } while (0);
    (void)hipFree(d_sums); (void)hipFree(d_out);
    return st;
}

// One pass of kernelPathTracing over the caller's share of the frame.  `inner`: this pass is one half of a first launch that
// trc_render split in two (below).
static trc_status render_pass(trc_ctx* ctx, const trc_params* p, bool inner);

// First launch of a block list (nothing is known about its blocks: new context, frame size, share, scene, camera or
// integrator): the launch order and the split plan come from the durations of the previous launch, and without them a launch
// runs row-major with every block whole -- config 2 +15 %, the mesh scenes +55-65 % (their heavy blocks start last and the
// launch ends on them; profiles/r04/cold_start.txt).  A pixel's samples are a chain through its RNG texel, so `spp` samples
// in one launch == h samples followed by spp - h (tested: test_spp_fusion_equals_per_frame_launches): the first launch is run
// as a HEAD of kColdHeadSpp samples, cold, and the REST ordered and planned by the head's per-block durations (costs are kept
// per sample, KRender::cost_div, so launches of different lengths speak of the same quantity).  No probe work is thrown
// away, no pixel changes; the only price is the head's own short tail.  Knob no_cold_probe switches it off.
constexpr uint32_t kPlanSettled = 8, kPlanReuse = 3;   // a settled list re-plans every fourth launch
constexpr uint32_t kColdHeadSpp = 8;           // >= 8: the head must run the same kernel and block list as the rest (k_render_strip below)
static trc_status render_check(trc_ctx* ctx, const trc_params* p) {
    if (!ctx || !p) return TRC_ERR_INVALID_ARG;
    if (!ctx->has_scene) return fail(ctx, TRC_ERR_NO_SCENE, "trc_render before trc_upload_scene");
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "trc_render before trc_resize");
    if (!ctx->has_camera) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_render before trc_set_camera");
    const uint32_t nranks = p->tile_nranks ? p->tile_nranks : 1;
    if (p->tile_rank >= nranks) return fail(ctx, TRC_ERR_INVALID_ARG, "tile_rank >= tile_nranks");
    if (p->integrator > TRC_INTEGRATOR_VOLUME) return fail(ctx, TRC_ERR_INVALID_ARG, "unknown integrator");
    if (p->integrator != TRC_INTEGRATOR_PATH && ctx->ks.sc.n_squares < 7)
        return fail(ctx, TRC_ERR_INVALID_ARG, "traceMIS / traceVolume sample squareList[5] and [6] (Render.metal:320-324,172-176)");
    if (p->flags & TRC_FLAG_SOBOL) {
        if (p->integrator == TRC_INTEGRATOR_VOLUME || (p->flags & TRC_FLAG_COLLECT_STATS))
            return fail(ctx, TRC_ERR_UNSUPPORTED, "TRC_FLAG_SOBOL: tracePath / traceMIS, production kernels only");
        if (2ull * p->max_depth > TRC_SOBOL_DIMS)
            return fail(ctx, TRC_ERR_UNSUPPORTED, "TRC_FLAG_SOBOL: 2 * max_depth exceeds the 40 generated dimensions");
        uint32_t m = 0;
        const uint32_t vh = (p->view_height != 0 && p->view_height < ctx->height) ? p->view_height : ctx->height;
        while ((1u << m) < std::max(ctx->width, vh)) ++m;
        if (m > TRC_SOBOL_MAX_LOG2RES) return fail(ctx, TRC_ERR_UNSUPPORTED, "TRC_FLAG_SOBOL: frame too large");
    }
    return TRC_OK;
}

// Launches of few samples, coalesced.  The reference dispatches ONE sample per frame (AAPLRenderer.mm:1195); such a launch
// has no second sample to regenerate finished lanes from and ends on its longest paths: 0.54 ms per sample against 0.31 in
// a fused launch.  A pixel's samples are one chain, so k calls of 1 sample == one call of k samples bit for bit (tested): a
// trc_render of fewer than kCoalesceBelow samples is therefore not launched at once but kept, and extended by the next call
// when that continues it (same parameters, frame0 following on); it is launched when kCoalesceUpTo samples have come
// together, when a call arrives that does not continue it, or when ANY other entry point of the library is entered
// (trc_flush at the top of each: downloads, tonemap, stats, seed, camera ...), so nothing observable changes.  A host that
// displays every frame (one trc_render, one trc_tonemap) gets exactly the launches it asked for; one that renders a run of
// samples before it looks gets them at the fused rate: 64 x 1 spp 34.8 -> 2x.x ms.  Knob no_coalesce switches it off.
constexpr uint32_t kCoalesceBelow = 8, kCoalesceUpTo = 16;
}  // extern "C" (trc_flush is internal: C++ linkage, declared in trc_ctx.hpp)
trc_status trc_flush(trc_ctx* ctx) {
    if (!ctx || !ctx->has_deferred) return TRC_OK;
    ctx->has_deferred = false;
    const trc_params q = ctx->deferred;
    const uint64_t calls = ctx->deferred_calls;
    const trc_status st = render_pass(ctx, &q, false);
    if (st == TRC_OK && calls > 1) ctx->launches += calls - 1;        // trc_stats.launches counts trc_render calls
    if (st != TRC_OK) {
        // the calls that were kept have already returned TRC_OK: the error of their launch surfaces in whatever entry point
        // flushes it, so it says WHICH samples did not run (trc_last_error) -- a host can re-issue exactly those
        ctx->error = "kept launch of " + std::to_string(calls) + " trc_render call(s), frames " + std::to_string(q.frame0) + " .. " +
                     std::to_string(q.frame0 + q.spp - 1) + " (" + std::to_string(q.spp) + " samples per pixel), did not run: " + ctx->error;
    }
    return st;
}
extern "C" {
trc_status trc_render(trc_ctx* ctx, const trc_params* p) {
    { const trc_status st = render_check(ctx, p); if (st != TRC_OK) return st; }
    const bool candidate = p->spp > 0 && p->spp < kCoalesceBelow && !(p->flags & TRC_FLAG_COLLECT_STATS) && !ctx->knobs.no_coalesce;
    if (ctx->has_deferred) {
        trc_params& d = ctx->deferred;
        const bool continues = candidate && p->frame0 == d.frame0 + d.spp && p->max_depth == d.max_depth && p->integrator == d.integrator &&
                               p->tile_rank == d.tile_rank && p->tile_nranks == d.tile_nranks && p->flags == d.flags && p->view_height == d.view_height;
        if (continues) {
            d.spp += p->spp;
            ctx->deferred_calls++;
            return d.spp >= kCoalesceUpTo ? trc_flush(ctx) : TRC_OK;
        }
        const trc_status st = trc_flush(ctx);
        if (st != TRC_OK) return st;
    }
    if (candidate) { ctx->deferred = *p; ctx->deferred_calls = 1; ctx->has_deferred = true; return TRC_OK; }
    return render_pass(ctx, p, false);
}

static trc_status render_pass(trc_ctx* ctx, const trc_params* p, bool inner) {
    { const trc_status st = render_check(ctx, p); if (st != TRC_OK) return st; }
    const uint32_t nranks = p->tile_nranks ? p->tile_nranks : 1;
    if (p->spp == 0) return TRC_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    collect_finished_events(ctx);        // before any launch of this call: it may consume a "not ready" sticky error

    // Launch geometry.  One 8x8 block per wavefront fills the GPU when there are many more blocks than wavefront slots
    // (32 400 blocks for 4 096 slots at 1080p).  A rank that owns 1/N of the frame (strong scaling) has about one block
    // per slot: the launch then lasts as long as its slowest wavefront, and a wavefront is as slow as the union of its
    // 64 pixels' branches.  4x4 blocks on 16 lanes give 4x the wavefronts, each with a quarter of the pixels to wait
    // for -- the same pixels, the same arithmetic per pixel (TRC_FLAG_SMALL_BLOCKS forces it, _LARGE_BLOCKS forbids it).
    uint32_t blk_shift = 3;
    const uint64_t blocks8 = (uint64_t)((ctx->width + 7) / 8) * ((ctx->height + 7) / 8) / nranks;
    const bool fits = ctx->width <= 65535u * 4u && ctx->height <= 65535u * 4u;
    if ((p->flags & TRC_FLAG_SMALL_BLOCKS) && fits) blk_shift = 2;
    if (ctx->knobs.force_blk_shift > 0) blk_shift = std::min(3u, (uint32_t)ctx->knobs.force_blk_shift - 1u);   // measurement knob: 2^k x 2^k pixel blocks
    { trc_status ts = trc_ensure_tiles(ctx, nranks, p->tile_rank, p->view_height, blk_shift); if (ts != TRC_OK) return ts; }
    if (ctx->n_tiles == 0) return TRC_OK;

    const bool stats = (p->flags & TRC_FLAG_COLLECT_STATS) != 0;
    const bool sobol = (p->flags & TRC_FLAG_SOBOL) != 0;
    if (trc_dyn_lds_bytes(ctx, stats) > 160 * 1024) return fail(ctx, TRC_ERR_UNSUPPORTED, "traversal stack exceeds the 160 KB LDS of a CU");

    KRender kp{};
    kp.ks = ctx->ks;
    if (ctx->knobs.descend_min > 0) kp.ks.sc.descend_min = (uint32_t)ctx->knobs.descend_min;      // A/B knob
    kp.cam = ctx->cam;
    kp.ambient[0] = ctx->ambient[0]; kp.ambient[1] = ctx->ambient[1]; kp.ambient[2] = ctx->ambient[2];
    kp.env_rgb = ctx->d_envmap; kp.env_w = ctx->env_w; kp.env_h = ctx->env_h;
    kp.fr.rng = ctx->d_rng; kp.fr.accum = ctx->d_accum; kp.fr.width = ctx->width; kp.fr.height = ctx->height;
    kp.spp = p->spp; kp.max_depth = p->max_depth; kp.frame0 = p->frame0;
    kp.view_height = (p->view_height != 0 && p->view_height < ctx->height) ? p->view_height : ctx->height;
    kp.tiles = ctx->d_tiles;
    kp.blk_shift = blk_shift;
    kp.stats = ctx->d_stats;
    // An instrumented launch (one wavefront per SIMD, counters in every loop) is no measurement of the production kernels'
    // blocks: its durations go to a scratch array, and it neither reads nor changes what the context knows about block costs.
    kp.block_cost = (p->flags & TRC_FLAG_COLLECT_STATS) ? ctx->d_cost_scratch : ctx->d_block_cost;
    kp.cost_div = std::max(1u, 4u * std::min(p->spp, 1u << 28));
    kp.order = nullptr;
    // launches of few samples per pixel give every wavefront a strip of consecutive blocks (k_render_strip); the unit of the
    // adaptive order is then the strip, and durations recorded for another strip length say nothing
    kp.n_tiles = ctx->n_tiles;
    kp.strip = 1;
    if (!stats) {
        // blocks per wavefront, measured at 1920x1080 (wall ms for 64 samples in launches of 1 / 4 spp) with the pooled
        // pixels of k_render_strip: strip 2: 37.6 / 28.4, 3: 36.6 / 29.7, 4: 37.5 / 31.5, >= 5: 38.8 / 35.6 -- longer strips
        // leave too few workgroups (the frame has 32 400 blocks for 4 096 wavefront slots); one block per wavefront: 80.8 / 33.0
        uint32_t want = p->spp <= 2 ? 3u : p->spp < 8 ? 2u : 1u;
        if (ctx->knobs.strip_len > 0) want = (uint32_t)ctx->knobs.strip_len;   // A/B knob: blocks per wavefront, any spp
        const uint32_t slots = (uint32_t)ctx->cu_count * 16u;
        const uint32_t room = ctx->n_tiles / (slots + slots / 2u);          // keep >= 1.5 workgroups per slot
        kp.strip = std::max(1u, std::min(want, room));
    }
    // Launch list.  (1) Order: most expensive blocks of the previous launch first (longest-processing-time order; cost = the
    // wavefront's measured duration): a block's samples are a sequential chain, so whatever starts last decides how long
    // the GPU drains.  Measured: config 2 25.2 -> 22.3 ms (ray counts as the key: 23.7), the 1 M-triangle scene 18.7 ->
    // 16.8 ms.  (2) Cost-adaptive block size (k_plan_split): the blocks that would decide the launch run as four 4x4
    // quarters.  Pixels depend on neither.
    const bool quarters_ok = kp.strip == 1 && blk_shift == 3;             // the list's blocks are 8x8: costs live in 4 slots per block
    kp.cost_stride = quarters_ok ? kCostSlots : 1u;
    if (!stats && (ctx->cost_strip != kp.strip || ctx->cost_quarters != quarters_ok)) {
        ctx->cost_valid = false; ctx->cost_strip = kp.strip; ctx->cost_quarters = quarters_ok; ctx->d_last_order = nullptr; ctx->d_stale_order = nullptr;
    }
    if (!stats && ctx->cost_integrator != p->integrator) { ctx->cost_valid = false; ctx->cost_integrator = p->integrator; ctx->d_last_order = nullptr; ctx->d_stale_order = nullptr; }
    {
        const uint32_t head = std::max(kColdHeadSpp, (uint32_t)ctx->knobs.probe_spp);
        if (!inner && !ctx->cost_valid && !stats && !ctx->knobs.no_cold_probe && !(p->flags & TRC_FLAG_FIXED_ORDER) && kp.strip == 1 &&
            p->spp >= 2u * head) {
            // Stages: the cold head, then -- where plenty of samples remain (four times the stage's) -- up to two more passes of
            // doubling length, each ordered and planned by its predecessor, then the rest.  A 64-sample launch is head + rest
            // (a third pass costs its drain: 21.8 -> 22.1 ms); a 256-sample launch is 8 + 16 + 32 + 200, which lets the split
            // plan's K ramp 16 -> 40 -> 76 INSIDE the first launch: config 3 431 -> 362 ms (and 331 at the second launch
            // instead of 368), config 4 220 -> 210, an eighth of config 3 290 -> 221 (knob head_stages = n caps the passes)
            trc_params r = *p;
            uint32_t stage = head, done = 0;
            const uint32_t max_stages = ctx->knobs.head_stages > 0 ? (uint32_t)ctx->knobs.head_stages : 3u;
            for (uint32_t k = 0; k < max_stages && p->spp - done >= (k == 0 ? 2u : 4u) * stage; ++k, stage *= 2u) {
                trc_params h = *p;
                h.spp = stage; h.frame0 = p->frame0 + done;
                trc_status st = render_pass(ctx, &h, true);
                if (st != TRC_OK) return st;
                ctx->launches--;                   // one trc_render call = one launch in trc_stats
                if (k == 0) ctx->cost_head_age = 1;
                done += stage;
            }
            r.spp = p->spp - done; r.frame0 = p->frame0 + done;
            return render_pass(ctx, &r, true);
        }
    }
    const bool may_split = quarters_ok && !stats && p->spp >= 8 && !ctx->knobs.no_split &&
                           !(p->flags & (TRC_FLAG_LARGE_BLOCKS | TRC_FLAG_FIXED_ORDER));
    // wavefront slots of the kernel this launch runs (the plan's model; the launch bounds of k_render / k_render_pwg)
    // a whole frame's worth of blocks per wavefront slot: the LDS-resident tracePath kernel at one more wavefront per SIMD
    const bool dense = ctx->lds_scene && p->integrator == TRC_INTEGRATOR_PATH && !stats && !sobol && kp.strip == 1 && !ctx->knobs.no_dense &&
                       ctx->n_tiles >= (uint32_t)TRC_DENSE_MIN_BLOCKS_PER_SLOT * (uint32_t)ctx->cu_count * 4u * TRC_PATH_WAVES_DENSE &&
                       ((dyn_lds_bytes(kp.ks.sc, false) + (size_t)TRC_PARK_DENSE * kBlock * 4u + 511u) & ~(size_t)511u) * 4u * TRC_PATH_WAVES_DENSE <= 160u * 1024u;
    const uint32_t waves_per_simd = dense ? TRC_PATH_WAVES_DENSE : p->integrator == TRC_INTEGRATOR_PATH ? (ctx->lds_scene ? TRC_PATH_WAVES : TRC_PATH_WAVES_GLOBAL)
                                  : p->integrator == TRC_INTEGRATOR_MIS ? (ctx->lds_scene ? TRC_MIS_WAVES_LDS : TRC_MIS_WAVES) : TRC_VOLUME_WAVES;
    const uint32_t wave_slots = (uint32_t)ctx->cu_count * 4u * waves_per_simd;
    if (!stats) { ctx->last_cost_div = kp.cost_div; ctx->last_wave_slots = wave_slots; }
    if (stats) {} else if (!ctx->cost_valid || (p->flags & TRC_FLAG_FIXED_ORDER) || kp.strip > 1) ctx->plan_streak = 0;      // nothing settled to reuse
    uint32_t grid_cap = ctx->n_tiles;                                     // workgroups of a one-block-per-workgroup launch
    bool planned = false;
    if (stats) {
        // row-major, every block whole, nothing recorded
    } else if (ctx->cost_valid && !(p->flags & TRC_FLAG_FIXED_ORDER) && kp.strip > 1 && ctx->d_last_order && ctx->order_age < 4) {
        kp.order = ctx->d_last_order;              // short launches: the order of a few launches ago is as good, and 13 tiny
        ctx->order_age++;                          // sort launches per 0.7 ms render are not
    } else if (ctx->cost_valid && !(p->flags & TRC_FLAG_FIXED_ORDER)) {
        const uint32_t n = (ctx->n_tiles + kp.strip - 1) / kp.strip;
        // the costs the order and the plan work on: the shortest durations seen lately (filter_block_costs), or the last launch's
        const bool filtered = !ctx->knobs.no_cost_filter;
        uint32_t* costs = filtered ? ctx->d_cost_est : ctx->d_block_cost;
        // A list whose plan has settled (kPlanSettled planned launches in a row) keeps its order and plan for kPlanReuse
        // launches: the filtered costs of a progressive render barely move from one launch to the next, and the dozen small
        // kernels below are 0.1 ms in front of every launch (schedule_ms in trc_stats) -- 2 % of an eighth of a frame.  The
        // launch that is reused ran with the same list, so the durations it leaves land in the same slots.
        const bool reuse = ctx->plan_streak >= kPlanSettled && ctx->plan_reused < kPlanReuse && ctx->plan_n == n && ctx->plan_split_mode == may_split &&
                           ctx->plan_wave_slots == wave_slots && !ctx->knobs.no_plan_reuse;
        if (reuse) {
            ctx->plan_reused++;
            if (may_split) { kp.order = ctx->d_launch; kp.n_launch = ctx->d_plan + 1; grid_cap = ctx->plan_grid_cap; planned = true; }
            else kp.order = ctx->d_last_order;
        } else {
        hipEvent_t s0 = get_event(ctx), s1 = get_event(ctx);
        if (s0) (void)hipEventRecord(s0, ctx->stream);
        hipLaunchKernelGGL(k_order_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_block_cost, costs, ctx->d_split, ctx->d_whole, ctx->d_qsplit,
                           kp.cost_stride, n, ctx->d_order_keys[0], ctx->d_order_vals[0], filtered, ctx->cost_head_age == 2 || ctx->cost_fresh_next);
        ctx->cost_fresh_next = false;
        int res = 0;
        trc_sort_pairs24(ctx->stream, ctx->d_order_keys, ctx->d_order_vals, ctx->d_order_hist, ctx->d_order_hist + trc_sort_hist_words(n), n, &res);
        kp.order = ctx->d_order_vals[res];
        ctx->d_last_order = kp.order;
        ctx->order_age = 0;
        if (may_split) {
            // the grid is sized before the plan is known: half the slots' worth of split blocks is more than any plan has
            // taken (a launch with fewer blocks than slots is capped to the slots anyway), + an eighth for sixteenths
            const uint32_t k_max = std::min(n, wave_slots / 2u);
            const uint32_t max_entries = std::min(ctx->launch_cap, std::max(n + 3u * k_max, wave_slots) + wave_slots / 8u);
            uint32_t* g_part = ctx->d_plan_gather;
            float* g_raw = reinterpret_cast<float*>(ctx->d_plan_gather + n);
            uint32_t* g_quart = ctx->d_plan_gather + 2 * (size_t)n;
            uint32_t* g_sixt = ctx->d_plan_gather + 6 * (size_t)n;
            hipLaunchKernelGGL(k_plan_gather, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_order_vals[res], ctx->d_split, costs, ctx->d_qsplit,
                               ctx->d_block_cost, n, g_part, g_raw, g_quart, g_sixt);
            hipLaunchKernelGGL(k_plan_split, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_order_keys[res], g_part, g_raw, g_quart, g_sixt, n, k_max, wave_slots,
                               max_entries, ctx->d_plan, ctx->d_launch);
            hipLaunchKernelGGL(k_build_launch, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_order_keys[res], ctx->d_order_vals[res], costs, n,
                               ctx->d_plan, ctx->d_launch, ctx->d_split, ctx->d_whole, ctx->d_qsplit, ctx->d_qwhole, ctx->d_swhole, filtered);
            kp.order = ctx->d_launch;
            kp.n_launch = ctx->d_plan + 1;
            grid_cap = max_entries;
            planned = true;
        }
        if (s0 && s1 && hipEventRecord(s1, ctx->stream) == hipSuccess) ctx->pending_sched.emplace_back(s0, s1);
        else { if (s0) ctx->event_pool.push_back(s0); if (s1) ctx->event_pool.push_back(s1); }
        ctx->plan_streak = (ctx->plan_n == n && ctx->plan_split_mode == may_split && ctx->plan_wave_slots == wave_slots) ? ctx->plan_streak + 1 : 1;
        ctx->plan_reused = 0; ctx->plan_n = n; ctx->plan_split_mode = may_split; ctx->plan_wave_slots = wave_slots; ctx->plan_grid_cap = grid_cap;
        }
    } else if (may_split && !ctx->cost_valid && kAutoSmallBlocks && fits && blocks8 <= (uint64_t)ctx->cu_count * 16u &&
               p->integrator == TRC_INTEGRATOR_PATH && ctx->lds_scene) {
        // nothing is known about the blocks yet and there are no more of them than wavefront slots (a small frame, or an
        // eighth of a 1080p frame): every block as quarters -- measured on whole small frames at 64 spp (920 / 2 040 / 3 600
        // blocks: 8.5 / 8.3 / 8.9 -> 7.0 / 7.2 / 7.4 ms); from the second launch on the plan decides block by block
        hipLaunchKernelGGL(k_build_launch_all_quarters, dim3((ctx->n_tiles + 255) / 256), dim3(256), 0, ctx->stream, ctx->n_tiles, ctx->d_plan, ctx->d_launch, ctx->d_split, ctx->d_whole, ctx->d_qsplit);
        kp.order = ctx->d_launch;
        kp.n_launch = ctx->d_plan + 1;
        grid_cap = 4u * ctx->n_tiles;
        planned = true;
    }
    if (!stats && !ctx->cost_valid && !planned && !kp.order && ctx->d_stale_order && kp.strip == 1 && !(p->flags & TRC_FLAG_FIXED_ORDER))
        kp.order = ctx->d_stale_order;              // a cold pass after a camera move: the previous view's order beats row-major
    if (!stats) ctx->d_stale_order = nullptr;       // (the buffer belongs to the next sort)
    if (!stats && !planned && ctx->split_live) {                          // this launch runs every block whole
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_split, 0, (size_t)ctx->n_tiles * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_qsplit, 0, (size_t)ctx->n_tiles * 16, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_cost_est, 0, (size_t)ctx->n_tiles * 4 * kCostSlots, ctx->stream));   // slot 0 held first quarters
    }
    if (!stats) ctx->split_live = planned;
    if (!stats && !ctx->cost_valid) HIP_TRY(ctx, hipMemsetAsync(ctx->d_cost_est, 0, (size_t)ctx->n_tiles * 4 * kCostSlots, ctx->stream));
    if (!stats) {
        ctx->cost_head_age = (ctx->cost_valid && ctx->cost_head_age == 1) ? 2 : 0;     // head -> the launch on its costs -> settled
        ctx->cost_valid = true;
    }
    kp.density = ctx->d_density;
    kp.dinfo = ctx->dinfo;
    kp.occupancy = ctx->d_occupancy;
    if (sobol) {
        // resolution = RoundUpPow2(max(wh.x, wh.y)) of the view, log2Resolution = Log2Int(resolution) (SobolSampler.hh:56-58)
        const uint32_t longest = ctx->width > kp.view_height ? ctx->width : kp.view_height;
        uint32_t m = 0;
        while ((1u << m) < longest) ++m;
        if (m > TRC_SOBOL_MAX_LOG2RES) return fail(ctx, TRC_ERR_UNSUPPORTED, "TRC_FLAG_SOBOL: frame too large");
        trc_status ts = ensure_sobol_tables(ctx, m);
        if (ts != TRC_OK) return ts;
        kp.sobol32 = ctx->d_sobol32;
        kp.sobol_vdc = ctx->d_sobol_vdc;
        kp.sobol_m = m;
    }

    bool pwg = false;
    uint32_t pwg_waves_n = 0, pwg_grid = 0, park_rows = dense ? (uint32_t)TRC_PARK_DENSE : 0u;      // LDS rows of parked per-pixel state (render_block)
    if (!stats && !ctx->lds_scene) {
        const bool strip = kp.strip > 1;
        const bool is_path = p->integrator == TRC_INTEGRATOR_PATH;
        const bool hybrid = hybrid_stack((int)p->integrator);
        if (!ctx->knobs.no_pwg && !strip && ctx->lds_prefix_ok) {                       // no_pwg: A/B knob
            pwg_waves_n = (uint32_t)pwg_waves((int)p->integrator);
            const uint32_t per_cu = (uint32_t)pwg_per_cu((int)p->integrator);
            park_rows = pwg_park_rows((int)p->integrator);
            pwg = plan_pwg_lds(ctx, kp.ks.sc, pwg_waves_n, per_cu, hybrid, pwg_stack_lds_levels((int)p->integrator), park_rows);
            if (!pwg) park_rows = 0u;
            pwg_grid = std::min((uint32_t)ctx->cu_count * per_cu, (grid_cap + pwg_waves_n - 1) / pwg_waves_n);   // small frames: no idle workgroups
        }
        if (!pwg) {
            const uint32_t waves = is_path ? (strip ? TRC_STRIP_PATH_WAVES : TRC_PATH_WAVES_GLOBAL)
                                 : p->integrator == TRC_INTEGRATOR_MIS ? (strip ? 4 : TRC_MIS_WAVES) : (strip ? 3 : TRC_VOLUME_WAVES);
            plan_launch_lds(ctx, kp.ks.sc, waves, hybrid);
        }
        const size_t rows = kp.ks.sc.stack_ovf_rows;
        const size_t need = rows * kBlock * sizeof(uint32_t) * (pwg ? (size_t)pwg_grid * pwg_waves_n : (size_t)grid_cap);   // rows per wavefront
        if (need > ctx->stack_ovf_bytes) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_stack_ovf) { (void)hipFree(ctx->d_stack_ovf); ctx->d_stack_ovf = nullptr; }
            ctx->stack_ovf_bytes = 0;
            if (hipMalloc((void**)&ctx->d_stack_ovf, need) != hipSuccess) return fail(ctx, TRC_ERR_OOM, "hipMalloc traversal-stack overflow rows");
            ctx->stack_ovf_bytes = need;
        }
        kp.stack_ovf = ctx->d_stack_ovf;
    }
    const size_t lds = pwg ? ((size_t)kp.ks.sc.lds_dwords + (size_t)pwg_waves_n * (kp.ks.sc.stack_lds + park_rows) * kBlock) * 4
                           : dyn_lds_bytes(kp.ks.sc, stats) + (dense ? (size_t)park_rows * kBlock * 4 : 0u);
    if (pwg) {
        if (!ctx->d_queue && hipMalloc((void**)&ctx->d_queue, sizeof(uint32_t)) != hipSuccess) return fail(ctx, TRC_ERR_OOM, "hipMalloc block queue");
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_queue, 0, sizeof(uint32_t), ctx->stream));
        kp.queue = ctx->d_queue;
    }

    HIP_TRY(ctx, hipGetLastError());     // the order / sort / memset launches above
    hipEvent_t e0 = get_event(ctx), e1 = get_event(ctx);
    auto give_back = [&]() { if (e0) ctx->event_pool.push_back(e0); if (e1) ctx->event_pool.push_back(e1); };
    if (!e0 || !e1) { give_back(); return fail(ctx, TRC_ERR_HIP, "hipEventCreate failed"); }
    hipError_t le = hipEventRecord(e0, ctx->stream);
    if (le == hipSuccess) {
        if (pwg) le = launch_render_pwg(ctx, kp, p->integrator, pwg_grid, lds);
        else if (ctx->lds_scene) launch_render<true>(ctx, kp, stats, p->integrator, lds, grid_cap, dense);
        else launch_render<false>(ctx, kp, stats, p->integrator, lds, grid_cap);
        if (le == hipSuccess) le = hipGetLastError();
    }
    if (le == hipSuccess) le = hipEventRecord(e1, ctx->stream);
    if (le != hipSuccess) { give_back(); return fail(ctx, TRC_ERR_HIP, std::string("k_render launch: ") + hipGetErrorString(le)); }
    ctx->pending.emplace_back(e0, e1);
    ctx->launches++;
    return TRC_OK;
}

trc_status trc_synchronize(trc_ctx* ctx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    collect_events(ctx);
    return TRC_OK;
}

trc_status trc_trace_rays(trc_ctx* ctx, const trc_ray* rays, size_t n, trc_hit* out, int any_hit) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || (n && (!rays || !out))) return TRC_ERR_INVALID_ARG;
    if (!ctx->has_scene) return fail(ctx, TRC_ERR_NO_SCENE, "trc_trace_rays before trc_upload_scene");
    if (n == 0) return TRC_OK;
    if (n > 0x7FFFFFFFu) return fail(ctx, TRC_ERR_INVALID_ARG, "too many rays in one call");
    if (any_hit & ~(TRC_TRACE_ANY_HIT | TRC_TRACE_PRODUCTION)) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_trace_rays: unknown mode bits");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    trc_ray* d_rays = nullptr; trc_hit* d_hits = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d_rays, n * sizeof(trc_ray)));
    if (hipMalloc((void**)&d_hits, n * sizeof(trc_hit)) != hipSuccess) { (void)hipFree(d_rays); return fail(ctx, TRC_ERR_OOM, "hipMalloc hits"); }
    trc_status st = TRC_OK;
    do {
        if (trc_copy_to_device(ctx, d_rays, rays, n * sizeof(trc_ray), ctx->stream) != TRC_OK) { st = fail(ctx, TRC_ERR_HIP, "H2D rays"); break; }
        KTrace kp{};
        kp.ks = ctx->ks; kp.rays = d_rays; kp.hits = d_hits; kp.n = (uint32_t)n;
        const size_t lds = trc_dyn_lds_bytes(ctx, true);
        dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
        const bool any = (any_hit & TRC_TRACE_ANY_HIT) != 0, prod = (any_hit & TRC_TRACE_PRODUCTION) != 0;
#define TRC_LAUNCH_TRACE(L, A, S) hipLaunchKernelGGL((k_trace<L, A, S>), grid, block, lds, ctx->stream, kp)
        if (ctx->lds_scene) {
            if (prod) { if (any) TRC_LAUNCH_TRACE(true, true, false); else TRC_LAUNCH_TRACE(true, false, false); }
            else      { if (any) TRC_LAUNCH_TRACE(true, true, true);  else TRC_LAUNCH_TRACE(true, false, true); }
        } else {
            if (prod) { if (any) TRC_LAUNCH_TRACE(false, true, false); else TRC_LAUNCH_TRACE(false, false, false); }
            else      { if (any) TRC_LAUNCH_TRACE(false, true, true);  else TRC_LAUNCH_TRACE(false, false, true); }
        }
#undef TRC_LAUNCH_TRACE
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { st = fail(ctx, TRC_ERR_HIP, std::string("k_trace launch: ") + hipGetErrorString(e)); break; }
        if (trc_copy_to_host(ctx, out, d_hits, n * sizeof(trc_hit), ctx->stream) != TRC_OK) { st = fail(ctx, TRC_ERR_HIP, "D2H hits"); break; }
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { st = fail(ctx, TRC_ERR_HIP, std::string("k_trace: ") + hipGetErrorString(e)); break; }
    } while (0);
    (void)hipFree(d_rays); (void)hipFree(d_hits);
    return st;
}

trc_status trc_get_stats(trc_ctx* ctx, trc_stats* out) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !out) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long h[kStatCount];
    hipLaunchKernelGGL(k_stats_sum, dim3(1), dim3(64), 0, ctx->stream, ctx->d_stats, ctx->d_stats_sum);
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats_sum, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    collect_events(ctx);
    std::memset(out, 0, sizeof *out);
    out->paths = h[kStatPaths]; out->rays = h[kStatRays]; out->shaded = h[kStatShaded];
    out->n_descend = h[kStatDescend]; out->n_return = h[kStatReturn];
    out->n_leaf_sphere = h[kStatLeafSphere]; out->n_leaf_square = h[kStatLeafSquare];
    out->n_leaf_cube = h[kStatLeafCube]; out->n_leaf_triangle = h[kStatLeafTriangle];
    out->n_hit_triangle = h[kStatHitTriangle]; out->n_hit_cube = h[kStatHitCube];
    out->launches = ctx->launches;
    out->kernel_ms = ctx->kernel_ms;
    out->schedule_ms = ctx->schedule_ms;
    return TRC_OK;
}

#ifdef TRC_TEST_HOOKS
// developer diagnostic: (lanes, wavefronts) that executed each ProfSite of the instrumented kernels
trc_status trc_debug_profile(trc_ctx* ctx, uint64_t* out, uint32_t n_sites) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !out) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long h[kStatCount + 3 * kProfCount];
    hipLaunchKernelGGL(k_stats_sum, dim3(1), dim3(64), 0, ctx->stream, ctx->d_stats, ctx->d_stats_sum);
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats_sum, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (uint32_t i = 0; i < n_sites && i < (uint32_t)kProfCount; ++i)
        for (int k = 0; k < 3; ++k) out[3 * i + k] = h[kStatCount + 3 * i + k];
    return TRC_OK;
}
#endif  // TRC_TEST_HOOKS

// developer diagnostic: the chain bound and the work bound of the last launch (tracer_abi.h)
trc_status trc_debug_launch_shape(trc_ctx* ctx, trc_launch_shape* out) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !out) return TRC_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, ctx->device) != hipSuccess || khz <= 0) khz = 2400000;
    out->clock_mhz = khz / 1000.0;
    out->wave_slots = ctx->last_wave_slots;
    const uint32_t n = ctx->cost_strip > 1 ? (ctx->n_tiles + ctx->cost_strip - 1) / ctx->cost_strip : ctx->n_tiles;
    if (n == 0 || !ctx->d_block_cost || !ctx->last_cost_div) return TRC_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const uint32_t stride = ctx->cost_quarters ? kCostSlots : 1u;
    std::vector<uint32_t> c((size_t)n * stride), sp(n, 0u), qs((size_t)n * 4u, 0u);
    { const trc_status cs = trc_copy_to_host(ctx, c.data(), ctx->d_block_cost, c.size() * 4, ctx->stream); if (cs != TRC_OK) return cs; }
    if (stride != 1u && ctx->split_live) {
        { const trc_status cs = trc_copy_to_host(ctx, sp.data(), ctx->d_split, (size_t)n * 4, ctx->stream); if (cs != TRC_OK) return cs; }
        { const trc_status cs = trc_copy_to_host(ctx, qs.data(), ctx->d_qsplit, (size_t)n * 16, ctx->stream); if (cs != TRC_OK) return cs; }
    }
    uint64_t sum = 0, longest = 0;
    uint32_t entries = 0;
    auto item = [&](uint32_t v) { sum += v; longest = std::max<uint64_t>(longest, v); entries++; };
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t* q = &c[(size_t)i * stride];
        if (!sp[i]) { item(q[0]); continue; }
        for_each_part(qs.data(), i, [&](uint32_t slot) { item(q[slot]); });
    }
    const double to_ms = (double)ctx->last_cost_div / ((double)khz);      // cost units -> shader clocks -> ms
    out->entries = entries;
    out->longest_entry_ms = (double)longest * to_ms;
    out->sum_entries_ms = (double)sum * to_ms;
    out->work_over_slots_ms = out->wave_slots ? out->sum_entries_ms / out->wave_slots : 0.0;
    return TRC_OK;
}

// developer diagnostic: the pixel blocks of the last trc_render (x | y << 16 in units of the block edge) and the duration
// each one's wavefront measured per sample (shader clocks / (4 spp), the adaptive order's sort key)
trc_status trc_debug_block_costs(trc_ctx* ctx, uint32_t* tiles, uint32_t* costs, uint32_t capacity, uint32_t* n_blocks, uint32_t* blk_shift) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (n_blocks) *n_blocks = ctx->n_tiles;
    if (blk_shift) *blk_shift = ctx->tiles_blk_shift;
    const uint32_t n = std::min(capacity, ctx->n_tiles);
    if (n == 0 || !ctx->d_tiles) return TRC_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (tiles) { const trc_status cs = trc_copy_to_host(ctx, tiles, ctx->d_tiles, (size_t)n * 4, ctx->stream); if (cs != TRC_OK) return cs; }
    if (costs) {
        const uint32_t stride = ctx->cost_quarters ? kCostSlots : 1u;
        std::vector<uint32_t> c((size_t)n * stride), sp(n, 0u), qs((size_t)n * 4u, 0u);
        { const trc_status cs = trc_copy_to_host(ctx, c.data(), ctx->d_block_cost, c.size() * 4, ctx->stream); if (cs != TRC_OK) return cs; }
        if (stride != 1u) {
            { const trc_status cs = trc_copy_to_host(ctx, sp.data(), ctx->d_split, (size_t)n * 4, ctx->stream); if (cs != TRC_OK) return cs; }
            { const trc_status cs = trc_copy_to_host(ctx, qs.data(), ctx->d_qsplit, (size_t)n * 16, ctx->stream); if (cs != TRC_OK) return cs; }
        }
        for (uint32_t i = 0; i < n; ++i) {        // a block that ran in parts: its slowest part, bit 31 set (bit 30: some of them 2x2)
            const uint32_t* q = &c[(size_t)i * stride];
            if (!sp[i]) { costs[i] = q[0]; continue; }
            uint32_t m = 0u, deep = 0u;      // bit 30: some of its quarters ran as 2x2 sixteenths; bit 29: some of those as single pixels
            for_each_part(qs.data(), i, [&](uint32_t slot) { m = std::max(m, q[slot]); if (slot >= 4u) deep |= 0x40000000u; if (slot >= 20u) deep |= 0x20000000u; });
            costs[i] = std::min(m, 0xFFFFFFu) | 0x80000000u | deep;
        }
    }
    return TRC_OK;
}

#ifdef TRC_TEST_HOOKS
trc_status trc_div_by_test(trc_ctx* ctx, const float* a, const float* b, size_t n, float* fast, float* plain) {
    if (!ctx || (n && (!a || !b || !fast || !plain))) return TRC_ERR_INVALID_ARG;
    if (n == 0) return TRC_OK;
    if (n > 0x7FFFFFFFu / 3u) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_div_by_test: too many pairs in one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, n * 8 * sizeof(float)));
    float *d_a = d, *d_b = d + n, *d_fast = d + 2 * n, *d_plain = d + 5 * n;
    trc_status ts = trc_copy_to_device(ctx, d_a, a, n * 4, ctx->stream);
    if (ts == TRC_OK) ts = trc_copy_to_device(ctx, d_b, b, n * 4, ctx->stream);
    if (ts == TRC_OK) {
        hipLaunchKernelGGL(k_div_by_test, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_a, d_b, (uint32_t)n, d_fast, d_plain);
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) ts = fail(ctx, TRC_ERR_HIP, std::string("trc_div_by_test: ") + hipGetErrorString(e));
    }
    if (ts == TRC_OK) ts = trc_copy_to_host(ctx, fast, d_fast, n * 12, ctx->stream);
    if (ts == TRC_OK) ts = trc_copy_to_host(ctx, plain, d_plain, n * 12, ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return ts;
}

trc_status trc_unary_test(trc_ctx* ctx, uint32_t op, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch, uint32_t* first_mismatch) {
    if (!ctx || !n_mismatch || op > 6u || count > (1ull << 32)) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long* d = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&d, 16));
    const unsigned long long init[2] = {0ull, ~0ull};
    unsigned long long h[2] = {0ull, ~0ull};
    hipError_t e = hipMemcpyAsync(d, init, 16, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && count) {
        hipLaunchKernelGGL(k_unary_test, dim3(ctx->cu_count * 16), dim3(256), 0, ctx->stream, op, first_bits, count, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream); else (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(ctx, TRC_ERR_HIP, std::string("trc_unary_test: ") + hipGetErrorString(e));
    *n_mismatch = h[0];
    if (first_mismatch) *first_mismatch = (uint32_t)h[1];
    return TRC_OK;
}
#endif  // TRC_TEST_HOOKS

trc_status trc_reset_stats(trc_ctx* ctx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    collect_events(ctx);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(unsigned long long) * kStatRows * kStatRowStride, ctx->stream));
    ctx->launches = 0;
    ctx->kernel_ms = 0.0;
    ctx->schedule_ms = 0.0;
    return TRC_OK;
}

trc_status trc_device_info(trc_ctx* ctx, char* name, size_t name_len, int* cu_count, size_t* hbm_bytes) {
    if (!ctx) return TRC_ERR_INVALID_ARG;
    hipDeviceProp_t prop;
    HIP_TRY(ctx, hipGetDeviceProperties(&prop, ctx->device));
    if (name && name_len) { std::snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName); }
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
    return TRC_OK;
}

trc_status trc_device_pci_bus_id(trc_ctx* ctx, char* out, size_t out_len) {
    if (!ctx || !out || out_len < 16) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipDeviceGetPCIBusId(out, (int)out_len, ctx->device));
    return TRC_OK;
}

// ----------------------------------------------------------------------- multi-GPU (RCCL over xGMI)
trc_status trc_group_unique_id(uint8_t id[TRC_UNIQUE_ID_BYTES]) {
    if (!id) return TRC_ERR_INVALID_ARG;
    std::string err;
    if (!trc_load_rccl(err)) return TRC_ERR_RCCL;
    IdBlob blob;
    std::memset(&blob, 0, sizeof blob);
    if (g_rccl.GetUniqueId(&blob) != 0) return TRC_ERR_RCCL;
    std::memcpy(id, &blob, TRC_UNIQUE_ID_BYTES);
    return TRC_OK;
}

trc_status trc_group_init(trc_ctx* ctx, const uint8_t id[TRC_UNIQUE_ID_BYTES], int nranks, int rank) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return TRC_ERR_INVALID_ARG;
    std::string err;
    if (!trc_load_rccl(err)) return fail(ctx, TRC_ERR_RCCL, err);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm) { g_rccl.CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ctx->coll_active = false;
    IdBlob blob;
    std::memcpy(&blob, id, TRC_UNIQUE_ID_BYTES);
    int rc = g_rccl.CommInitRank(&ctx->comm, nranks, blob, rank);
    if (rc != 0) {
        ctx->comm = nullptr;
        return fail(ctx, TRC_ERR_RCCL, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"));
    }
    ctx->nranks = nranks; ctx->rank = rank;
    return TRC_OK;
}

trc_status trc_group_reduce_accum(trc_ctx* ctx, int root) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->grouped()) return fail(ctx, TRC_ERR_RCCL, "trc_group_reduce_accum before trc_group_init / trc_group_set_collectives");
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "no frame");
    if (root < 0 || root >= ctx->nranks) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t count = (size_t)ctx->width * ctx->height * 4;
    return trc_coll_reduce(ctx, ctx->d_accum, count, kNcclFloat, kNcclSum, root, ctx->stream, "reduce(sum) of the accumulator");
}

// Sample sharding (the alternative to tile sharding; the definition is in tracer_abi.h): the composed pixel is
// the rank-ORDERED sum of the ranks' accumulator texels over the number of sample groups.  Rank r owns the r-th of nranks
// equal pixel slices: all-to-all of the slices, k_fold_shards, gather (root) or all-gather (every rank) of the results.
__global__ void __launch_bounds__(256) k_fold_shards(const float4* __restrict__ in, float4* __restrict__ out, uint32_t n_px,
                                                     uint32_t slice_px, uint32_t nranks, float groups) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_px) return;
    float4 a = in[i];                                        // rank 0's texel starts the sum (not 0 + it: -0 stays -0)
    for (uint32_t p = 1; p < nranks; ++p) {
        const float4 b = in[(size_t)p * slice_px + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    a.x /= groups; a.y /= groups; a.z /= groups; a.w /= groups;
    out[i] = a;
}

extern "C++" {
namespace {

// pixels of slice p when n_px pixels are cut into nranks slices of slice_px (the last ones may be short or empty)
inline size_t slice_count(size_t n_px, size_t slice_px, int p) {
    const size_t lo = std::min(n_px, (size_t)p * slice_px), hi = std::min(n_px, (size_t)(p + 1) * slice_px);
    return hi - lo;
}

// root >= 0: the composed frame lands in ctx->d_shard_out on the root; root < 0: on every rank.  `src` is left untouched.
trc_status compose_samples(trc_ctx* ctx, const float* src, int root, uint32_t groups, hipStream_t st, const char* what) {
    const int N = ctx->nranks, me = ctx->rank;
    const size_t n_px = (size_t)ctx->width * ctx->height;
    const size_t slice_px = (n_px + (size_t)N - 1) / (size_t)N;
    const size_t slice_f = slice_px * 4, slice_bytes = slice_px * 16, total_bytes = slice_bytes * (size_t)N;
    if (!ctx->d_shard_in || ctx->shard_px != slice_px || ctx->shard_nranks != N) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
        if (ctx->d_composed == ctx->d_shard_out) ctx->d_composed = nullptr;     // never leave it pointing at freed memory (a failed hipMalloc below returns)
        (void)hipFree(ctx->d_shard_in); (void)hipFree(ctx->d_shard_out); ctx->d_shard_in = ctx->d_shard_out = nullptr;
        ctx->shard_px = 0; ctx->shard_nranks = 0;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_shard_in, total_bytes));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_shard_out, total_bytes));
        // zero-filled in stream order with the first use (hipMemset runs on the NULL stream, which the context's
        // non-blocking streams do not wait for: it could land on top of the slices copied in below)
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_shard_in, 0, total_bytes, st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_shard_out, 0, total_bytes, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        ctx->shard_px = slice_px; ctx->shard_nranks = N;
    }
    const size_t mine = slice_count(n_px, slice_px, me);
    // 1. slice `me` of every rank's accumulator -> d_shard_in[p]
    if (ctx->coll_active) {
        const trc_collectives& c = ctx->coll;
        if (!c.alltoall || (root >= 0 ? !c.gather : !c.allgather))
            return trc_fail(ctx, TRC_ERR_UNSUPPORTED, std::string(what) + ": the collectives table has no alltoall / gather");
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_shard_in, src, n_px * 16, hipMemcpyDeviceToDevice, st));
        if (total_bytes > n_px * 16) HIP_TRY(ctx, hipMemsetAsync(reinterpret_cast<char*>(ctx->d_shard_in) + n_px * 16, 0, total_bytes - n_px * 16, st));
        if (!c.host_staged) {
            const int rc = c.alltoall(c.user, ctx->d_shard_in, slice_bytes, (void*)st);
            if (rc != 0) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's alltoall returned " + std::to_string(rc));
        } else {
            trc_status cs = staged(ctx, ctx->d_shard_in, total_bytes, true, st, what, [&](void* h) { return c.alltoall(c.user, h, slice_bytes, nullptr); });
            if (cs != TRC_OK) return cs;
        }
    } else {
        if (!ctx->comm) return trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + " before trc_group_init / trc_group_set_collectives");
        if (!g_rccl.Send || !g_rccl.Recv || !g_rccl.GroupStart || !g_rccl.GroupEnd)
            return trc_fail(ctx, TRC_ERR_UNSUPPORTED, std::string(what) + ": librccl has no ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
        if (mine) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_shard_in + (size_t)me * slice_f, src + (size_t)me * slice_f, mine * 16, hipMemcpyDeviceToDevice, st));
        int rc = g_rccl.GroupStart();
        for (int p = 0; p < N && rc == 0; ++p) {
            if (p == me) continue;
            const size_t theirs = slice_count(n_px, slice_px, p);
            if (theirs) rc = g_rccl.Send(src + (size_t)p * slice_f, theirs * 4, kNcclFloat, p, ctx->comm, st);
            if (rc == 0 && mine) rc = g_rccl.Recv(ctx->d_shard_in + (size_t)p * slice_f, mine * 4, kNcclFloat, p, ctx->comm, st);
        }
        const int rc_end = g_rccl.GroupEnd();
        if (rc != 0 || rc_end != 0) return trc_fail(ctx, TRC_ERR_RCCL, rccl_error(what, rc != 0 ? rc : rc_end));
    }
    // 2. fold the N texels of every pixel of the slice in rank order, divide by the number of sample groups
    if (mine) {
        hipLaunchKernelGGL(k_fold_shards, dim3((unsigned)((mine + 255) / 256)), dim3(256), 0, st, reinterpret_cast<const float4*>(ctx->d_shard_in),
                           reinterpret_cast<float4*>(ctx->d_shard_out + (size_t)me * slice_f), (uint32_t)mine, (uint32_t)slice_px, (uint32_t)N, (float)groups);
        HIP_TRY(ctx, hipGetLastError());
    }
    // 3. the composed slices to the root, or to everybody
    if (root < 0) return trc_coll_allgather(ctx, ctx->d_shard_out, slice_bytes, st, what);
    if (ctx->coll_active) {
        const trc_collectives& c = ctx->coll;
        if (!c.host_staged) {
            const int rc = c.gather(c.user, ctx->d_shard_out, slice_bytes, root, (void*)st);
            return rc == 0 ? TRC_OK : trc_fail(ctx, TRC_ERR_RCCL, std::string(what) + ": the caller's gather returned " + std::to_string(rc));
        }
        return staged(ctx, ctx->d_shard_out, total_bytes, me == root, st, what, [&](void* h) { return c.gather(c.user, h, slice_bytes, root, nullptr); });
    }
    int rc = g_rccl.GroupStart();
    if (me == root) {
        for (int p = 0; p < N && rc == 0; ++p) {
            const size_t theirs = slice_count(n_px, slice_px, p);
            if (p != me && theirs) rc = g_rccl.Recv(ctx->d_shard_out + (size_t)p * slice_f, theirs * 4, kNcclFloat, p, ctx->comm, st);
        }
    } else if (mine) {
        rc = g_rccl.Send(ctx->d_shard_out + (size_t)me * slice_f, mine * 4, kNcclFloat, root, ctx->comm, st);
    }
    const int rc_end = g_rccl.GroupEnd();
    if (rc != 0 || rc_end != 0) return trc_fail(ctx, TRC_ERR_RCCL, rccl_error(what, rc != 0 ? rc : rc_end));
    return TRC_OK;
}

trc_status check_compose(trc_ctx* ctx, int root, uint32_t groups, const char* what) {
    if (!ctx->grouped()) return fail(ctx, TRC_ERR_RCCL, std::string(what) + " before trc_group_init / trc_group_set_collectives");
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "no frame");
    if (root >= ctx->nranks) return fail(ctx, TRC_ERR_INVALID_ARG, std::string(what) + ": root");
    if (groups < 1 || (uint32_t)ctx->nranks % groups != 0) return fail(ctx, TRC_ERR_INVALID_ARG, std::string(what) + ": nranks is not sample_groups x tile ranks");
    return TRC_OK;
}

}  // namespace
}  // extern "C++"

uint64_t trc_shard_seed(uint64_t seed, uint32_t sample_group) { return seed + (uint64_t)sample_group * 0x9E3779B97F4A7C15ull; }

trc_status trc_group_compose_samples(trc_ctx* ctx, int root, uint32_t sample_groups) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || root < 0) return TRC_ERR_INVALID_ARG;
    if (sample_groups == 0) sample_groups = (uint32_t)ctx->nranks;
    { trc_status cs = check_compose(ctx, root, sample_groups, "trc_group_compose_samples"); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));    // an earlier pipelined compose still owns the slice buffers
    { trc_status cs = compose_samples(ctx, ctx->d_accum, root, sample_groups, ctx->stream, "compose of the sample shards"); if (cs != TRC_OK) return cs; }
    ctx->d_composed = ctx->rank == root ? ctx->d_shard_out : nullptr;      // the composed frame exists on the root only
    return TRC_OK;
}

trc_status trc_group_allreduce_mean_accum(trc_ctx* ctx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    { trc_status cs = check_compose(ctx, -1, (uint32_t)ctx->nranks, "trc_group_allreduce_mean_accum"); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    { trc_status cs = compose_samples(ctx, ctx->d_accum, -1, (uint32_t)ctx->nranks, ctx->stream, "compose of the sample shards"); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accum, ctx->d_shard_out, (size_t)ctx->width * ctx->height * 16, hipMemcpyDeviceToDevice, ctx->stream));
    return TRC_OK;
}

// Pipelined variant: the reduce of the frame just rendered runs on a second stream while the context goes on
// rendering into its OTHER accumulator, so an xGMI ring reduce of a multi-view frame (265 MB at N = 8, ~6 ms)
// hides under the next step's render instead of adding to it.
extern "C++" {
namespace {
// the frame just rendered goes to the communication stream (`collective` is queued there), the context to its other accumulator
template <typename Collective>
trc_status compose_async(trc_ctx* ctx, Collective&& collective) {
    const size_t count = (size_t)ctx->width * ctx->height * 4;
    if (!ctx->comm_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_rendered, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_busy, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_busy_alt, hipEventDisableTiming));
    }
    if (!ctx->d_accum_alt) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->d_accum_alt, count * sizeof(float)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum_alt, 0, count * sizeof(float), ctx->stream));
        ctx->busy_alt = false;
    }
    // compose the current accumulator once everything queued so far on the render stream has finished
    HIP_TRY(ctx, hipEventRecord(ctx->ev_rendered, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_rendered, 0));
    { trc_status cs = collective(); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_busy, ctx->comm_stream));
    ctx->busy = true;
    // swap accumulators (and their events); the render stream may touch the new current one only after ITS last compose
    std::swap(ctx->d_accum, ctx->d_accum_alt);
    std::swap(ctx->ev_busy, ctx->ev_busy_alt);
    std::swap(ctx->busy, ctx->busy_alt);
    if (ctx->busy) { HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_busy, 0)); ctx->busy = false; }
    return TRC_OK;
}
}  // namespace
}  // extern "C++"

trc_status trc_group_reduce_accum_async(trc_ctx* ctx, int root) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->grouped()) return fail(ctx, TRC_ERR_RCCL, "trc_group_reduce_accum_async before trc_group_init / trc_group_set_collectives");
    if (!ctx->d_accum) return fail(ctx, TRC_ERR_NO_FRAME, "no frame");
    if (root < 0 || root >= ctx->nranks) return TRC_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t count = (size_t)ctx->width * ctx->height * 4;
    float* frame = ctx->d_accum;
    trc_status s = compose_async(ctx, [&] { return trc_coll_reduce(ctx, frame, count, kNcclFloat, kNcclSum, root, ctx->comm_stream, "reduce(sum) of the accumulator"); });
    if (s == TRC_OK) ctx->d_composed = frame;
    return s;
}

trc_status trc_group_compose_samples_async(trc_ctx* ctx, int root, uint32_t sample_groups) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || root < 0) return TRC_ERR_INVALID_ARG;
    if (sample_groups == 0) sample_groups = (uint32_t)ctx->nranks;
    { trc_status cs = check_compose(ctx, root, sample_groups, "trc_group_compose_samples_async"); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // Unlike the tile reduce (which composes IN the accumulator and therefore switches to the other one), the sample compose only
    // reads: it works on a SNAPSHOT of the accumulator (a 33 MB device copy: ~20 us) taken in render-stream order, so the rank
    // goes on accumulating in place -- a progressive host calls this after every trc_render and never clears.
    const size_t bytes = (size_t)ctx->width * ctx->height * 16;
    if (!ctx->comm_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->comm_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_rendered, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_busy, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_busy_alt, hipEventDisableTiming));
    }
    if (!ctx->ev_snapshot_free) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_snapshot_free, hipEventDisableTiming));
    if (!ctx->d_shard_src) HIP_TRY(ctx, hipMalloc((void**)&ctx->d_shard_src, bytes));
    if (ctx->snapshot_busy) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_snapshot_free, 0));      // the previous compose still reads it
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_shard_src, ctx->d_accum, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_rendered, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->ev_rendered, 0));
    { trc_status cs = compose_samples(ctx, ctx->d_shard_src, root, sample_groups, ctx->comm_stream, "compose of the sample shards"); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_snapshot_free, ctx->comm_stream));
    ctx->snapshot_busy = true;
    ctx->d_composed = ctx->rank == root ? ctx->d_shard_out : nullptr;      // the composed frame exists on the root only
    return TRC_OK;
}

trc_status trc_download_composed(trc_ctx* ctx, float* rgba) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !rgba) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_composed) return fail(ctx, TRC_ERR_NO_FRAME, "trc_download_composed before trc_group_reduce_accum_async / trc_group_compose_samples");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->comm_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->comm_stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    { const trc_status cs = trc_copy_to_host(ctx, rgba, ctx->d_composed, (size_t)ctx->width * ctx->height * 16, ctx->stream); if (cs != TRC_OK) return cs; }
    return TRC_OK;
}

trc_status trc_group_finalize(trc_ctx* ctx) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (ctx->grouped()) {
        (void)hipSetDevice(ctx->device);
        if (ctx->comm_stream) (void)hipStreamSynchronize(ctx->comm_stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->comm) g_rccl.CommDestroy(ctx->comm);
        ctx->comm = nullptr;
        ctx->coll_active = false;
    }
    ctx->nranks = 1; ctx->rank = 0;
    return TRC_OK;
}

trc_status trc_group_set_collectives(trc_ctx* ctx, const trc_collectives* table, int nranks, int rank) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!table) return trc_group_finalize(ctx);
    if (nranks < 1 || rank < 0 || rank >= nranks) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_group_set_collectives: rank / nranks");
    if (!table->reduce || !table->allreduce || !table->allgather) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_group_set_collectives: the table needs reduce, allreduce and allgather (alltoall / gather: only for sample shards)");
    { trc_status fs = trc_group_finalize(ctx); if (fs != TRC_OK) return fs; }
    ctx->coll = *table;
    ctx->coll_active = true;
    ctx->nranks = nranks; ctx->rank = rank;
    return TRC_OK;
}

trc_status trc_debug_set(trc_ctx* ctx, const char* knob, int value) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx || !knob) return TRC_ERR_INVALID_ARG;
    const std::string k(knob);
    int* slot = k == "no_lds_fit" ? &ctx->knobs.no_lds_fit : k == "stack_lds_levels" ? &ctx->knobs.stack_lds_levels
              : k == "strip_len" ? &ctx->knobs.strip_len : k == "no_pwg" ? &ctx->knobs.no_pwg
              : k == "sppm_serial_camera" ? &ctx->knobs.sppm_serial_camera : k == "sppm_timing" ? &ctx->knobs.sppm_timing
              : k == "force_blk_shift" ? &ctx->knobs.force_blk_shift : k == "no_split" ? &ctx->knobs.no_split : k == "no_cost_filter" ? &ctx->knobs.no_cost_filter
              : k == "no_cold_probe" ? &ctx->knobs.no_cold_probe : k == "probe_spp" ? &ctx->knobs.probe_spp
              : k == "no_plan_reuse" ? &ctx->knobs.no_plan_reuse : k == "no_coalesce" ? &ctx->knobs.no_coalesce : k == "no_dense" ? &ctx->knobs.no_dense : k == "head_stages" ? &ctx->knobs.head_stages : k == "descend_min" ? &ctx->knobs.descend_min
              : k == "camera_policy" ? &ctx->knobs.camera_policy : nullptr;
    if (!slot) return fail(ctx, TRC_ERR_INVALID_ARG, "trc_debug_set: unknown knob " + k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));       // a launch in flight keeps the plan it was made with
    *slot = value < 0 ? 0 : value;
    ctx->cost_valid = false;                               // block costs recorded under another launch geometry say nothing
    ctx->plan_streak = 0; ctx->d_stale_order = nullptr;
    return TRC_OK;
}

}  // extern "C"

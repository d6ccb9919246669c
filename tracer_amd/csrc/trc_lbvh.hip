// trc_lbvh.hip -- on-device LBVH build + fat-node repack.
//
// The reference builds its BVH on the host (RT_Metal/Metal/BVH.hh:35-269) and lists "LBVHs, Morton Encoding"
// as its own to-do (RT_Metal/README.md:42).  trc_upload_scene_lbvh takes the LEAF records (what BVH::buildNode
// writes, BVH.hh:273-314) and builds the hierarchy on the GPU:
//
//   k_lbvh_bounds     centroid bounds of the leaf boxes            wave min/max -> ordered-uint atomics
//   k_lbvh_keys       30-bit Morton code of every centroid          key = code, value = leaf index
//   k_radix_*         stable LSD radix sort, 4 passes x 8 bits      (ties keep leaf-index order)
//   k_lbvh_hierarchy  T. Karras' binary radix tree (HPG 2012) over the 64-bit keys (code << 32 | index)
//   k_lbvh_refit_pass boxes + subtree heights bottom-up             one launch per level, no atomics / fences
//   k_lbvh_rotate_pass two sweeps of SAH tree rotations (Kensler 2008)  one launch per level, refit in between
//   k_lbvh_emit       fat nodes into the scene blob (dev_scene.hpp) + the tree in the reference's own array
//                     layout [root, leaf 0..n-1, interior 1..n-2] (BVH.hh:246-269) for trc_download_bvh
//
// All of it is integer / min-max work on a few bytes per leaf: HBM-bound streaming passes, no MFMA.  The CPU
// statement of the same build is oracle/oracle_lbvh.cpp; tests compare all 2n-1 records bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "trc_ctx.hpp"
#include "trc_scene_prep.hpp"

namespace {

constexpr int kSortBlock = 256;
constexpr int kSortItems = 16;                         // keys per thread per tile
constexpr int kSortTile = kSortBlock * kSortItems;     // 4096 keys per workgroup

struct DLeaf {            // 32 B: what the build needs of a 64-B leaf record
    float mn[3]; uint32_t tag;      // tag = pType << 29 | pIndex
    float mx[3]; uint32_t _pad;
};

__device__ __forceinline__ uint32_t ordered_of(float f) {      // monotone float -> uint map
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float float_of(uint32_t o) {
    return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__device__ __forceinline__ void leaf_centroid(const DLeaf& l, float c[3]) {
    c[0] = (l.mn[0] + l.mx[0]) * 0.5f; c[1] = (l.mn[1] + l.mx[1]) * 0.5f; c[2] = (l.mn[2] + l.mx[2]) * 0.5f;
}

// leaf records as uploaded (reference layout, ref_leaves = slot 1 of the output array) -> compact DLeaf; checks
// what trc_upload_scene checks on the host (primitive type / index range) and clears the link fields
struct LeafLimits { uint32_t n[4]; };      // spheres, squares, cubes, triangles
__global__ void __launch_bounds__(256) k_lbvh_prepare(trc_BVH* ref_leaves, uint32_t n, LeafLimits lim, DLeaf* out, uint32_t* bad) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    trc_BVH& r = ref_leaves[i];
    const int32_t t = r.pType;
    const uint32_t pi = r.pIndex;
    if (t == TRC_PRIM_BVH) atomicOr(bad, 1u);
    else if (t < 0 || t > TRC_PRIM_TRIANGLE || pi >= lim.n[t]) atomicOr(bad, 2u);
    else if (pi > kTagIndexMask) atomicOr(bad, 4u);
    DLeaf d;
    d.mn[0] = r.bBOX.mini.x; d.mn[1] = r.bBOX.mini.y; d.mn[2] = r.bBOX.mini.z;
    d.mx[0] = r.bBOX.maxi.x; d.mx[1] = r.bBOX.maxi.y; d.mx[2] = r.bBOX.maxi.z;
    d.tag = ((uint32_t)t << kTagIndexBits) | (pi & kTagIndexMask); d._pad = 0;
    out[i] = d;
    r.parent = 0; r.left = 0; r.right = 0;
}

// bounds[0..2] = min, bounds[3..5] = max of the centroids, as ordered uints (init: 0xFFFFFFFF / 0).
// Grid-stride loop, wavefront shuffle reduction, LDS across the 4 wavefronts, 6 atomics per workgroup.
__global__ void __launch_bounds__(256) k_lbvh_bounds(const DLeaf* leaves, uint32_t n, uint32_t* bounds) {
    __shared__ uint32_t part[4][6];
    uint32_t lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        float c[3];
        leaf_centroid(leaves[i], c);
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (c[a] == c[a]) {                                            // NaN centroids do not take part
                const uint32_t o = ordered_of(c[a]);
                lo[a] = min(lo[a], o); hi[a] = max(hi[a], o);
            }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = min(lo[a], (uint32_t)__shfl_xor((int)lo[a], off, 64));
            hi[a] = max(hi[a], (uint32_t)__shfl_xor((int)hi[a], off, 64));
        }
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { part[wave][a] = lo[a]; part[wave][3 + a] = hi[a]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const uint32_t a = threadIdx.x;
        atomicMin(&bounds[a], min(min(part[0][a], part[1][a]), min(part[2][a], part[3][a])));
        atomicMax(&bounds[3 + a], max(max(part[0][3 + a], part[1][3 + a]), max(part[2][3 + a], part[3][3 + a])));
    }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t quantise(float c, float lo, float ext) {
    if (!(ext > 0.0f)) return 0;
    float v = ((c - lo) / ext) * 1024.0f;
    if (!(v >= 0.0f)) v = 0.0f;
    if (v > 1023.0f) v = 1023.0f;
    return (uint32_t)v;
}

__global__ void __launch_bounds__(256) k_lbvh_keys(const DLeaf* leaves, uint32_t n, const uint32_t* bounds,
                                                  uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float lo[3], ext[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { lo[a] = float_of(bounds[a]); ext[a] = float_of(bounds[3 + a]) - lo[a]; }
    leaf_centroid(leaves[i], c);
    keys[i] = (expand_bits10(quantise(c[0], lo[0], ext[0])) << 2) | (expand_bits10(quantise(c[1], lo[1], ext[1])) << 1) |
              expand_bits10(quantise(c[2], lo[2], ext[2]));
    vals[i] = i;
}

// ---- stable LSD radix sort, one 8-bit digit per pass.  hist is digit-major: hist[digit * n_blocks + block].
__global__ void __launch_bounds__(kSortBlock) k_radix_hist(const uint32_t* keys, uint32_t n, uint32_t shift, uint32_t* hist, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t i = base + r * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];
}

// hist[digit][block] -> exclusive scan along the row of each digit (one workgroup per digit) + row totals;
// k_radix_digit_base then scans the 256 totals.  Output slot of a key = digit_base[d] + hist[d][block] + rank.
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)v, off, 64);
        if (lane >= (uint32_t)off) v += t;
    }
    return v;
}
__device__ __forceinline__ uint32_t block_exclusive_scan256(uint32_t v, uint32_t* wave_tot /* [4] LDS */, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t w = 0; w < wave; ++w) before += wave_tot[w];
    total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    __syncthreads();
    return before + inc - v;
}
__global__ void __launch_bounds__(256) k_radix_row_scan(uint32_t* hist, uint32_t n_blocks, uint32_t* row_total) {
    __shared__ uint32_t wave_tot[4];
    uint32_t* row = hist + (size_t)blockIdx.x * n_blocks;
    uint32_t carry = 0;
    for (uint32_t b = 0; b < n_blocks; b += 256) {
        const uint32_t i = b + threadIdx.x;
        const uint32_t v = i < n_blocks ? row[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan256(v, wave_tot, total);
        if (i < n_blocks) row[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) row_total[blockIdx.x] = carry;
}
__global__ void __launch_bounds__(256) k_radix_digit_base(uint32_t* row_total) {
    __shared__ uint32_t wave_tot[4];
    uint32_t total;
    row_total[threadIdx.x] = block_exclusive_scan256(row_total[threadIdx.x], wave_tot, total);
}

__global__ void __launch_bounds__(kSortBlock) k_radix_scatter(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                                                             uint32_t* vals_out, uint32_t n, uint32_t shift, const uint32_t* hist,
                                                             const uint32_t* digit_base, uint32_t n_blocks) {
    __shared__ uint32_t run[256];            // next output slot of each digit for this tile
    __shared__ uint32_t wave_cnt[4][256];    // keys of each digit per wavefront in the current round
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    run[threadIdx.x] = digit_base[threadIdx.x] + hist[threadIdx.x * n_blocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < 4; ++w) wave_cnt[w][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * kSortTile;
    for (int r = 0; r < kSortItems; ++r) {
        const uint32_t i = base + r * kSortBlock + threadIdx.x;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u, val = valid ? vals_in[i] : 0u;
        const uint32_t digit = (key >> shift) & 255u;
        // lanes of this wavefront that hold the same digit (and are valid)
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (digit >> b) & 1u;
            const unsigned long long m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
        if (valid && rank == 0) wave_cnt[wave][digit] = (uint32_t)__popcll(peers);
        __syncthreads();
        if (valid) {
            uint32_t off = run[digit] + rank;
            for (uint32_t w = 0; w < wave; ++w) off += wave_cnt[w][digit];
            keys_out[off] = key;
            vals_out[off] = val;
        }
        __syncthreads();
        {
            uint32_t s = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s += wave_cnt[w][threadIdx.x]; wave_cnt[w][threadIdx.x] = 0; }
            run[threadIdx.x] += s;
        }
        __syncthreads();
    }
}

// ---- Karras 2012.  Child encoding: bit 31 = leaf, low bits = sorted position (leaf) or interior index.
constexpr uint32_t kChildLeaf = 0x80000000u;

struct DTopo {
    uint32_t* child_l; uint32_t* child_r;    // [n-1]
    uint32_t* parent_interior;               // [n-1] parent interior index of interior i (root: 0)
    uint32_t* parent_leaf;                   // [n]   parent interior index of sorted leaf position p (kept for tools)
    uint32_t* axis;                          // [n-1]
};

__device__ __forceinline__ int lbvh_delta(const uint32_t* keys, const uint32_t* vals, uint64_t ki, int64_t j, uint32_t n) {
    if (j < 0 || j >= (int64_t)n) return -1;
    const uint64_t kj = ((uint64_t)keys[j] << 32) | vals[j];
    const uint64_t x = ki ^ kj;
    return x ? __clzll((long long)x) : 64;
}

__global__ void __launch_bounds__(256) k_lbvh_hierarchy(const uint32_t* keys, const uint32_t* vals, uint32_t n, DTopo tp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 >= (int64_t)n) return;
    const uint64_t ki = ((uint64_t)keys[i] << 32) | vals[i];
    auto delta = [&](int64_t j) { return lbvh_delta(keys, vals, ki, j, n); };
    const int d = (delta(i + 1) - delta(i - 1)) < 0 ? -1 : 1;
    const int dmin = delta(i - d);
    int64_t lmax = 2;
    while (delta(i + lmax * d) > dmin) lmax *= 2;
    int64_t l = 0;
    for (int64_t t = lmax / 2; t >= 1; t /= 2)
        if (delta(i + (l + t) * d) > dmin) l += t;
    const int64_t j = i + l * d;
    const int dnode = delta(j);
    int64_t s = 0;
    for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(i + (s + t) * d) > dnode) s += t;
        if (t <= 1) break;
    }
    const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
    const int64_t first = i < j ? i : j, last = i < j ? j : i;
    const bool leaf_l = first == gamma, leaf_r = last == gamma + 1;
    tp.child_l[i] = (uint32_t)gamma | (leaf_l ? kChildLeaf : 0u);
    tp.child_r[i] = (uint32_t)(gamma + 1) | (leaf_r ? kChildLeaf : 0u);
    if (leaf_l) tp.parent_leaf[gamma] = (uint32_t)i; else tp.parent_interior[gamma] = (uint32_t)i;
    if (leaf_r) tp.parent_leaf[gamma + 1] = (uint32_t)i; else tp.parent_interior[gamma + 1] = (uint32_t)i;
    const int mbit = dnode - 2;
    tp.axis[i] = (mbit >= 0 && mbit < 30) ? (uint32_t)(mbit % 3) : 0u;
    if (i == 0) tp.parent_interior[0] = 0;
}

// boxes[i] = 6 floats (min xyz, max xyz) of interior i; height[i] = depth of the deepest leaf below i.
// Bottom-up in PASSES, one kernel launch each: a node is fitted in pass p when both children are leaves or were
// fitted in a pass < p (done[i] = pass of fitting, 0 = not yet).  Launch boundaries are the only synchronisation:
// the usual single-kernel climb with one atomic counter per node needs two device-scope fences per level, and on
// the 8-XCD part each fence writes back / invalidates an L2 (measured 6.6 ms for 1 M leaves vs 0.4 ms like this).
// Which of two EQUAL bounds a merged box keeps matters when they are -0 and +0 (the record is compared bit for bit): the
// oracle of the LBVH build merges with std::min / std::max (the FIRST operand on a tie), the reference's SAH build and its
// restatements with fminf / fmaxf, which on the host compile to minss / maxss (the SECOND operand on a tie).  v_min_f32 would
// pick -0 whatever its position.
template <bool SECOND_ON_TIE> __device__ __forceinline__ float tie_min(float a, float b) { return SECOND_ON_TIE ? (a < b ? a : b) : (b < a ? b : a); }
template <bool SECOND_ON_TIE> __device__ __forceinline__ float tie_max(float a, float b) { return SECOND_ON_TIE ? (a > b ? a : b) : (a < b ? b : a); }
template <bool SAH>
__global__ void __launch_bounds__(256) k_lbvh_refit_pass(const DLeaf* leaves, const uint32_t* vals, uint32_t n, DTopo tp,
                                                        float* boxes, uint32_t* height, uint32_t* done, uint32_t pass) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1 >= n || done[i] != 0u) return;
    const uint32_t ch[2] = {tp.child_l[i], tp.child_r[i]};
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if (!(ch[k] & kChildLeaf)) { const uint32_t d = done[ch[k]]; if (d == 0u || d >= pass) return; }
    float mn[3], mx[3];
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float cmn[3], cmx[3];
        uint32_t hc;
        if (ch[k] & kChildLeaf) {
            const DLeaf& lf = leaves[vals[ch[k] & ~kChildLeaf]];
            cmn[0] = lf.mn[0]; cmn[1] = lf.mn[1]; cmn[2] = lf.mn[2]; cmx[0] = lf.mx[0]; cmx[1] = lf.mx[1]; cmx[2] = lf.mx[2];
            hc = 1;
        } else {
            const float* b = boxes + (size_t)ch[k] * 6;
            cmn[0] = b[0]; cmn[1] = b[1]; cmn[2] = b[2]; cmx[0] = b[3]; cmx[1] = b[4]; cmx[2] = b[5];
            hc = height[ch[k]] + 1;
        }
        if (k == 0) { for (int a = 0; a < 3; ++a) { mn[a] = cmn[a]; mx[a] = cmx[a]; } h = hc; }
        else { for (int a = 0; a < 3; ++a) { mn[a] = tie_min<SAH>(mn[a], cmn[a]); mx[a] = tie_max<SAH>(mx[a], cmx[a]); } h = max(h, hc); }
    }
    float* b = boxes + (size_t)i * 6;
    b[0] = mn[0]; b[1] = mn[1]; b[2] = mn[2]; b[3] = mx[0]; b[4] = mx[1]; b[5] = mx[2];
    height[i] = h;
    done[i] = pass;
}

// ---- tree rotations (A. Kensler, "Tree Rotations for Improving Bounding Volume Hierarchies", RT 2008)
// One bottom-up sweep: every interior node, after all nodes below it, may swap one of its children with a grandchild
// on the other side when that shrinks the surface area of the child that is rebuilt (greedy: best of the <= 4 swaps,
// candidates in a fixed order, strict improvement).  Nodes of equal height have disjoint subtrees, so a sweep is one
// launch per height of the refit that precedes it (sched[i] = height of i then); a rotation changes heights, so
// the tree is refitted before the next sweep.  A Morton-order tree gains most near its top, where the split planes
// ignore the geometry: on the 1 M-triangle scene two sweeps take the box steps per ray from 10.6 to 9.4 (the
// reference's SAH builder: 10.3) and the frame from 20.2 to 19.4 ms, for 1.1 ms of build time.
constexpr int kRotationSweeps = 2;
__device__ __forceinline__ void lbvh_child_box(const DLeaf* leaves, const uint32_t* vals, const float* boxes, uint32_t c, float b[6]) {
    if (c & kChildLeaf) {
        const DLeaf& lf = leaves[vals[c & ~kChildLeaf]];
        b[0] = lf.mn[0]; b[1] = lf.mn[1]; b[2] = lf.mn[2]; b[3] = lf.mx[0]; b[4] = lf.mx[1]; b[5] = lf.mx[2];
    } else {
        const float* q = boxes + (size_t)c * 6;
#pragma unroll
        for (int a = 0; a < 6; ++a) b[a] = q[a];
    }
}
__device__ __forceinline__ float lbvh_area(const float b[6]) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return 2.0f * ((dx * dy + dy * dz) + dz * dx);
}
__device__ __forceinline__ float lbvh_merged_area(const float a[6], const float b[6]) {
    float m[6];
#pragma unroll
    for (int k = 0; k < 3; ++k) { m[k] = fminf(a[k], b[k]); m[3 + k] = fmaxf(a[3 + k], b[3 + k]); }
    return lbvh_area(m);
}
__device__ __forceinline__ void lbvh_set_parent(DTopo tp, uint32_t c, uint32_t parent) {
    if (c & kChildLeaf) tp.parent_leaf[c & ~kChildLeaf] = parent; else tp.parent_interior[c] = parent;
}
__global__ void __launch_bounds__(256) k_lbvh_rotate_pass(const DLeaf* leaves, const uint32_t* vals, uint32_t n, DTopo tp, float* boxes,
                                                         const uint32_t* sched, uint32_t pass) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1 >= n || sched[i] != pass) return;
    const uint32_t L = tp.child_l[i], R = tp.child_r[i];
    float bl[6], br[6];
    lbvh_child_box(leaves, vals, boxes, L, bl);
    lbvh_child_box(leaves, vals, boxes, R, br);
    float best = 0.0f;
    int which = -1;
    uint32_t g0 = 0, g1 = 0;                       // the two children of the interior child of the best swap
    if (!(R & kChildLeaf)) {
        const uint32_t RL = tp.child_l[R], RR = tp.child_r[R];
        float b0[6], b1[6];
        lbvh_child_box(leaves, vals, boxes, RL, b0);
        lbvh_child_box(leaves, vals, boxes, RR, b1);
        const float old = lbvh_area(br);
        const float a0 = lbvh_merged_area(bl, b1);      // swap 0, L <-> RL: R' = (L, RR)
        const float a1 = lbvh_merged_area(b0, bl);      // swap 1, L <-> RR: R' = (RL, L)
        if (old - a0 > best) { best = old - a0; which = 0; g0 = RL; g1 = RR; }
        if (old - a1 > best) { best = old - a1; which = 1; g0 = RL; g1 = RR; }
    }
    if (!(L & kChildLeaf)) {
        const uint32_t LL = tp.child_l[L], LR = tp.child_r[L];
        float b0[6], b1[6];
        lbvh_child_box(leaves, vals, boxes, LL, b0);
        lbvh_child_box(leaves, vals, boxes, LR, b1);
        const float old = lbvh_area(bl);
        const float a2 = lbvh_merged_area(br, b1);      // swap 2, R <-> LL: L' = (R, LR)
        const float a3 = lbvh_merged_area(b0, br);      // swap 3, R <-> LR: L' = (LL, R)
        if (old - a2 > best) { best = old - a2; which = 2; g0 = LL; g1 = LR; }
        if (old - a3 > best) { best = old - a3; which = 3; g0 = LL; g1 = LR; }
    }
    if (which < 0) return;
    // X = the interior child that is rebuilt, `up` = its child that moves up to i, `keep` = the one that stays,
    // `down` = i's other child, which takes the place of `up` below X
    const bool right_side = which < 2;
    const uint32_t X = right_side ? R : L, down = right_side ? L : R;
    const bool first = (which == 0 || which == 2);      // the grandchild that moves up is X's left child
    const uint32_t up = first ? g0 : g1, keep = first ? g1 : g0;
    if (right_side) tp.child_l[i] = up; else tp.child_r[i] = up;
    lbvh_set_parent(tp, up, i);
    if (first) tp.child_l[X] = down; else tp.child_r[X] = down;
    lbvh_set_parent(tp, down, X);
    float bd[6], bk[6];
    lbvh_child_box(leaves, vals, boxes, down, bd);
    lbvh_child_box(leaves, vals, boxes, keep, bk);
    float* q = boxes + (size_t)X * 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) { q[k] = tie_min<false>(bd[k], bk[k]); q[3 + k] = tie_max<false>(bd[3 + k], bk[3 + k]); }      // merged(down, keep)
}

// depth of interior node i (root 0): a walk up the parent links, <= TRC_MAX_BVH_DEPTH steps
__global__ void __launch_bounds__(256) k_lbvh_depth(uint32_t n_interior, const uint32_t* parent_interior, uint32_t* keys, uint32_t* vals) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_interior) return;
    uint32_t d = 0;
    for (uint32_t j = i; j != 0u && d < 255u; j = parent_interior[j]) ++d;
    keys[i] = d; vals[i] = i;
}
// rank[interior] = its position in the depth-sorted order
__global__ void __launch_bounds__(256) k_lbvh_rank(uint32_t n_interior, const uint32_t* sorted_vals, uint32_t* rank) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p < n_interior) rank[sorted_vals[p]] = p;
}

// fat node of interior i (dev_scene.hpp) + its record in the reference layout; one thread per interior node.  The fat nodes
// are numbered by DEPTH (rank[]: a stable sort of the interior nodes by their depth, root = 0), not by Karras index: any
// prefix of the array is then a top of the tree, which is what the render kernels stage in LDS (stage_scene, k_render_pwg)
// -- the numbering host-built trees get from their BFS walk (trc_abi.hip::build_blob).
__global__ void __launch_bounds__(256) k_lbvh_emit(const DLeaf* leaves, const uint32_t* vals, uint32_t n, DTopo tp, const float* boxes,
                                                  const uint32_t* rank, uint32_t* blob_nodes, trc_BVH* ref) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i + 1 >= n) return;
    const uint32_t ch[2] = {tp.child_l[i], tp.child_r[i]};
    float cb[2][6];
    uint32_t tag[2], slot[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint32_t c = ch[k] & ~kChildLeaf;
        if (ch[k] & kChildLeaf) {
            const uint32_t leaf = vals[c];
            const DLeaf& lf = leaves[leaf];
            cb[k][0] = lf.mn[0]; cb[k][1] = lf.mn[1]; cb[k][2] = lf.mn[2]; cb[k][3] = lf.mx[0]; cb[k][4] = lf.mx[1]; cb[k][5] = lf.mx[2];
            tag[k] = lf.tag;
            slot[k] = leaf + 1u;
        } else {
            const float* b = boxes + (size_t)c * 6;
#pragma unroll
            for (int a = 0; a < 6; ++a) cb[k][a] = b[a];
            tag[k] = (kTagInterior << kTagIndexBits) | rank[c];
            slot[k] = n + c;                                   // interior c >= 1 here (0 is the root)
        }
    }
    uint32_t* q = blob_nodes + (size_t)rank[i] * kNodeDwords;
    q[0] = __float_as_uint(cb[0][0]); q[1] = __float_as_uint(cb[0][1]); q[2] = __float_as_uint(cb[0][2]); q[3] = __float_as_uint(cb[0][3]);
    q[4] = __float_as_uint(cb[0][4]); q[5] = __float_as_uint(cb[0][5]); q[6] = __float_as_uint(cb[1][0]); q[7] = __float_as_uint(cb[1][1]);
    q[8] = __float_as_uint(cb[1][2]); q[9] = __float_as_uint(cb[1][3]); q[10] = __float_as_uint(cb[1][4]); q[11] = __float_as_uint(cb[1][5]);
    q[12] = 0; q[13] = 0; q[14] = tag[0]; q[15] = tag[1];

    const uint32_t self = i == 0 ? 0u : n + i;
    trc_BVH nd;
    memset(&nd, 0, sizeof nd);                                 // padding bytes of the POD are part of the compared record
    const uint32_t pi = tp.parent_interior[i];
    nd.parent = i == 0 ? 0u : (pi == 0 ? 0u : n + pi);
    nd.left = slot[0]; nd.right = slot[1];
    nd.axis = tp.axis[i];
    nd.pType = TRC_PRIM_BVH; nd.pIndex = 0; nd._pad[0] = 0; nd._pad[1] = 0;
    const float* b = boxes + (size_t)i * 6;
    nd.bBOX.mini.x = b[0]; nd.bBOX.mini.y = b[1]; nd.bBOX.mini.z = b[2];
    nd.bBOX.maxi.x = b[3]; nd.bBOX.maxi.y = b[4]; nd.bBOX.maxi.z = b[5];
    ref[self] = nd;
    if (ch[0] & kChildLeaf) ref[slot[0]].parent = self;      // leaf records were uploaded with parent 0;
    if (ch[1] & kChildLeaf) ref[slot[1]].parent = self;      // interior children write their own record
}

struct Buffers {
    std::vector<void*> ptrs;
    ~Buffers() { for (void* p : ptrs) (void)hipFree(p); }
    template <class T> hipError_t alloc(T** out, size_t count) {
        hipError_t e = hipMalloc((void**)out, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(*out);
        return e;
    }
};

#include "trc_sah_build.hpp"

}  // namespace

// stable LSD sort of (key, value) pairs on the 24 low key bits, 3 passes of the radix kernels above; the result is in
// buffer *result (0 or 1) of the two ping-pong arrays.  hist: 256 * ceil(n / 4096) words, digit_base: 256 words.
void trc_sort_pairs24(hipStream_t st, uint32_t* keys[2], uint32_t* vals[2], uint32_t* hist, uint32_t* digit_base, uint32_t n, int* result) {
    const uint32_t n_sort_blocks = (n + kSortTile - 1) / kSortTile;
    int cur = 0;
    for (uint32_t shift = 0; shift < 24; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(n_sort_blocks), dim3(kSortBlock), 0, st, keys[cur], n, shift, hist, n_sort_blocks);
        hipLaunchKernelGGL(k_radix_row_scan, dim3(256), dim3(256), 0, st, hist, n_sort_blocks, digit_base);
        hipLaunchKernelGGL(k_radix_digit_base, dim3(1), dim3(256), 0, st, digit_base);
        hipLaunchKernelGGL(k_radix_scatter, dim3(n_sort_blocks), dim3(kSortBlock), 0, st, keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], n,
                           shift, hist, digit_base, n_sort_blocks);
        cur ^= 1;
    }
    *result = cur;
}
uint32_t trc_sort_hist_words(uint32_t n) { return 256u * ((n + kSortTile - 1) / kSortTile); }

// trc_upload_scene_lbvh (sah = false: Morton order, radix tree, rotations) and trc_upload_scene_sah (sah = true: the
// reference's own binned-SAH build, trc_sah_build.hpp) share everything around the topology: leaf intake, boxes bottom-up,
// fat nodes by depth, the tree in the reference's array layout.
static trc_status upload_device_tree(trc_ctx* ctx, const trc_scene* s, bool sah, bool triangle_leaves) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!s) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "device tree: no scene");
    // triangle_leaves: bvhList holds the analytic primitives' leaves only; one leaf per triangle follows them, written on the device
    const uint64_t n_given = s->bvhList ? s->n_bvh : 0u, n_all = n_given + (triangle_leaves ? s->n_index / 3 : 0u);
    if ((!s->bvhList && !triangle_leaves) || n_all < 2) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "lbvh: need >= 2 leaf records");
    if (n_all > (1u << 28)) return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "lbvh: more than 2^28 leaves");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { trc_status st = validate_primitives(ctx, s); if (st != TRC_OK) return st; }
    const uint32_t n = (uint32_t)n_all, n_interior = n - 1, n_nodes = 2 * n - 1;

    DScene sc{};
    uint64_t total = 0;
    { trc_status st = layout_scene(ctx, s, n_interior, sc, total); if (st != TRC_OK) return st; }
    // host side of the blob: analytic primitives + materials (the prefix).  The fat nodes are written by k_lbvh_emit, the
    // triangle records by k_repack_triangles from the caller's vertex / index arrays: neither is staged on the host
    std::unique_ptr<uint32_t[]> blob(new (std::nothrow) uint32_t[(size_t)sc.off_nodes]);
    if (!blob) return trc_fail(ctx, TRC_ERR_OOM, "lbvh: host staging buffer");
    std::memset(blob.get(), 0, (size_t)sc.off_nodes * 4);
    fill_primitives(s, sc, blob.get());

    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_blob) { (void)hipFree(ctx->d_blob); ctx->d_blob = nullptr; }
    if (ctx->d_bvh_ref) { (void)hipFree(ctx->d_bvh_ref); ctx->d_bvh_ref = nullptr; }
    ctx->has_scene = false; ctx->n_bvh_ref = 0;
    ctx->blob_bytes = (size_t)total * 4;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_blob, ctx->blob_bytes));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_bvh_ref, sizeof(trc_BVH) * n_nodes));
    hipStream_t st = ctx->stream;
    { const trc_status cs = trc_copy_to_device(ctx, ctx->d_blob, blob.get(), (size_t)sc.off_nodes * 4, st); if (cs != TRC_OK) return cs; }
    // the caller's leaf records go straight to slots 1..n of the reference-layout array (BVH.hh:246-269), the triangles' behind them
    { trc_status rs = trc_repack_triangles(ctx, s, sc, ctx->d_blob, triangle_leaves ? ctx->d_bvh_ref + 1 + n_given : nullptr); if (rs != TRC_OK) return rs; }
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_bvh_ref, 0, sizeof(trc_BVH), st));
    if (n_given) { const trc_status cs = trc_copy_to_device(ctx, ctx->d_bvh_ref + 1, s->bvhList, sizeof(trc_BVH) * n_given, st); if (cs != TRC_OK) return cs; }

    Buffers buf;
    DLeaf* d_leaves; uint32_t *d_keys[2], *d_vals[2], *d_hist, *d_bounds, *d_height, *d_arrived;
    float* d_boxes;
    DTopo tp{};
    const uint32_t n_sort_blocks = (n + kSortTile - 1) / kSortTile;
    HIP_TRY(ctx, buf.alloc(&d_leaves, n));
    for (int k = 0; k < 2; ++k) { HIP_TRY(ctx, buf.alloc(&d_keys[k], n)); HIP_TRY(ctx, buf.alloc(&d_vals[k], n)); }
    HIP_TRY(ctx, buf.alloc(&d_hist, (size_t)256 * n_sort_blocks));
    uint32_t* d_digit_base;
    HIP_TRY(ctx, buf.alloc(&d_digit_base, 256));
    HIP_TRY(ctx, buf.alloc(&d_bounds, 8));       // 6 ordered bounds + [6] = bad-leaf flags
    HIP_TRY(ctx, buf.alloc(&d_height, n_interior));
    HIP_TRY(ctx, buf.alloc(&d_arrived, n_interior));
    HIP_TRY(ctx, buf.alloc(&d_boxes, (size_t)n_interior * 6));
    HIP_TRY(ctx, buf.alloc(&tp.child_l, n_interior)); HIP_TRY(ctx, buf.alloc(&tp.child_r, n_interior));
    HIP_TRY(ctx, buf.alloc(&tp.parent_interior, n_interior)); HIP_TRY(ctx, buf.alloc(&tp.parent_leaf, n));
    HIP_TRY(ctx, buf.alloc(&tp.axis, n_interior));
    const uint32_t bounds_init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    HIP_TRY(ctx, hipMemcpyAsync(d_bounds, bounds_init, sizeof bounds_init, hipMemcpyHostToDevice, st));

    struct Events {                 // destroyed on every exit path
        hipEvent_t a = nullptr, b = nullptr;
        ~Events() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } ev;
    HIP_TRY(ctx, hipEventCreate(&ev.a)); HIP_TRY(ctx, hipEventCreate(&ev.b));
    const hipEvent_t e0 = ev.a, e1 = ev.b;
    HIP_TRY(ctx, hipEventRecord(e0, st));
    const dim3 g_leaf((n + 255) / 256), g_int((n_interior + 255) / 256), b256(256);
    LeafLimits lim;
    lim.n[0] = s->n_sphere; lim.n[1] = s->n_square; lim.n[2] = s->n_cube; lim.n[3] = s->n_index / 3;
    hipLaunchKernelGGL(k_lbvh_prepare, g_leaf, b256, 0, st, ctx->d_bvh_ref + 1, n, lim, d_leaves, d_bounds + 6);
    int cur = 0;
    uint32_t first_chunk = 40;                  // refit passes before the root is looked at for the first time
    if (!sah) {
        hipLaunchKernelGGL(k_lbvh_bounds, dim3(std::min<uint32_t>((n + 255) / 256, 1024u)), b256, 0, st, d_leaves, n, d_bounds);
        hipLaunchKernelGGL(k_lbvh_keys, g_leaf, b256, 0, st, d_leaves, n, d_bounds, d_keys[0], d_vals[0]);
        for (uint32_t shift = 0; shift < 32; shift += 8) {
            hipLaunchKernelGGL(k_radix_hist, dim3(n_sort_blocks), dim3(kSortBlock), 0, st, d_keys[cur], n, shift, d_hist, n_sort_blocks);
            hipLaunchKernelGGL(k_radix_row_scan, dim3(256), b256, 0, st, d_hist, n_sort_blocks, d_digit_base);
            hipLaunchKernelGGL(k_radix_digit_base, dim3(1), b256, 0, st, d_digit_base);
            hipLaunchKernelGGL(k_radix_scatter, dim3(n_sort_blocks), dim3(kSortBlock), 0, st, d_keys[cur], d_vals[cur], d_keys[cur ^ 1],
                               d_vals[cur ^ 1], n, shift, d_hist, d_digit_base, n_sort_blocks);
            cur ^= 1;
        }
        hipLaunchKernelGGL(k_lbvh_hierarchy, g_int, b256, 0, st, d_keys[cur], d_vals[cur], n, tp);
    } else {
        const trc_status bs = sah_build_topology(ctx, buf, d_leaves, n, d_bounds + 6, tp, d_vals[0], &first_chunk);
        if (bs != TRC_OK) return bs;
        first_chunk = std::min(std::max(first_chunk, 1u), TRC_MAX_BVH_DEPTH + 1u);      // the builder knows the tree's height: that many refit passes
    }
    // refit passes: a tree of height h needs h passes; check the root every few passes beyond the usual depth
    uint32_t root_done = 0, bad_leaves = 0;
    const uint32_t pass_limit = TRC_MAX_BVH_DEPTH + 1;
    auto refit = [&]() -> trc_status {
        uint32_t pass = 0;
        root_done = 0;
        HIP_TRY(ctx, hipMemsetAsync(d_arrived, 0, sizeof(uint32_t) * n_interior, st));
        for (uint32_t chunk = first_chunk; pass < pass_limit && !root_done; chunk = 8) {
            for (uint32_t k = 0; k < chunk && pass < pass_limit; ++k)
                if (sah) hipLaunchKernelGGL(k_lbvh_refit_pass<true>, g_int, b256, 0, st, d_leaves, d_vals[cur], n, tp, d_boxes, d_height, d_arrived, ++pass);
                else hipLaunchKernelGGL(k_lbvh_refit_pass<false>, g_int, b256, 0, st, d_leaves, d_vals[cur], n, tp, d_boxes, d_height, d_arrived, ++pass);
            HIP_TRY(ctx, hipMemcpyAsync(&root_done, d_arrived, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipMemcpyAsync(&bad_leaves, d_bounds + 6, 4, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipStreamSynchronize(st));
            if (bad_leaves & 1u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "lbvh: input must be leaf records only");
            if (bad_leaves & 2u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "bvh: leaf with bad primitive type/index");
            if (bad_leaves & 4u) return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "bvh: primitive index exceeds 29 bits");
            if (bad_leaves & 8u) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "sah: leaf box not finite or beyond 1e37");
        }
        if (!root_done) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "lbvh: tree deeper than TRC_MAX_BVH_DEPTH");
        return TRC_OK;
    };
    { trc_status rs = refit(); if (rs != TRC_OK) return rs; }
    for (int sweep = 0; sweep < (sah ? 0 : kRotationSweeps); ++sweep) {
        // d_arrived[i] = pass in which i was fitted = its height; the root's is the height of the tree.  Nodes of
        // height 1 have two leaf children and nothing to rotate.
        const uint32_t h = root_done;
        for (uint32_t pass = 2; pass <= h; ++pass)
            hipLaunchKernelGGL(k_lbvh_rotate_pass, g_int, b256, 0, st, d_leaves, d_vals[cur], n, tp, d_boxes, d_arrived, pass);
        trc_status rs = refit();
        if (rs != TRC_OK) return rs;
    }
    // fat-node numbering: interior nodes sorted by depth (one stable 8-bit radix pass; the root is the only node of depth 0)
    uint32_t* d_rank = nullptr;                         // reuses the arrival array (the heights in d_height are read back below)
    {
        uint32_t* dk[2] = {d_keys[cur ^ 1], nullptr};
        uint32_t* dv[2] = {d_vals[cur ^ 1], nullptr};
        HIP_TRY(ctx, buf.alloc(&dk[1], n_interior)); HIP_TRY(ctx, buf.alloc(&dv[1], n_interior));
        d_rank = d_arrived;
        hipLaunchKernelGGL(k_lbvh_depth, g_int, b256, 0, st, n_interior, tp.parent_interior, dk[0], dv[0]);
        const uint32_t nsb = (n_interior + kSortTile - 1) / kSortTile;
        hipLaunchKernelGGL(k_radix_hist, dim3(nsb), dim3(kSortBlock), 0, st, dk[0], n_interior, 0u, d_hist, nsb);
        hipLaunchKernelGGL(k_radix_row_scan, dim3(256), b256, 0, st, d_hist, nsb, d_digit_base);
        hipLaunchKernelGGL(k_radix_digit_base, dim3(1), b256, 0, st, d_digit_base);
        hipLaunchKernelGGL(k_radix_scatter, dim3(nsb), dim3(kSortBlock), 0, st, dk[0], dv[0], dk[1], dv[1], n_interior, 0u, d_hist, d_digit_base, nsb);
        hipLaunchKernelGGL(k_lbvh_rank, g_int, b256, 0, st, n_interior, dv[1], d_rank);
    }
    hipLaunchKernelGGL(k_lbvh_emit, g_int, b256, 0, st, d_leaves, d_vals[cur], n, tp, d_boxes, d_rank, ctx->d_blob + sc.off_nodes, ctx->d_bvh_ref);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(e1, st));

    uint32_t height = 0;
    float root_box[6];
    HIP_TRY(ctx, hipMemcpyAsync(&height, d_height, 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(root_box, d_boxes, sizeof root_box, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (height > TRC_MAX_BVH_DEPTH) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "lbvh: tree deeper than TRC_MAX_BVH_DEPTH");

    plan_lds(sc, height, true);       // the fat nodes are numbered by depth: any prefix may be staged
    sc.blob = ctx->d_blob;
    KScene ks{};
    ks.sc = sc;
    for (int a = 0; a < 6; ++a) ks.root_box[a] = root_box[a];
    ctx->ks = ks;
    ctx->lds_scene = sc.n_lds_nodes == sc.n_nodes;
    ctx->lds_prefix_ok = true;
    ctx->n_bvh_ref = n_nodes;
    ctx->lbvh_height = height;
    ctx->lbvh_build_ms = ms;
    ctx->has_scene = true;
    ctx->cost_valid = false; ctx->d_last_order = nullptr;      // another scene: the recorded block costs say nothing about it
    return TRC_OK;
}

extern "C" {

trc_status trc_upload_scene_lbvh(trc_ctx* ctx, const trc_scene* s) { return upload_device_tree(ctx, s, false, false); }
trc_status trc_upload_scene_sah(trc_ctx* ctx, const trc_scene* s) { return upload_device_tree(ctx, s, true, false); }
trc_status trc_upload_scene_device(trc_ctx* ctx, const trc_scene* s, uint32_t flags) {
    if (flags & ~(uint32_t)(TRC_TREE_SAH | TRC_TREE_TRIANGLE_LEAVES)) return ctx ? trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_upload_scene_device: unknown flag") : TRC_ERR_INVALID_ARG;
    return upload_device_tree(ctx, s, (flags & TRC_TREE_SAH) != 0, (flags & TRC_TREE_TRIANGLE_LEAVES) != 0);
}

trc_status trc_download_bvh(trc_ctx* ctx, trc_BVH* out, uint32_t capacity, uint32_t* n_nodes) {
    { const trc_status fs_ = trc_flush(ctx); if (fs_ != TRC_OK) return fs_; }      // a kept launch of few samples goes first (trc_render)
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (!ctx->d_bvh_ref || ctx->n_bvh_ref == 0) return trc_fail(ctx, TRC_ERR_NO_SCENE, "trc_download_bvh: no device-built tree (trc_upload_scene_lbvh)");
    if (n_nodes) *n_nodes = ctx->n_bvh_ref;
    if (!out) return TRC_OK;
    if (capacity < ctx->n_bvh_ref) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "trc_download_bvh: capacity too small");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    { const trc_status cs = trc_copy_to_host(ctx, out, ctx->d_bvh_ref, sizeof(trc_BVH) * ctx->n_bvh_ref, ctx->stream); if (cs != TRC_OK) return cs; }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return TRC_OK;
}

trc_status trc_lbvh_info(trc_ctx* ctx, uint32_t* n_nodes, uint32_t* height, float* device_build_ms) {
    if (!ctx) return TRC_ERR_INVALID_ARG;
    if (ctx->n_bvh_ref == 0) return trc_fail(ctx, TRC_ERR_NO_SCENE, "trc_lbvh_info: no device-built tree");
    if (n_nodes) *n_nodes = ctx->n_bvh_ref;
    if (height) *height = ctx->lbvh_height;
    if (device_build_ms) *device_build_ms = ctx->lbvh_build_ms;
    return TRC_OK;
}

}  // extern "C"

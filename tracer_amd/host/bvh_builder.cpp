// bvh_builder.cpp -- host-side SAH BVH builder (the reference keeps this on the host:
// RT_Metal/Metal/BVH.hh:30-314, host branch).
//
// Behaviour kept from the reference:
//   * leaf records come from buildNode (BVH.hh:273-314): world AABB of the 8 box corners;
//   * split axis = max extent of the CENTROID bounds (BVH.hh:81-89, AABB.hh:42-49);
//   * 10 buckets, cost_i = 1 + (n0*A0 + n1*A1)/A(centroid box), first minimum wins
//     (BVH.hh:91-139); partition by bucket <= minBucket (BVH.hh:141-168);
//   * degenerate partition -> order by centroid on the axis, split at the median (BVH.hh:187-195);
//   * two leaves -> ordered by centroid on the max-extent axis (BVH.hh:60-77);
//   * array layout after buildTree (BVH.hh:246-269): [root, leaf 0..n-1, interiors...],
//     i.e. every original leaf index is shifted by one and the root has parent 0.
//
// Design of THIS builder (not the reference's):
//   * the interior numbering is the serial post-order of the reference's recursion -- the
//     reference's own numbering depends on GCD scheduling (BVH.hh:204-218,233-238), post-order
//     is the schedule a single worker produces.  A subtree with m leaves owns exactly m-1
//     interior slots, so every recursive call knows its slot range up front: no mutex, no
//     append, subtrees build in parallel (std::thread) with a deterministic result.
#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "host_math.hpp"

using namespace trc;

namespace {

constexpr uint32_t kBuckets = 10;

struct Builder {
    trc_BVH* nodes;            // final array; leaves live at [1, n]
    uint32_t n;                // leaf count
    std::vector<uint32_t> idx; // permutation of final leaf slots
    std::vector<trc_float3> centroid;  // per final slot (leaves only, index - 1)
    std::atomic<int> spare_threads;

    const trc_float3& cen(uint32_t slot) const { return centroid[slot - 1]; }

    // returns the final index of the subtree root. `base` = first interior slot owned by it.
    uint32_t build(uint32_t start, uint32_t end, uint32_t base, bool is_root) {
        const uint32_t span = end - start;
        if (span == 1) return idx[start];

        uint32_t dim = 0, left = 0, right = 0, mid = start + 1;
        if (span == 2) {
            const uint32_t a = idx[start], b = idx[start + 1];
            dim = box_max_extent(box_of(cen(a), cen(b)));
            if (get(cen(a), dim) < get(cen(b), dim)) { left = a; right = b; } else { left = b; right = a; }
        } else {
            trc_AABB cbox = empty_box();
            for (uint32_t i = start; i < end; ++i) cbox = box_grow(cbox, cen(idx[i]));
            dim = box_max_extent(cbox);
            const float lo = get(cbox.mini, dim);
            const float extent = get(cbox.maxi, dim) - lo;

            auto bucket_of = [&](uint32_t slot) -> uint32_t {
                // uint(nBuckets * relative(centroid)[dim]), clamped (BVH.hh:99-100)
                float rel = (get(cen(slot), dim) - lo) / extent;
                float fb = kBuckets * rel;
                uint32_t b = (fb >= 0.0f) ? (uint32_t)fb : 0u;
                return std::min(b, kBuckets - 1);
            };

            bool degenerate = !(extent > 0.0f);
            if (!degenerate) {
                uint32_t count[kBuckets] = {0};
                trc_AABB bbox[kBuckets];
                for (auto& b : bbox) b = empty_box();
                for (uint32_t i = start; i < end; ++i) {
                    const uint32_t slot = idx[i], b = bucket_of(slot);
                    bbox[b] = box_union(bbox[b], nodes[slot].bBOX);
                    count[b]++;
                }
                const float inv_area_denominator = box_area(cbox);
                float best = 0.0f; uint32_t best_i = 0;
                for (uint32_t i = 0; i + 1 < kBuckets; ++i) {
                    trc_AABB b0 = empty_box(), b1 = empty_box();
                    int c0 = 0, c1 = 0;
                    for (uint32_t j = 0; j <= i; ++j) { b0 = box_union(b0, bbox[j]); c0 += count[j]; }
                    for (uint32_t j = i + 1; j < kBuckets; ++j) { b1 = box_union(b1, bbox[j]); c1 += count[j]; }
                    // an empty side has area(-inf box); the reference multiplies it by a zero count,
                    // which yields NaN and never wins a "<" comparison -- mirror that outcome.
                    float cost = 1 + (c0 * box_area(b0) + c1 * box_area(b1)) / inv_area_denominator;
                    if (i == 0 || cost < best) { best = cost; best_i = i; }
                }
                auto first = idx.begin() + start, last = idx.begin() + end;
                auto pm = std::partition(first, last, [&](uint32_t slot) { return bucket_of(slot) <= best_i; });
                mid = (uint32_t)(pm - idx.begin());
                degenerate = (mid <= start || mid >= end);
            }
            if (degenerate) {
                std::stable_sort(idx.begin() + start, idx.begin() + end, [&](uint32_t a, uint32_t b) {
                    return get(cen(a), dim) < get(cen(b), dim);
                });
                mid = start + span / 2;
            }

            const uint32_t left_base = base;
            const uint32_t right_base = base + (mid - start - 1);
            bool forked = false;
            if (span > 4096 && spare_threads.fetch_sub(1) > 0) {
                forked = true;
                std::thread t([&] { left = build(start, mid, left_base, false); });
                right = build(mid, end, right_base, false);
                t.join();
                spare_threads.fetch_add(1);
            } else if (span > 4096) {
                spare_threads.fetch_add(1);
            }
            if (!forked) {
                left = build(start, mid, left_base, false);
                right = build(mid, end, right_base, false);
            }
        }

        const uint32_t self = is_root ? 0u : (base + span - 2);
        trc_BVH& node = nodes[self];
        node.parent = 0; node.left = left; node.right = right; node.axis = dim;
        node.pType = TRC_PRIM_BVH; node.pIndex = 0; node._pad[0] = node._pad[1] = 0;
        node.bBOX = box_union(nodes[left].bBOX, nodes[right].bBOX);
        nodes[left].parent = self;
        nodes[right].parent = self;
        return self;
    }
};

}  // namespace

extern "C" {

void trc_host_build_node(const trc_AABB* box, const trc_float4x4* model_matrix,
                         int32_t pType, uint32_t pIndex, trc_BVH* out) {
    const trc_float3 ele[2] = {box->mini, box->maxi};
    trc_float3 lo = f3(FLT_MAX), hi = f3(-FLT_MAX);
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
        trc_float4 corner = mul(*model_matrix, f4(ele[i].x, ele[j].y, ele[k].z, 1.0f));
        lo = f3(std::fmin(lo.x, corner.x), std::fmin(lo.y, corner.y), std::fmin(lo.z, corner.z));
        hi = f3(std::fmax(hi.x, corner.x), std::fmax(hi.y, corner.y), std::fmax(hi.z, corner.z));
    }
    std::memset(out, 0, sizeof(*out));
    out->pType = pType;
    out->pIndex = pIndex;
    out->bBOX.mini = lo;
    out->bBOX.maxi = hi;
}

trc_status trc_host_build_tree(trc_BVH* nodes, uint32_t n_leaves, uint32_t* out_n_nodes) {
    if (!nodes || n_leaves < 2) return TRC_ERR_INVALID_ARG;
    // leaves move from [0, n) to [1, n]; slot 0 becomes the root (BVH.hh:263-268)
    for (uint32_t i = n_leaves; i > 0; --i) nodes[i] = nodes[i - 1];

    Builder b;
    b.nodes = nodes;
    b.n = n_leaves;
    b.idx.resize(n_leaves);
    b.centroid.resize(n_leaves);
    for (uint32_t i = 0; i < n_leaves; ++i) {
        b.idx[i] = i + 1;
        b.centroid[i] = box_centroid(nodes[i + 1].bBOX);
        nodes[i + 1].left = nodes[i + 1].right = 0;
    }
    unsigned hw = std::thread::hardware_concurrency();
    b.spare_threads.store(hw > 1 ? (int)hw - 1 : 0);
    b.build(0, n_leaves, n_leaves + 1, true);
    nodes[0].parent = 0;
    if (out_n_nodes) *out_n_nodes = 2 * n_leaves - 1;
    return TRC_OK;
}

trc_status trc_host_tree_depth(const trc_BVH* nodes, uint32_t n_nodes, uint32_t* out_depth) {
    if (!nodes || n_nodes < 3 || nodes[0].pType != TRC_PRIM_BVH) return TRC_ERR_BVH_INVALID;
    std::vector<uint32_t> stack_node{0}, stack_depth{0};
    uint32_t deepest = 0, visited = 0;
    while (!stack_node.empty()) {
        uint32_t i = stack_node.back(), d = stack_depth.back();
        stack_node.pop_back(); stack_depth.pop_back();
        if (++visited > n_nodes) return TRC_ERR_BVH_INVALID;   // cycle
        if (nodes[i].pType == TRC_PRIM_BVH) {
            uint32_t l = nodes[i].left, r = nodes[i].right;
            if (l == 0 || r == 0 || l >= n_nodes || r >= n_nodes || l == r) return TRC_ERR_BVH_INVALID;
            if (nodes[l].parent != i || nodes[r].parent != i) return TRC_ERR_BVH_INVALID;
            stack_node.push_back(l); stack_depth.push_back(d + 1);
            stack_node.push_back(r); stack_depth.push_back(d + 1);
        } else {
            if (nodes[i].pType < 0 || nodes[i].pType > TRC_PRIM_TRIANGLE) return TRC_ERR_BVH_INVALID;
            deepest = std::max(deepest, d);
        }
    }
    if (visited != n_nodes) return TRC_ERR_BVH_INVALID;
    if (out_depth) *out_depth = deepest;
    return TRC_OK;
}

}  // extern "C"

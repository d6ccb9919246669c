/*
 * oracle_sah.cpp -- CPU restatement of the reference's host SAH build (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Follows RT_Metal/Metal/BVH.hh:35-314 (host branch) and AABB.hh:7-49,213-253 step by step, on the reference's
 * own data structure: ONE growing list whose first n records are the leaves, interiors appended as the recursion
 * unwinds, children stored as "list position + 1" because buildTree moves the root to the front afterwards
 * (BVH.hh:246-269).  The product's builder (tracer_amd/host/bvh_builder.cpp) never appends: it computes every
 * interior's final slot up front so subtrees can build in parallel.  tests/test_bvh_builder.py holds the two
 * against each other record for record; they share no code (this file uses nothing from tracer_amd/host/).
 *
 * One schedule had to be chosen: the reference dispatches the two recursive calls of the top three levels to a
 * concurrent GCD queue (BVH.hh:197-218), so the ORDER in which interiors are appended -- not the tree -- varies
 * from run to run there.  This restatement takes the serial schedule (left call, right call, append), which is
 * what the reference does everywhere below depth 2 (BVH.hh:199-202).
 *
 * The degenerate-partition fallback sorts with std::sort in the reference (BVH.hh:187-195); the order of equal
 * centroids is then the standard library's choice.  Here equal keys keep their order (insertion-stable merge).
 */
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

struct V3 { float c[3]; };

struct Box {                                           // AABB.hh:7-9: an empty box is (+FLT_MAX, -FLT_MAX)
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
};

Box load(const trc_AABB& b) {
    Box r;
    r.lo[0] = b.mini.x; r.lo[1] = b.mini.y; r.lo[2] = b.mini.z;
    r.hi[0] = b.maxi.x; r.hi[1] = b.maxi.y; r.hi[2] = b.maxi.z;
    return r;
}
void store(const Box& b, trc_AABB* out) {
    out->mini.x = b.lo[0]; out->mini.y = b.lo[1]; out->mini.z = b.lo[2];
    out->maxi.x = b.hi[0]; out->maxi.y = b.hi[1]; out->maxi.z = b.hi[2];
}

V3 centroid(const Box& b) {                            // AABB.hh:22-25: mini + (maxi - mini) / 2
    V3 r;
    for (int a = 0; a < 3; ++a) r.c[a] = b.lo[a] + (b.hi[a] - b.lo[a]) / 2;
    return r;
}
float area(const Box& b) {                             // AABB.hh:27-30
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
unsigned widest_axis(const Box& b) {                   // AABB.hh:42-49: x only if strictly widest, else y over z
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    return dy > dz ? 1 : 2;
}
Box merged(const Box& a, const Box& b) {               // AABB.hh:227-239
    Box r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return r;
}
Box grown(const Box& a, const V3& p) {                 // AABB.hh:241-253
    Box r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], p.c[k]); r.hi[k] = fmaxf(a.hi[k], p.c[k]); }
    return r;
}

constexpr unsigned kBuckets = 10;                      // BVH.hh:93

struct Sah {
    std::vector<trc_BVH> list;                         // the reference's bvh_list
    std::vector<uint32_t> order;                       // the reference's idx_list

    float key(uint32_t i, unsigned axis) const { return centroid(load(list[i].bBOX)).c[axis]; }

    // BVH.hh:99-100,144-149: uint(nBuckets * relative(centroid)[dim]), then min(b, nBuckets - 1)
    static unsigned bucket(const Box& cbox, const V3& c, unsigned axis) {
        const float rel = (c.c[axis] - cbox.lo[axis]) / (cbox.hi[axis] - cbox.lo[axis]);   // AABB.hh:37-40
        const float scaled = kBuckets * rel;
        const unsigned b = scaled >= 0.0f ? (unsigned)scaled : 0u;       // a negative / NaN product never occurs on a
        return b < kBuckets - 1 ? b : kBuckets - 1;                      // non-degenerate axis; 0 keeps the cast defined
    }

    void sort_by_centroid(uint32_t first, uint32_t last, unsigned axis) {
        // bottom-up merge sort on (key, position): equal keys keep their order
        const uint32_t n = last - first;
        std::vector<uint32_t> a(order.begin() + first, order.begin() + last), b(n);
        std::vector<float> k(n), kb(n);
        for (uint32_t i = 0; i < n; ++i) k[i] = key(a[i], axis);
        for (uint32_t width = 1; width < n; width *= 2) {
            for (uint32_t lo = 0; lo < n; lo += 2 * width) {
                const uint32_t mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
                uint32_t i = lo, j = mid, o = lo;
                while (i < mid && j < hi) {
                    if (k[j] < k[i]) { b[o] = a[j]; kb[o++] = k[j++]; } else { b[o] = a[i]; kb[o++] = k[i++]; }
                }
                while (i < mid) { b[o] = a[i]; kb[o++] = k[i++]; }
                while (j < hi) { b[o] = a[j]; kb[o++] = k[j++]; }
            }
            a.swap(b); k.swap(kb);
        }
        for (uint32_t i = 0; i < n; ++i) order[first + i] = a[i];
    }

    // returns the subtree root's position in `list` (BVH.hh:35-244)
    uint32_t make(uint32_t first, uint32_t last) {
        const uint32_t span = last - first;
        if (span == 1) return order[first];                                           // BVH.hh:51-53

        unsigned axis = 0;
        uint32_t left = 0, right = 0;
        if (span == 2) {                                                              // BVH.hh:59-77
            const uint32_t ia = order[first], ib = order[first + 1];
            Box pair;                                                                 // make(centroid_a, centroid_b)
            const V3 ca = centroid(load(list[ia].bBOX)), cb = centroid(load(list[ib].bBOX));
            for (int k = 0; k < 3; ++k) { pair.lo[k] = fminf(ca.c[k], cb.c[k]); pair.hi[k] = fmaxf(ca.c[k], cb.c[k]); }
            axis = widest_axis(pair);
            if (ca.c[axis] < cb.c[axis]) { left = ia; right = ib; } else { left = ib; right = ia; }
        } else {
            Box cbox;                                                                 // BVH.hh:81-89
            for (uint32_t i = first; i < last; ++i) cbox = grown(cbox, centroid(load(list[order[i]].bBOX)));
            axis = widest_axis(cbox);

            unsigned count[kBuckets] = {};                                            // BVH.hh:93-110
            Box bbox[kBuckets];
            for (uint32_t i = first; i < last; ++i) {
                const Box prim = load(list[order[i]].bBOX);
                const unsigned b = bucket(cbox, centroid(prim), axis);
                bbox[b] = merged(bbox[b], prim);
                count[b]++;
            }
            float cost[kBuckets - 1];                                                 // BVH.hh:112-131
            for (unsigned i = 0; i + 1 < kBuckets; ++i) {
                Box b0, b1; int n0 = 0, n1 = 0;
                for (unsigned j = 0; j <= i; ++j) { b0 = merged(b0, bbox[j]); n0 += (int)count[j]; }
                for (unsigned j = i + 1; j < kBuckets; ++j) { b1 = merged(b1, bbox[j]); n1 += (int)count[j]; }
                cost[i] = 1 + (n0 * area(b0) + n1 * area(b1)) / area(cbox);
            }
            float best = cost[0]; int split = 0;                                      // BVH.hh:133-140: first minimum
            for (int i = 1; i < (int)kBuckets - 1; ++i) if (cost[i] < best) { best = cost[i]; split = i; }

            auto below = [&](uint32_t pos) {                                          // BVH.hh:142-152
                return (int)bucket(cbox, centroid(load(list[order[pos]].bBOX)), axis) <= split;
            };
            uint32_t lo = first, hi = last;                                           // BVH.hh:154-168
            bool done = false;
            while (!done && lo != hi) {
                while (below(lo)) { if (++lo == hi) { done = true; break; } }
                if (done) break;
                do { if (--hi == lo) { done = true; break; } } while (!below(hi));
                if (done) break;
                const uint32_t t = order[lo]; order[lo] = order[hi]; order[hi] = t;
                ++lo;
            }
            uint32_t mid = lo;
            if (mid <= first || mid >= last) {                                        // BVH.hh:187-195
                sort_by_centroid(first, last, axis);
                mid = first + span / 2;
            }
            left = make(first, mid);                                                  // BVH.hh:199-202
            right = make(mid, last);
        }

        trc_BVH node;                                                                 // BVH.hh:221-242
        std::memset(&node, 0, sizeof(node));
        node.axis = axis;
        node.left = left + 1;
        node.right = right + 1;
        node.pType = TRC_PRIM_BVH;
        store(merged(load(list[left].bBOX), load(list[right].bBOX)), &node.bBOX);
        const uint32_t parent = (uint32_t)list.size() + 1;
        list.push_back(node);
        list[left].parent = parent;
        list[right].parent = parent;
        return parent - 1;
    }
};

}  // namespace

extern "C" {

/* BVH.hh:273-314 buildNode: the world-space box of a primitive = min / max over its box's eight corners taken
 * through the model matrix.  column-major 4x4 times (x, y, z, 1), summed in column order. */
void orc_sah_leaf(const trc_AABB* box, const trc_float4x4* m, int32_t pType, uint32_t pIndex, trc_BVH* out) {
    const float* col = (const float*)m;                // 4 columns of 4 floats
    const float ex[2] = {box->mini.x, box->maxi.x}, ey[2] = {box->mini.y, box->maxi.y}, ez[2] = {box->mini.z, box->maxi.z};
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k)
        for (int c = 0; c < 3; ++c) {
            const float w = col[0 + c] * ex[i] + col[4 + c] * ey[j] + col[8 + c] * ez[k] + col[12 + c] * 1.0f;
            lo[c] = fminf(lo[c], w);
            hi[c] = fmaxf(hi[c], w);
        }
    std::memset(out, 0, sizeof(*out));
    out->pType = pType;
    out->pIndex = pIndex;
    out->bBOX.mini.x = lo[0]; out->bBOX.mini.y = lo[1]; out->bBOX.mini.z = lo[2];
    out->bBOX.maxi.x = hi[0]; out->bBOX.maxi.y = hi[1]; out->bBOX.maxi.z = hi[2];
}

/* BVH.hh:246-269 buildTree: n leaf records in, 2n-1 records out, root first. */
void orc_sah_build(const trc_BVH* leaves, uint32_t n, trc_BVH* out) {
    Sah s;
    s.list.assign(leaves, leaves + n);
    s.list.reserve(2 * (size_t)n - 1);
    s.order.resize(n);
    for (uint32_t i = 0; i < n; ++i) { s.order[i] = i; s.list[i].left = s.list[i].right = 0; s.list[i].parent = 0; }
    s.make(0, n);
    trc_BVH root = s.list.back();                      // the root was appended last; it goes to the front
    root.parent = 0;
    s.list.pop_back();
    out[0] = root;
    std::memcpy(out + 1, s.list.data(), sizeof(trc_BVH) * s.list.size());
    out[root.left].parent = 0;
    out[root.right].parent = 0;
}

}  // extern "C"

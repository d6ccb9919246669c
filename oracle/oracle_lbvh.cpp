/*
 * oracle_lbvh.cpp -- CPU statement of the LBVH build (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Arithmetic contract with tracer_amd/csrc/trc_lbvh.hip: binary32 add / sub / mul / div / min / max only, no
 * contraction, fixed order; everything after the Morton code is integer work, so device and oracle agree bit
 * for bit on all 2n-1 records.
 */
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline uint32_t expand_bits10(uint32_t v) {      // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
inline uint32_t quantise(float c, float lo, float ext) {
    if (!(ext > 0.0f)) return 0;
    float v = ((c - lo) / ext) * 1024.0f;
    if (!(v >= 0.0f)) v = 0.0f;                  // also NaN
    if (v > 1023.0f) v = 1023.0f;
    return (uint32_t)v;
}
inline int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }

}  // namespace

extern "C" void orc_lbvh_build(const trc_BVH* leaves, uint32_t n, trc_BVH* out, uint32_t* out_height) {
    // 1-2. centroids and their bounds
    std::vector<float> cx(n), cy(n), cz(n);
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()};
    float hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (uint32_t k = 0; k < n; ++k) {
        const trc_AABB& b = leaves[k].bBOX;
        cx[k] = (b.mini.x + b.maxi.x) * 0.5f; cy[k] = (b.mini.y + b.maxi.y) * 0.5f; cz[k] = (b.mini.z + b.maxi.z) * 0.5f;
        const float c[3] = {cx[k], cy[k], cz[k]};
        for (int a = 0; a < 3; ++a) { if (c[a] < lo[a]) lo[a] = c[a]; if (c[a] > hi[a]) hi[a] = c[a]; }
    }
    const float ext[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
    // 3-5. keys, sorted
    std::vector<uint64_t> key(n);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t m = (expand_bits10(quantise(cx[k], lo[0], ext[0])) << 2) | (expand_bits10(quantise(cy[k], lo[1], ext[1])) << 1) |
                           expand_bits10(quantise(cz[k], lo[2], ext[2]));
        key[k] = ((uint64_t)m << 32) | k;
    }
    std::sort(key.begin(), key.end());
    auto delta = [&](int64_t i, int64_t j) -> int { return (j < 0 || j >= (int64_t)n) ? -1 : clz64(key[i] ^ key[j]); };
    auto leaf_slot = [&](uint32_t pos) { return (uint32_t)(key[pos] & 0xFFFFFFFFu) + 1u; };
    auto interior_slot = [&](uint32_t i) { return i == 0 ? 0u : n + i; };

    const uint32_t n_nodes = 2 * n - 1;
    std::memset(out, 0, sizeof(trc_BVH) * n_nodes);
    for (uint32_t k = 0; k < n; ++k) {
        out[k + 1] = leaves[k];
        out[k + 1].parent = 0; out[k + 1].left = 0; out[k + 1].right = 0;
    }
    // 6. Karras 2012, fig. 4
    for (int64_t i = 0; i + 1 < (int64_t)n; ++i) {
        const int d = (delta(i, i + 1) - delta(i, i - 1)) < 0 ? -1 : 1;
        const int dmin = delta(i, i - d);
        int64_t lmax = 2;
        while (delta(i, i + lmax * d) > dmin) lmax *= 2;
        int64_t l = 0;
        for (int64_t t = lmax / 2; t >= 1; t /= 2)
            if (delta(i, i + (l + t) * d) > dmin) l += t;
        const int64_t j = i + l * d;
        const int dnode = delta(i, j);
        int64_t s = 0;
        for (int64_t t = (l + 1) / 2;; t = (t + 1) / 2) {
            if (delta(i, i + (s + t) * d) > dnode) s += t;
            if (t <= 1) break;
        }
        const int64_t gamma = i + s * d + (d < 0 ? -1 : 0);
        const int64_t first = i < j ? i : j, last = i < j ? j : i;
        const uint32_t self = interior_slot((uint32_t)i);
        const uint32_t left = (first == gamma) ? leaf_slot((uint32_t)gamma) : interior_slot((uint32_t)gamma);
        const uint32_t right = (last == gamma + 1) ? leaf_slot((uint32_t)gamma + 1) : interior_slot((uint32_t)gamma + 1);
        trc_BVH& nd = out[self];
        nd.left = left; nd.right = right;
        nd.pType = TRC_PRIM_BVH; nd.pIndex = 0;
        const int mbit = dnode - 2;              // keys are < 2^62: bit 0 of the Morton field is key bit 61
        nd.axis = (mbit >= 0 && mbit < 30) ? (uint32_t)(mbit % 3) : 0u;
        out[left].parent = self;
        out[right].parent = self;
    }
    out[0].parent = 0;
    // 8-9. boxes bottom-up and leaf depths top-down (explicit stack; order does not matter for min/max)
    std::vector<uint32_t> order; order.reserve(n);
    std::vector<uint32_t> depth(n_nodes, 0);
    order.push_back(0);
    uint32_t height = 0;
    for (size_t h = 0; h < order.size(); ++h) {
        const uint32_t i = order[h];
        for (uint32_t c : {out[i].left, out[i].right}) {
            depth[c] = depth[i] + 1;
            if (out[c].pType == TRC_PRIM_BVH) order.push_back(c); else height = std::max(height, depth[c]);
        }
    }
    for (size_t h = order.size(); h-- > 0;) {
        trc_BVH& nd = out[order[h]];
        const trc_AABB &L = out[nd.left].bBOX, &R = out[nd.right].bBOX;
        nd.bBOX.mini.x = std::min(L.mini.x, R.mini.x); nd.bBOX.mini.y = std::min(L.mini.y, R.mini.y); nd.bBOX.mini.z = std::min(L.mini.z, R.mini.z);
        nd.bBOX.maxi.x = std::max(L.maxi.x, R.maxi.x); nd.bBOX.maxi.y = std::max(L.maxi.y, R.maxi.y); nd.bBOX.maxi.z = std::max(L.maxi.z, R.maxi.z);
    }
    // 10. two sweeps of tree rotations (trc_lbvh.hip k_lbvh_rotate_pass): bottom-up, every interior node may swap a child
    // with a grandchild on the other side when the rebuilt child's surface area shrinks (best of <= 4, fixed candidate
    // order, strict improvement).  Reverse BFS order visits every node after all nodes below it.
    auto area = [](const trc_AABB& b) {
        const float dx = b.maxi.x - b.mini.x, dy = b.maxi.y - b.mini.y, dz = b.maxi.z - b.mini.z;
        return 2.0f * ((dx * dy + dy * dz) + dz * dx);
    };
    auto merged = [](const trc_AABB& a, const trc_AABB& b) {
        trc_AABB m = a;
        m.mini.x = std::min(a.mini.x, b.mini.x); m.mini.y = std::min(a.mini.y, b.mini.y); m.mini.z = std::min(a.mini.z, b.mini.z);
        m.maxi.x = std::max(a.maxi.x, b.maxi.x); m.maxi.y = std::max(a.maxi.y, b.maxi.y); m.maxi.z = std::max(a.maxi.z, b.maxi.z);
        return m;
    };
    for (int sweep = 0; sweep < 2; ++sweep) {
        for (size_t h = order.size(); h-- > 0;) {
            const uint32_t i = order[h];
            const uint32_t L = out[i].left, R = out[i].right;
            float best = 0.0f;
            int which = -1;
            if (out[R].pType == TRC_PRIM_BVH) {
                const float old = area(out[R].bBOX);
                const float a0 = area(merged(out[L].bBOX, out[out[R].right].bBOX));     // L <-> RL
                const float a1 = area(merged(out[out[R].left].bBOX, out[L].bBOX));      // L <-> RR
                if (old - a0 > best) { best = old - a0; which = 0; }
                if (old - a1 > best) { best = old - a1; which = 1; }
            }
            if (out[L].pType == TRC_PRIM_BVH) {
                const float old = area(out[L].bBOX);
                const float a2 = area(merged(out[R].bBOX, out[out[L].right].bBOX));     // R <-> LL
                const float a3 = area(merged(out[out[L].left].bBOX, out[R].bBOX));      // R <-> LR
                if (old - a2 > best) { best = old - a2; which = 2; }
                if (old - a3 > best) { best = old - a3; which = 3; }
            }
            if (which < 0) continue;
            const bool right_side = which < 2, first = (which == 0 || which == 2);
            const uint32_t X = right_side ? R : L, down = right_side ? L : R;
            const uint32_t up = first ? out[X].left : out[X].right, keep = first ? out[X].right : out[X].left;
            if (right_side) out[i].left = up; else out[i].right = up;
            out[up].parent = i;
            if (first) out[X].left = down; else out[X].right = down;
            out[down].parent = X;
            out[X].bBOX = merged(out[down].bBOX, out[keep].bBOX);
        }
        // the tree changed: BFS order and height again, and every box again from its children as they are now (left, right) --
        // what the device's refit after a sweep does; the values are the ones the rotations left, but which of a -0 and a +0
        // a box keeps depends on the order of its operands
        order.clear(); order.push_back(0);
        height = 0;
        std::fill(depth.begin(), depth.end(), 0u);
        for (size_t h = 0; h < order.size(); ++h) {
            const uint32_t i = order[h];
            for (uint32_t c : {out[i].left, out[i].right}) {
                depth[c] = depth[i] + 1;
                if (out[c].pType == TRC_PRIM_BVH) order.push_back(c); else height = std::max(height, depth[c]);
            }
        }
        for (size_t h = order.size(); h-- > 0;) {
            trc_BVH& nd = out[order[h]];
            nd.bBOX = merged(out[nd.left].bBOX, out[nd.right].bBOX);
        }
    }
    if (out_height) *out_height = height;
}

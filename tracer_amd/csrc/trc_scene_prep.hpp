// trc_scene_prep.hpp -- host-side preparation of the device scene blob (layout in dev_scene.hpp), shared by the
// host-tree upload (trc_upload_scene, trc_abi.hip) and the on-device LBVH upload (trc_lbvh.hip).
#pragma once

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trc_ctx.hpp"

inline uint32_t f2u(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// fn(begin, end) over [0, n) on up to 16 host threads (the per-triangle loops of a 1 M-triangle upload)
template <class Fn>
inline void prep_parallel_for(size_t n, Fn fn) {
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = n < (1u << 16) ? 1 : std::min<size_t>(std::min<size_t>(hw, 16), n >> 15);
    if (nt <= 1) { fn((size_t)0, n); return; }
    std::vector<std::thread> th;
    const size_t chunk = (n + nt - 1) / nt;
    for (size_t k = 0; k < nt; ++k) {
        const size_t b = k * chunk, e = std::min(n, b + chunk);
        if (b < e) th.emplace_back([=] { fn(b, e); });
    }
    for (auto& t : th) t.join();
}

// checks shared by the host-tree and the device-LBVH upload paths: primitive arrays, indices, materials
inline trc_status validate_primitives(trc_ctx* ctx, const trc_scene* s) {
    if (!s->materials || s->n_material == 0) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "scene: no materials");
    if (s->n_index % 3) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "scene: n_index not a multiple of 3");
    const uint32_t n_tri = s->n_index / 3;
    std::atomic<bool> bad_index{false};
    prep_parallel_for(s->n_index, [&](size_t b, size_t e) {
        for (size_t t = b; t < e; ++t) if (s->idxList[t] >= s->n_vertex) { bad_index = true; return; }
    });
    if (bad_index) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "scene: triangle index out of range");
    auto bad_mat = [&](uint32_t m) { return m >= s->n_material; };
    for (uint32_t i = 0; i < s->n_sphere; ++i) if (bad_mat(s->sphereList[i].material)) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "sphere material out of range");
    for (uint32_t i = 0; i < s->n_square; ++i) if (bad_mat(s->squareList[i].material)) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "square material out of range");
    for (uint32_t i = 0; i < s->n_cube; ++i) if (bad_mat(s->cubeList[i].material)) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "cube material out of range");
    if (n_tri && s->n_material <= 19) return trc_fail(ctx, TRC_ERR_INVALID_ARG, "triangles use material 19 (Triangle.hh:82): need >= 20 materials");
    for (uint32_t i = 0; i < s->n_square; ++i)
        if (s->squareList[i].axis_i > 2 || s->squareList[i].axis_j > 2 || s->squareList[i].axis_k > 2)
            return trc_fail(ctx, TRC_ERR_INVALID_ARG, "square axis out of range");
    return TRC_OK;
}
inline trc_status validate_leaf(trc_ctx* ctx, const trc_scene* s, const trc_BVH& leaf) {
    const int32_t t = leaf.pType;
    const uint32_t pi = leaf.pIndex, n_tri = s->n_index / 3;
    const uint32_t limit = t == TRC_PRIM_SPHERE ? s->n_sphere : t == TRC_PRIM_SQUARE ? s->n_square
                         : t == TRC_PRIM_CUBE ? s->n_cube : t == TRC_PRIM_TRIANGLE ? n_tri : 0;
    if (t < 0 || t > TRC_PRIM_TRIANGLE || pi >= limit) return trc_fail(ctx, TRC_ERR_BVH_INVALID, "bvh: leaf with bad primitive type/index");
    if (pi > kTagIndexMask) return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "bvh: primitive index exceeds 29 bits");
    return TRC_OK;
}

#ifndef TRC_DESCEND_MIN_GLOBAL
#define TRC_DESCEND_MIN_GLOBAL 12     // 8 until the loop lost its done flag; with the leaner step: 8 / 12 / 16 / 24 / 32 = 23.54 / 23.22 / 23.32 / 23.63 / 24.07 ms (config 4), 40.85 / 40.49 / 40.56 / 40.90 / 41.19 (config 3)
#endif
#ifndef TRC_DESCEND_MIN_BEYOND_CACHE
#define TRC_DESCEND_MIN_BEYOND_CACHE 6
#endif
// blob offsets for `n_interior` fat nodes (see dev_scene.hpp)
inline trc_status layout_scene(trc_ctx* ctx, const trc_scene* s, uint32_t n_interior, DScene& sc, uint64_t& total_dwords) {
    const uint32_t n_tri = s->n_index / 3;
    auto align4 = [](uint32_t v) { return (v + 3u) & ~3u; };
    sc = DScene{};
    sc.off_spheres = 0;
    sc.off_squares = align4(sc.off_spheres + s->n_sphere * kSphereDwords);
    sc.off_cubes = align4(sc.off_squares + s->n_square * kSquareDwords);
    sc.off_materials = align4(sc.off_cubes + s->n_cube * kCubeDwords);
    sc.off_nodes = align4(sc.off_materials + s->n_material * kMaterialDwords);
    if ((uint64_t)sc.off_nodes * 4 + kNodeDwords * 4 > kLdsSceneBytes)
        return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "analytic primitives + materials exceed the LDS staging budget");
    sc.off_tripos = align4(sc.off_nodes + n_interior * kNodeDwords);
    total_dwords = (uint64_t)sc.off_tripos + (uint64_t)n_tri * kTriPosDwords + (uint64_t)n_tri * kTriAttrDwords;
    if (total_dwords > 0xFFFFFFF0ull) return trc_fail(ctx, TRC_ERR_UNSUPPORTED, "scene too large for 32-bit dword offsets");
    sc.off_triattr = sc.off_tripos + n_tri * kTriPosDwords;
    sc.n_nodes = n_interior; sc.n_spheres = s->n_sphere; sc.n_squares = s->n_square; sc.n_cubes = s->n_cube;
    sc.n_materials = s->n_material; sc.n_triangles = n_tri;
    // Descent threshold of the production walk (trav_iter): a scene beyond the 256 MiB Infinity Cache pays HBM latency for most
    // box steps, and a wavefront whose few remaining descenders keep everybody else waiting pays it once per step -- leaving the
    // descent earlier wins there (teapot x 256, 708 MB: 19.3 / 18.3 / 17.9 / 17.9 ms per launch at 12 / 8 / 6 / 4), while the scenes that
    // fit prefer 12 (config 4: 22.83 / 22.88 / 23.08 / 24.02; config 3: 40.06 / 40.29 / 40.72 / 41.51; profiles/r05/ab_descend*.txt).
    sc.descend_min = total_dwords * 4ull > (256ull << 20) ? TRC_DESCEND_MIN_BEYOND_CACHE : TRC_DESCEND_MIN_GLOBAL;
    return TRC_OK;
}

// Upload-time LDS plan (k_trace, the SPPM passes, the instrumented kernels; production render launches refine it per
// kernel: trc_abi.hip::plan_launch_lds).  LDS per workgroup = staged prefix + traversal stack; keep 16 one-wavefront
// workgroups per CU resident (measured on MI355X: occupancy beats top-of-tree staging -- 8/16/24/40/64 KB staged on the
// 1 M-triangle scene gave 1826/1553/1155/1165/666 Mrays/s), so nodes are staged only into what the stack leaves of 9.5 KB.
// `prefix_ok`: the first fat nodes are the top of the tree (BFS order); otherwise all or nothing.
inline void plan_lds(DScene& sc, uint32_t max_leaf_depth, bool prefix_ok) {
    const uint32_t stack_dwords = std::max(1u, max_leaf_depth) * kBlock;
    const uint32_t per_block = 38u * 1024u / 4u * kBlock / 256u;      // 16 one-wavefront workgroups per CU
    uint32_t budget_dwords = (per_block > stack_dwords) ? per_block - stack_dwords : 0u;
    budget_dwords = std::min(budget_dwords, kLdsSceneBytes / 4);
    budget_dwords = std::max(budget_dwords, sc.off_nodes + kNodeDwords);
    sc.n_lds_nodes = std::min<uint32_t>(sc.n_nodes, (budget_dwords - sc.off_nodes) / kNodeDwords);
    if (!prefix_ok && sc.n_lds_nodes < sc.n_nodes) sc.n_lds_nodes = 0;
    sc.lds_dwords = sc.off_nodes + sc.n_lds_nodes * kNodeDwords;
    sc.stack_depth = std::max(1u, max_leaf_depth);
    sc.stack_lds = sc.stack_depth;
    sc.stack_ovf_rows = 0;
}

// analytic primitives and materials into the blob (the fat nodes are written by the caller, the triangle records on the
// device: trc_repack_triangles)
inline void fill_primitives(const trc_scene* s, const DScene& sc, uint32_t* blob) {
    for (uint32_t i = 0; i < s->n_sphere; ++i) {
        const trc_Sphere& sp = s->sphereList[i];
        uint32_t* q = &blob[sc.off_spheres + (size_t)i * kSphereDwords];
        q[0] = f2u(sp.center.x); q[1] = f2u(sp.center.y); q[2] = f2u(sp.center.z); q[3] = f2u(sp.radius);
        q[4] = sp.material;
    }
    for (uint32_t i = 0; i < s->n_square; ++i) {
        const trc_Square& sq = s->squareList[i];
        uint32_t* q = &blob[sc.off_squares + (size_t)i * kSquareDwords];
        q[0] = f2u(sq.range_i.x); q[1] = f2u(sq.range_i.y); q[2] = f2u(sq.range_j.x); q[3] = f2u(sq.range_j.y);
        // Square::area() = 2*i*j and aeraPDF() = 1/area (Square.hh:31-38), evaluated once here in binary32
        const float di = sq.range_i.y - sq.range_i.x, dj = sq.range_j.y - sq.range_j.x;
        const float area = 2 * di * dj;
        const float pdf = 1 / area;
        q[4] = f2u(sq.value_k); q[5] = f2u(pdf);
        q[6] = (uint32_t)sq.axis_i | ((uint32_t)sq.axis_j << 2) | ((uint32_t)sq.axis_k << 4);
        q[7] = sq.material;
    }
    for (uint32_t i = 0; i < s->n_cube; ++i) {
        const trc_Cube& cb = s->cubeList[i];
        uint32_t* q = &blob[sc.off_cubes + (size_t)i * kCubeDwords];
        auto put_cols = [&](uint32_t* dst, const trc_float4x4& m, int ncols) {
            for (int c = 0; c < ncols; ++c) { dst[3 * c] = f2u(m.columns[c].x); dst[3 * c + 1] = f2u(m.columns[c].y); dst[3 * c + 2] = f2u(m.columns[c].z); }
        };
        put_cols(q, cb.inverse_matrix, 4);
        put_cols(q + 12, cb.model_matrix, 4);
        put_cols(q + 24, cb.normal_matrix, 3);
        q[33] = f2u(cb.box.mini.x); q[34] = f2u(cb.box.mini.y); q[35] = f2u(cb.box.mini.z);
        q[36] = f2u(cb.box.maxi.x); q[37] = f2u(cb.box.maxi.y); q[38] = f2u(cb.box.maxi.z);
        q[39] = cb.material;
    }
    for (uint32_t i = 0; i < s->n_material; ++i) {
        const trc_Material& m = s->materials[i];
        uint32_t* q = &blob[sc.off_materials + (size_t)i * kMaterialDwords];
        q[0] = (uint32_t)m.type; q[1] = (uint32_t)m.textureInfo.type;
        q[2] = f2u(m.textureInfo.albedo.x); q[3] = f2u(m.textureInfo.albedo.y); q[4] = f2u(m.textureInfo.albedo.z);
        q[5] = m.specular ? 1u : 0u;
        q[6] = (uint32_t)m.medium;
    }
}


// Triangle records of the blob (dev_scene.hpp: positions 48 B, attributes 64 B per triangle) built ON THE DEVICE from the
// caller's vertex and index arrays: 32 B per vertex + 12 B per triangle cross PCIe instead of 112 B per triangle of
// host-staged records, and the gather runs at HBM speed (1 M triangles: 29 MB up + 0.1 ms, against 112 MB staged by the
// host).  Queued on the context stream; the temporary copies are freed after the stream has drained.
// d_tri_leaves != nullptr: also one leaf record per triangle (BVH::buildNode under the identity matrix), for the device-built trees
trc_status trc_repack_triangles(trc_ctx* ctx, const trc_scene* s, const DScene& sc, uint32_t* d_blob, trc_BVH* d_tri_leaves);

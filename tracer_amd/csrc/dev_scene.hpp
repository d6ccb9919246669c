// dev_scene.hpp -- HBM/LDS data layout of the scene on the device.
//
// The ABI takes the reference's arrays unchanged (include/tracer_abi.h).  trc_upload_scene repacks
// them ONCE into the layout below; hit results stay bit-identical because every per-ray
// operation of Scene::hit (RT_Metal/Metal/Render.hh:135-252) is preserved, only the bytes that
// one traversal step needs are co-located.
//
//   fat node (64 B, one per INTERIOR node, BFS order, root = 0):
//       float4 q0 = L.min.xyz, L.max.x     the reference reads 16 B header + 2 x 32 B child boxes
//       float4 q1 = L.max.yz,  R.min.xy    + 8 B child tag from three different 64 B records
//       float4 q2 = R.min.z,   R.max.xyz   per descend step (Render.hh:151-160,211-213);
//       float4 q3 = 0, 0, tagL, tagR       here it is ONE 64 B record = 4 x ds_read_b128.
//     tag = type << 29 | index : type 4 = interior (index = fat-node id), else the primitive
//     type (Sphere0 Square1 Cube2 Triangle3) with index = pIndex.  Leaves need no record at all.
//   sphere   32 B : center.xyz, radius | material
//   square   32 B : range_i.xy, range_j.xy | value_k, 1/area (hitRecord.PDF), axes i|j<<2|k<<4, material
//   cube    160 B : inverse c0..c3 (12), model c0..c3 (12), normal c0..c2 (9), box min/max (6), material
//   material 32 B : type, texture type, albedo.rgb
//   triangle positions 48 B : v0, v1, v2 as float4 (what a TEST reads)
//   triangle attributes 64 B: n0 n1 n2 (9), uv0 uv1 uv2 (6)  (read only on an accepted hit)
//
// Blob order: spheres, squares, cubes, materials, fat nodes (BFS), triangle positions, triangle
// attributes.  Every workgroup stages the PREFIX that fits kLdsSceneBytes into LDS: always the
// analytic primitives + materials, then as many fat nodes as fit -- the whole tree for the Cornell
// scenes (traversal never touches L2/HBM), the top levels of the tree (BFS prefix, ~550 nodes) for
// mesh scenes, whose deeper nodes and triangles come from global memory (L2 / Infinity Cache).
#pragma once

#include <stdint.h>

namespace trcdev {

constexpr uint32_t kNodeDwords = 16;
constexpr uint32_t kSphereDwords = 8;
constexpr uint32_t kSquareDwords = 8;
constexpr uint32_t kCubeDwords = 40;
constexpr uint32_t kMaterialDwords = 8;
constexpr uint32_t kTriPosDwords = 12;
constexpr uint32_t kTriAttrDwords = 16;

constexpr uint32_t kTagInterior = 4u;
constexpr uint32_t kTagIndexBits = 29u;
constexpr uint32_t kTagIndexMask = (1u << kTagIndexBits) - 1u;
constexpr uint32_t kTagNone = 0xFFFFFFFFu;

constexpr uint32_t kLdsSceneBytes = 40 * 1024;   // staged-scene budget per workgroup

// Offsets are in dwords from `blob`.
struct DScene {
    const uint32_t* blob;
    uint32_t off_nodes, off_spheres, off_squares, off_cubes, off_materials, off_tripos, off_triattr;
    uint32_t lds_dwords;      // staged prefix: prims + materials + the first n_lds_nodes fat nodes
    uint32_t n_lds_nodes;
    uint32_t n_nodes, n_spheres, n_squares, n_cubes, n_materials, n_triangles;
    uint32_t stack_depth;     // max pending siblings = tree depth (checked <= TRC_MAX_BVH_DEPTH)
    uint32_t stack_lds;       // stack entries per lane kept in LDS (= stack_depth unless a launch provides overflow rows)
    uint32_t stack_ovf_rows;  // rows of 64 entries per wavefront in global memory behind them (0: the stack is all LDS)
    uint32_t descend_min;     // trees read from memory: lanes that must still be descending for the box-step loop to go on while others
                              // wait with a leaf (dev_intersect.hpp trav_iter); chosen by where the scene lives (trc_scene_prep.hpp)
};

struct DCamera {
    float lookFrom[3], u[3], v[3], vertical[3], horizontal[3], cornerLowLeft[3];
    float lenRadius;
};

struct DFrame {
    uint32_t* rng;        // RGBA32Uint, W*H texels
    float* accum;         // RGBA32F, W*H texels
    uint32_t width, height;
};

// exact work counters (device side, one 64-bit atomic per wave per counter at kernel end)
enum StatSlot {
    kStatPaths = 0, kStatRays, kStatShaded, kStatDescend, kStatReturn,
    kStatLeafSphere, kStatLeafSquare, kStatLeafCube, kStatLeafTriangle,
    kStatHitTriangle, kStatHitCube, kStatCount
};

}  // namespace trcdev

// host_scene.hpp -- the host scene container and the reference's primitive constructors (Tracer.mm:127-172), shared by
// scene.cpp (the hard-coded Cornell scenes) and pbrt_scene.cpp (scenes read from pbrt-v3 files).
#pragma once

#include <cstring>
#include <vector>

#include "host_math.hpp"

struct trc_host_scene {
    std::vector<trc_BVH> bvh;
    std::vector<trc_Sphere> spheres;
    std::vector<trc_Square> squares;
    std::vector<trc_Cube> cubes;
    std::vector<trc_TriangleVertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<trc_Material> materials;
};

namespace trc {

inline trc_Material make_material(int32_t type) {
    trc_Material m;
    std::memset(&m, 0, sizeof(m));
    m.type = type;
    m.medium = TRC_MEDIUM_NIL;
    m.specular = 0;
    m.eta = 0.0f;
    m.roughness = 1.0f;                       // Material.hh:30
    m.textureInfo.type = TRC_TEX_CONSTANT;
    m.textureInfo.albedo = f3(0.0f);
    return m;
}

// Tracer.mm:127-153
inline trc_Square make_square(uint8_t axis_i, float i0, float i1, uint8_t axis_j, float j0, float j1,
                       uint8_t axis_k, float k, uint32_t material) {
    trc_Square r;
    std::memset(&r, 0, sizeof(r));
    r.axis_i = axis_i; r.axis_j = axis_j; r.axis_k = axis_k;
    r.range_i.x = i0; r.range_i.y = i1;
    r.range_j.x = j0; r.range_j.y = j1;
    r.value_k = k;
    const float delta = 1.0f / 512.0f;        // SquarePadding, Square.hh:7-9
    trc_float3 a = f3(0.0f), b = f3(0.0f);
    set(a, axis_i, i0); set(a, axis_j, j0); set(a, axis_k, k - delta);
    set(b, axis_i, i1); set(b, axis_j, j1); set(b, axis_k, k + delta);
    r.boundingBOX = box_of(a, b);
    r.model_matrix = identity4x4();
    r.material = material;
    return r;
}

// Tracer.mm:155-163 + the T*R*S set-up of prepareCubeList
inline trc_Cube make_cube(uint32_t material, trc_float4x4 translate, trc_float4x4 rotate, trc_float4x4 scale) {
    trc_Cube r;
    std::memset(&r, 0, sizeof(r));
    r.box = box_of(f3(0, 0, 0), f3(1, 1, 1));
    r.material = material;
    r.model_matrix = mul(mul(translate, rotate), scale);
    r.inverse_matrix = inverse(r.model_matrix);
    r.normal_matrix = transpose(r.inverse_matrix);
    return r;
}

// Tracer.mm:165-172 -- radius inflated by 1e-4, AABB not (quirk B-13)
inline trc_Sphere make_sphere(float radius, trc_float3 c, uint32_t material) {
    trc_Sphere s;
    std::memset(&s, 0, sizeof(s));
    s.radius = radius + 0.0001f;
    s.center = c;
    s.boundingBOX = box_of(c - f3(radius), c + f3(radius));
    s.model_matrix = identity4x4();
    s.material = material;
    return s;
}

}  // namespace trc

// host_math.hpp -- small float3 / float4x4 helpers over the ABI PODs (host side only).
// Column-major 4x4 like simd_float4x4 (reference: RT_Metal/Tracer/Tracer.mm:3-34).
#pragma once

#include <cfloat>
#include <cmath>
#include <cstring>

#include "tracer_abi.h"

namespace trc {

inline trc_float3 f3(float x, float y, float z) { trc_float3 r; r.x = x; r.y = y; r.z = z; r._pad = 0.0f; return r; }
inline trc_float3 f3(float s) { return f3(s, s, s); }
inline trc_float3 operator+(trc_float3 a, trc_float3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline trc_float3 operator-(trc_float3 a, trc_float3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline trc_float3 operator*(trc_float3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
inline trc_float3 operator*(float s, trc_float3 a) { return a * s; }
inline trc_float3 operator/(trc_float3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
inline trc_float3 operator-(trc_float3 a) { return f3(-a.x, -a.y, -a.z); }
inline float dot(trc_float3 a, trc_float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline trc_float3 cross(trc_float3 a, trc_float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
inline float length(trc_float3 a) { return std::sqrt(dot(a, a)); }
inline trc_float3 normalize(trc_float3 a) { return a / length(a); }
inline float get(const trc_float3& a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
inline void set(trc_float3& a, int i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }

inline trc_float4 f4(float x, float y, float z, float w) { trc_float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

inline trc_float4x4 identity4x4() {
    trc_float4x4 m;
    m.columns[0] = f4(1, 0, 0, 0); m.columns[1] = f4(0, 1, 0, 0);
    m.columns[2] = f4(0, 0, 1, 0); m.columns[3] = f4(0, 0, 0, 1);
    return m;
}
inline float& at(trc_float4x4& m, int col, int row) { return (&m.columns[col].x)[row]; }
inline float at(const trc_float4x4& m, int col, int row) { return (&m.columns[col].x)[row]; }

// M * v  (simd_mul(matrix, vector)): sum of columns scaled by the vector lanes
inline trc_float4 mul(const trc_float4x4& m, trc_float4 v) {
    trc_float4 r;
    for (int row = 0; row < 4; ++row)
        (&r.x)[row] = at(m, 0, row) * v.x + at(m, 1, row) * v.y + at(m, 2, row) * v.z + at(m, 3, row) * v.w;
    return r;
}
inline trc_float4x4 mul(const trc_float4x4& a, const trc_float4x4& b) {
    trc_float4x4 r;
    for (int c = 0; c < 4; ++c) r.columns[c] = mul(a, b.columns[c]);
    return r;
}
inline trc_float4x4 transpose(const trc_float4x4& m) {
    trc_float4x4 r;
    for (int c = 0; c < 4; ++c) for (int row = 0; row < 4; ++row) at(r, c, row) = at(m, row, c);
    return r;
}
// general inverse by Gauss-Jordan in double, rounded once to float
inline trc_float4x4 inverse(const trc_float4x4& m) {
    double a[4][8];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { a[r][c] = at(m, c, r); a[r][c + 4] = (r == c) ? 1.0 : 0.0; }
    for (int i = 0; i < 4; ++i) {
        int piv = i;
        for (int r = i + 1; r < 4; ++r) if (std::fabs(a[r][i]) > std::fabs(a[piv][i])) piv = r;
        if (piv != i) for (int c = 0; c < 8; ++c) std::swap(a[i][c], a[piv][c]);
        double d = a[i][i];
        for (int c = 0; c < 8; ++c) a[i][c] /= d;
        for (int r = 0; r < 4; ++r) if (r != i) { double f = a[r][i]; if (f != 0.0) for (int c = 0; c < 8; ++c) a[r][c] -= f * a[i][c]; }
    }
    trc_float4x4 out;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) at(out, c, r) = (float)a[r][c + 4];
    return out;
}

// Tracer.mm:3-34
inline trc_float4x4 scale4x4(float sx, float sy, float sz) {
    trc_float4x4 m = identity4x4(); at(m, 0, 0) = sx; at(m, 1, 1) = sy; at(m, 2, 2) = sz; return m;
}
inline trc_float4x4 translation4x4(float tx, float ty, float tz) {
    trc_float4x4 m = identity4x4(); m.columns[3] = f4(tx, ty, tz, 1); return m;
}
inline trc_float4x4 rotation4x4(float radians, trc_float3 axis) {
    axis = normalize(axis);
    float ct = std::cos(radians), st = std::sin(radians), ci = 1 - ct;
    float x = axis.x, y = axis.y, z = axis.z;
    trc_float4x4 m;
    m.columns[0] = f4(ct + x * x * ci, y * x * ci + z * st, z * x * ci - y * st, 0);
    m.columns[1] = f4(x * y * ci - z * st, ct + y * y * ci, z * y * ci + x * st, 0);
    m.columns[2] = f4(x * z * ci + y * st, y * z * ci - x * st, ct + z * z * ci, 0);
    m.columns[3] = f4(0, 0, 0, 1);
    return m;
}

// AABB helpers (AABB.hh:17-49,212-252 host branch)
inline trc_AABB empty_box() { trc_AABB b; b.mini = f3(FLT_MAX); b.maxi = f3(-FLT_MAX); return b; }
inline trc_AABB box_of(trc_float3 a, trc_float3 b) {
    trc_AABB r;
    r.mini = f3(std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z));
    r.maxi = f3(std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z));
    return r;
}
inline trc_AABB box_union(const trc_AABB& a, const trc_AABB& b) {
    trc_AABB r;
    r.mini = f3(std::fmin(a.mini.x, b.mini.x), std::fmin(a.mini.y, b.mini.y), std::fmin(a.mini.z, b.mini.z));
    r.maxi = f3(std::fmax(a.maxi.x, b.maxi.x), std::fmax(a.maxi.y, b.maxi.y), std::fmax(a.maxi.z, b.maxi.z));
    return r;
}
inline trc_AABB box_grow(const trc_AABB& a, trc_float3 p) {
    trc_AABB r;
    r.mini = f3(std::fmin(a.mini.x, p.x), std::fmin(a.mini.y, p.y), std::fmin(a.mini.z, p.z));
    r.maxi = f3(std::fmax(a.maxi.x, p.x), std::fmax(a.maxi.y, p.y), std::fmax(a.maxi.z, p.z));
    return r;
}
inline trc_float3 box_diagonal(const trc_AABB& b) { return b.maxi - b.mini; }
inline trc_float3 box_centroid(const trc_AABB& b) { return b.mini + box_diagonal(b) / 2.0f; }
inline float box_area(const trc_AABB& b) {
    trc_float3 d = box_diagonal(b);
    return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
}
inline unsigned box_max_extent(const trc_AABB& b) {
    trc_float3 d = box_diagonal(b);
    if (d.x > d.y && d.x > d.z) return 0;
    return d.y > d.z ? 1 : 2;
}

}  // namespace trc

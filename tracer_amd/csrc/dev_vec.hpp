// dev_vec.hpp -- device-side float3 helpers with a FIXED evaluation order.
//
// The arithmetic contract shared with the CPU oracle (oracle/oracle.cpp header comment):
//   dot = x*x' + y*y' + z*z' (left to right), length = sqrt(dot), normalize(v) = v * (1/length),
//   IEEE-754 binary32 everywhere, no FMA contraction (-ffp-contract=off), NaN-ignoring min/max.
#pragma once

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "trc_detmath.h"

#define TRC_DEV __device__ __forceinline__

namespace trcdev {

struct F2 { float x, y; };
struct F3 { float x, y, z; };

TRC_DEV F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
TRC_DEV F3 f3(float s) { return f3(s, s, s); }
TRC_DEV F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
TRC_DEV F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
TRC_DEV F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
TRC_DEV F3 operator/(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
TRC_DEV F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
TRC_DEV F3 operator*(float s, F3 a) { return f3(s * a.x, s * a.y, s * a.z); }
TRC_DEV F3 operator/(F3 a, float s) { return f3(a.x / s, a.y / s, a.z / s); }
TRC_DEV F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
TRC_DEV float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
TRC_DEV F3 cross(F3 a, F3 b) { return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
TRC_DEV float length(F3 a) { return sqrtf(dot(a, a)); }
TRC_DEV F3 normalize(F3 a) { float inv = 1.0f / length(a); return a * inv; }
// per-lane dynamic component access stays in registers (select chains, no scratch)
TRC_DEV float comp(F3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
TRC_DEV void set_comp(F3& a, uint32_t i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
TRC_DEV float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
TRC_DEV float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
TRC_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
TRC_DEV bool is_inf(float x) { return fabsf(x) == __builtin_inff(); }
TRC_DEV bool is_nan(float x) { return x != x; }

constexpr float kPi = 3.14159265358979323846f;      // M_PI_F
constexpr float kPi2 = 1.57079632679489661923f;     // M_PI_2_F

}  // namespace trcdev

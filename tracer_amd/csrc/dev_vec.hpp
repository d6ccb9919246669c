// dev_vec.hpp -- device-side float3 helpers with a FIXED evaluation order.
//
// The arithmetic contract shared with the CPU oracle (oracle/oracle.cpp header comment):
//   dot = x*x' + y*y' + z*z' (left to right), length = sqrt(dot), normalize(v) = v * (1/length),
//   IEEE-754 binary32 everywhere, no FMA contraction (-ffp-contract=off), NaN-ignoring min/max.
#pragma once
// a value the compiler must have computed HERE (it may not sink the computation behind a loop that follows): an empty asm that
// claims to read and write the register
#if defined(__HIP_DEVICE_COMPILE__)
#define TRC_PIN(x) asm volatile("" : "+v"(x))
#else
#define TRC_PIN(x) ((void)0)
#endif

#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#define TRC_DEV __device__ __forceinline__

// trc_detmath.h's square roots (asin, acos) go through the kernels' own statement of sqrtf (sqrt_cr below: the same bits)
#if defined(__HIP_DEVICE_COMPILE__)
namespace trcdev { TRC_DEV float sqrt_cr(float x); }
#define DM_SQRTF(x) trcdev::sqrt_cr(x)
#endif
#include "trc_detmath.h"

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

// ---------------------------------------------------------------- 1 / x and sqrt(x), same bits as the compiler's, fewer instructions
// hipcc's correctly rounded `1.0f / x` is 11 instructions and `sqrtf(x)` 16: a core (v_rcp_f32 + six fma; v_sqrt_f32, its two
// neighbours, two fma residuals, two selects) wrapped in operand guards (v_div_scale x 2, v_div_fmas, v_div_fixup; a 2^32
// pre-scale below 2^-96 and its undo, a class test for 0 / inf).  For an operand in [2^-60, 2^60] every guard is the identity:
// nothing is scaled, no special value is patched, so the core ALONE returns the same bits (checked over all 2^32 operands on
// the hardware: trc_unary_test, tests/test_gpu_unary.py).  The guards are replaced by ONE range test per site and wavefront:
// all active lanes in range -> the core; else -> the compiler's sequence for everybody (rare: lengths of degenerate vectors).
// The two sequences make 31 % of the static instructions of tracePath (157 divisions, 79 square roots).
#ifndef TRC_FAST_UNARY
#define TRC_FAST_UNARY 1
#endif
#if defined(__HIP_DEVICE_COMPILE__) && TRC_FAST_UNARY && !defined(TRC_FAST_MATH)
#define TRC_WAVE_GUARDS 1
TRC_DEV bool unary_in_range(float x) {            // 2^-60 <= |x| < 2^60 (exponent field 67 .. 186)
    return fabsf(x) >= 0x1p-60f && fabsf(x) < 0x1p60f;      // two compares with |x| as a source modifier (NaN fails both)
}
TRC_DEV bool wave_all(bool ok) { return __builtin_amdgcn_ballot_w64(!ok) == 0ull; }
// the compiler's sequence for 1.0f / x without v_div_scale / v_div_fmas' scaling / v_div_fixup -- and without its LAST residual
// correction: over every operand of the guarded range the Newton step and ONE correction already give the correctly rounded
// reciprocal (exhaustive test, op 0; TRC_RCP_STEPS=2 keeps both)
TRC_DEV float rcp_core(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    const float q1 = __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
    return q1;
}
// (both corrections are needed on this hardware: over the 2^30 in-range operands v_sqrt_f32 alone is wrong 152 127 120 times,
// one ulp low in all but 58 920 of them -- measured with the test hook)
TRC_DEV float sqrt_core(float x) {                // ... for sqrtf(x) without the pre-scale and the class test
    const float s = __builtin_amdgcn_sqrtf(x);
    const float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    const float rm = __builtin_fmaf(-sm, s, x), rp = __builtin_fmaf(-sp, s, x);
    float r = (0.0f >= rm) ? sm : s;
    r = (0.0f < rp) ? sp : r;
    return r;
}
// The core is computed FIRST and the range test decides afterwards whether everybody redoes it the long way: the test's compare
// is independent of the core's chain, so the branch finds its condition ready instead of stalling a lone wavefront on it
// (a chain-bound share of a frame runs at one wavefront's latency: 5.9 -> 5.x ms on an eighth of config 2).
// TRC_RCP_GUARD_CLASS: the guard reads v_rcp_f32's own result -- "a normal number" is ONE v_cmp_class, and it is false exactly
// for the operands the core cannot serve (0 and denormals: inf; inf: 0; above 2^126: a denormal, flushed; NaN: NaN).  It hangs
// on the core's FIRST instruction only, so the branch still finds its condition long before the chain ends.
TRC_DEV float rcp_cr(float x) {
    const float r0 = __builtin_amdgcn_rcpf(x);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-x, r0, 1.0f), r0, r0);
    float r = __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
    if (__builtin_expect(!wave_all(__builtin_amdgcn_classf(r0, 0x108)), 0)) r = 1.0f / x;
    return r;
}
// sqrt_core is exact for every operand from 2^-60 up, +inf included (exhaustive test; below ~2^-100 the residuals of the two
// candidates underflow): ONE compare, which a NaN fails.  1 / sqrt needs the upper bound as well (1 / sqrt(inf) is not the core's).
TRC_DEV bool sqrt_in_range(float x) {
    return x >= 0x1p-60f;
}
TRC_DEV bool rsqrt_in_range(float x) {
    return x >= 0x1p-60f && x < 0x1p60f;
}
TRC_DEV float sqrt_cr(float x) {
    float r = sqrt_core(x);
    if (__builtin_expect(!wave_all(sqrt_in_range(x)), 0)) r = sqrtf(x);
    return r;
}
// 1 / sqrt(x) as two correctly rounded steps (normalize): in range, the root lies in [2^-30, 2^30] -- in range again
TRC_DEV float rsqrt_cr(float x) {
    float r = rcp_core(sqrt_core(x));
    if (__builtin_expect(!wave_all(rsqrt_in_range(x)), 0)) r = 1.0f / sqrtf(x);
    return r;
}
#else
#define TRC_WAVE_GUARDS 0
TRC_DEV float rcp_cr(float x) { return 1.0f / x; }
TRC_DEV float sqrt_cr(float x) { return sqrtf(x); }
TRC_DEV float rsqrt_cr(float x) { return 1.0f / sqrtf(x); }
#endif
TRC_DEV float rcp1(float x) {                     // the shading code's `1 / x`
    return rcp_cr(x);
}
// x / c for a divisor known when the code is written (pi in the cosine lobe, the squared roughnesses in the microfacet
// distributions): the reciprocal is a constant too, so the quotient is the product and ONE residual correction -- 3 instructions
// for the compiler's 11.  y must be RN(1 / c).  Exactness is not argued but TESTED, pair by pair, over every one of the 2^32
// numerators (tests/test_gpu_unary.py, ops 3 ..): only the pairs listed here may be used.
struct DivConst { float c, y; };
TRC_DEV DivConst div_by_pi()     { DivConst d; d.c = 3.14159265358979323846f; d.y = 0x1.45f306p-2f; return d; }
TRC_DEV DivConst div_by_sqr001() { DivConst d; d.c = 0.01f * 0.01f; d.y = 10000.0f; return d; }            // alpha 0.01, squared
TRC_DEV DivConst div_by_sqr002() { DivConst d; d.c = 0.02f * 0.02f; d.y = 2500.0f; return d; }             // alpha 0.02
TRC_DEV DivConst div_by_sqr01()  { DivConst d; d.c = 0.1f * 0.1f; d.y = 0x1.8ffffep+6f; return d; }        // alpha 0.1
TRC_DEV float div_const_core(float x, const DivConst& d) {
    const float q0 = x * d.y;
    return __builtin_fmaf(__builtin_fmaf(-d.c, q0, x), d.y, q0);
}
// (TRC_DIVCONST_GUARD_CLASS=1 -- "the product x * y is a normal number", one v_cmp_class -- is NOT enough: the exhaustive test
// finds numerators whose product is normal but whose residual underflows; the range test stays)
#if TRC_WAVE_GUARDS
TRC_DEV bool div_const_ok(float x, const DivConst& d) {
    return unary_in_range(x);
}
#endif
TRC_DEV float div_const(float x, const DivConst& d) {
#if TRC_WAVE_GUARDS
    float q = div_const_core(x, d);
    if (__builtin_expect(!wave_all(div_const_ok(x, d)), 0)) q = x / d.c;
    return q;
#else
    return x / d.c;
#endif
}
// two numerators, two constant divisors, one guard
TRC_DEV void div_const2(float x0, const DivConst& d0, float x1, const DivConst& d1, float& q0, float& q1) {
#if TRC_WAVE_GUARDS
    q0 = div_const_core(x0, d0); q1 = div_const_core(x1, d1);
    if (__builtin_expect(!wave_all(div_const_ok(x0, d0) && div_const_ok(x1, d1)), 0)) { q0 = x0 / d0.c; q1 = x1 / d1.c; }
#else
    q0 = x0 / d0.c; q1 = x1 / d1.c;
#endif
}
TRC_DEV float div_pi(float x) { return div_const(x, div_by_pi()); }
TRC_DEV F3 rcp_cr(F3 a) {                         // 1 / direction: one wave-level branch for the three
#if defined(__HIP_DEVICE_COMPILE__) && TRC_FAST_UNARY && !defined(TRC_FAST_MATH)
    const F3 r0 = f3(__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y), __builtin_amdgcn_rcpf(a.z));
    auto finish = [](float x, float e0) {
        const float r1 = __builtin_fmaf(__builtin_fmaf(-x, e0, 1.0f), e0, e0);
        return __builtin_fmaf(__builtin_fmaf(-x, r1, 1.0f), r1, r1);
    };
    F3 r = f3(finish(a.x, r0.x), finish(a.y, r0.y), finish(a.z, r0.z));
    const bool ok = __builtin_amdgcn_classf(r0.x, 0x108) && __builtin_amdgcn_classf(r0.y, 0x108) && __builtin_amdgcn_classf(r0.z, 0x108);
    if (__builtin_expect(!wave_all(ok), 0)) r = f3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
    return r;
#else
    return f3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
#endif
}
TRC_DEV float length(F3 a) { return sqrt_cr(dot(a, a)); }
TRC_DEV F3 normalize(F3 a) { float inv = rsqrt_cr(dot(a, a)); return a * inv; }
// per-lane dynamic component access stays in registers (select chains, no scratch)
TRC_DEV float comp(F3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
TRC_DEV void set_comp(F3& a, uint32_t i, float v) { if (i == 0) a.x = v; else if (i == 1) a.y = v; else a.z = v; }
TRC_DEV float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
TRC_DEV float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
TRC_DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
TRC_DEV bool is_inf(float x) { return fabsf(x) == __builtin_inff(); }
TRC_DEV bool is_nan(float x) { return x != x; }
TRC_DEV float canon_nan(float x) { return x != x ? __uint_as_float(0x7FC00000u) : x; }       // stored NaNs: one bit pattern (trc_sppm.hip)
TRC_DEV F3 canon_nan(F3 v) { F3 r; r.x = canon_nan(v.x); r.y = canon_nan(v.y); r.z = canon_nan(v.z); return r; }

// ---------------------------------------------------------------- a / b for several a's and one b, same bits as `/`
// hipcc's correctly rounded binary32 division is 11 instructions: v_div_scale x 2 (bring the operands into a range where
// the arithmetic below cannot under- / overflow), v_rcp_f32, a Newton step on the reciprocal (2 fma), the product, two fused
// residual corrections (4 fma), v_div_fmas / v_div_fixup (undo the scaling, special operands).  The reciprocal and its Newton
// step depend on b alone: GuardedDivBy takes them once and a quotient keeps the product and the SAME two corrections.  What
// v_div_scale / v_div_fixup do for free has to be paid here: the fast path is taken only when both operands lie in
// [2^-60, 2^60] (no scaling would have happened: exponent difference < 96, no denormal operand, quotient or reciprocal), a
// zero numerator is the signed zero of the product, everything else -- denormals, huge values, infinities, NaNs, a zero
// divisor -- goes through the plain division.  tests/test_gpu_divby.py compares it with `/` bit for bit over adversarial
// operand pairs (trc_div_by_test).  Whether it pays depends on how many quotients share a divisor: docs/HISTORY.md section 9.
struct GuardedDivBy { float b, y; bool ok; };
TRC_DEV bool div_in_range(float x) { const float ax = fabsf(x); return ax >= 0x1p-60f && ax <= 0x1p60f; }
TRC_DEV GuardedDivBy guarded_div_by(float b) {
    GuardedDivBy d; d.b = b; d.ok = div_in_range(b);
    const float y0 = __builtin_amdgcn_rcpf(b);
    d.y = __builtin_fmaf(__builtin_fmaf(-b, y0, 1.0f), y0, y0);
    return d;
}
TRC_DEV float guarded_div(float a, const GuardedDivBy& d) {
    if (!(d.ok && (div_in_range(a) || a == 0.0f))) return a / d.b;
    const float q0 = a * d.y;
    float q = __builtin_fmaf(__builtin_fmaf(-d.b, q0, a), d.y, q0);
    q = __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.y, q);
    return a == 0.0f ? q0 : q;
}
// the quotient's arithmetic alone (operands known to be in range, numerator not zero): the wave-uniform form of the guard
// lives at the call site (cube slab: nine operands, one ballot)
TRC_DEV float div_core(float a, const GuardedDivBy& d) {
    const float q0 = a * d.y;
    const float q = __builtin_fmaf(__builtin_fmaf(-d.b, q0, a), d.y, q0);
    return __builtin_fmaf(__builtin_fmaf(-d.b, q, a), d.y, q);
}
TRC_DEV float div_core(float a, float b) { return div_core(a, guarded_div_by(b)); }      // one quotient, no guard at all
TRC_DEV F3 div_core(F3 a, F3 b) { return f3(div_core(a.x, b.x), div_core(a.y, b.y), div_core(a.z, b.z)); }
TRC_DEV F3 guarded_div(F3 a, const GuardedDivBy& d) { return f3(guarded_div(a.x, d), guarded_div(a.y, d), guarded_div(a.z, d)); }
TRC_DEV F3 div_shared(F3 a, float s) {      // vec3 / scalar at the sites that share the divisor
    return a / s;
}

constexpr float kPi = 3.14159265358979323846f;      // M_PI_F
constexpr float kPi2 = 1.57079632679489661923f;     // M_PI_2_F

}  // namespace trcdev

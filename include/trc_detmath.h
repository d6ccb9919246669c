/*
 * trc_detmath.h -- deterministic single-precision elementary functions.
 *
 * WHY: the reference shades with Metal stdlib intrinsics under MTL_FAST_MATH
 * (RT_Metal/Tracer.xcodeproj/project.pbxproj:525), whose results are
 * unspecified and not reproducible off Apple hardware.  Path tracing is
 * chaotic: a 1-ulp difference in one sin() changes which primitive a later
 * bounce hits.  To make "HIP kernel == CPU oracle" checkable BIT-EXACTLY for
 * whole frames (not just statistically), both sides compute sin/cos/exp/log/
 * pow/asin/acos/atan2 with THIS header: plain IEEE-754 binary32 + - * / sqrt
 * (correctly rounded on x86-64 SSE and on gfx950), no FMA contraction
 * (both sides compile with -ffp-contract=off), fixed evaluation order.
 *
 * The algorithms are the classic single-precision Cephes routines
 * (S. Moshier, sinf.c/cosf.c/expf.c/logf.c/asinf.c/atanf.c): Cody-Waite
 * range reduction + minimax polynomials, accurate to ~1-2 ulp on the ranges
 * the path uses.  tests/test_detmath.py bounds the error against libm.
 *
 * This is an arithmetic contract, not part of the reference's algorithm.
 * frexpf/ldexpf/floorf/sqrtf/fabsf are exact (or correctly rounded)
 * operations on both platforms and are used as such.
 */
#ifndef TRC_DETMATH_H
#define TRC_DETMATH_H

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define TRC_HD __host__ __device__ inline
#else
#define TRC_HD static inline
#endif

/* the correctly rounded square root: sqrtf, or the including file's own statement of it (the HIP kernels' guard-free form,
 * tracer_amd/csrc/dev_vec.hpp -- the same bits for every operand, tests/test_gpu_unary.py) */
#ifndef DM_SQRTF
#define DM_SQRTF(x) sqrtf(x)
#endif

#define DM_PI_F      3.14159265358979323846f   /* M_PI_F */
#define DM_PIO2_F    1.57079632679489661923f
#define DM_PIO4_F    0.78539816339744830962f
#define DM_INF_F     __builtin_inff()
#define DM_NAN_F     __builtin_nanf("")

/* sin and cos of x, |x| < 8192; shared Cody-Waite reduction to [-pi/4, pi/4] */
/* libtracer_amd_fast.so (TRC_FAST_MATH, device code only): the hardware's own transcendentals (v_exp_f32, v_log_f32,
 * v_sin_f32, v_cos_f32) instead of the reproducible polynomials -- that build is not compared bit for bit anyway */
#if defined(TRC_FAST_MATH) && defined(__HIP_DEVICE_COMPILE__) && !defined(TRC_FAST_KEEP_DETMATH)
#define DM_FAST_DEVICE 1
#else
#define DM_FAST_DEVICE 0
#endif

TRC_HD void dm_sincosf(float xx, float* s_out, float* c_out) {
#if DM_FAST_DEVICE
    *s_out = __sinf(xx); *c_out = __cosf(xx); return;
#endif
    const float FOPI = 1.27323954473516f;          /* 4/pi */
    const float DP1 = 0.78515625f;
    const float DP2 = 2.4187564849853515625e-4f;
    const float DP3 = 3.77489497744594108e-8f;

    float x = fabsf(xx);
    int sin_neg = xx < 0.0f;
    int cos_neg = 0;

    int j = (int)(FOPI * x);        /* integer part of x/(pi/4) */
    if (j & 1) j += 1;              /* map zeros to origin */
    float y = (float)j;
    j &= 7;
    if (j > 3) { sin_neg = !sin_neg; cos_neg = !cos_neg; j -= 4; }
    if (j > 1) cos_neg = !cos_neg;

    x = ((x - y * DP1) - y * DP2) - y * DP3;
    float z = x * x;

    /* sine and cosine polynomials on [-pi/4, pi/4] */
    float ps = ((-1.9515295891E-4f * z + 8.3321608736E-3f) * z - 1.6666654611E-1f) * z * x + x;
    float pc = ((2.443315711809948E-005f * z - 1.388731625493765E-003f) * z + 4.166664568298827E-002f) * z * z
               - 0.5f * z + 1.0f;

    float s, c;
    if (j == 1 || j == 2) { s = pc; c = ps; } else { s = ps; c = pc; }
    *s_out = sin_neg ? -s : s;
    *c_out = cos_neg ? -c : c;
}

TRC_HD float dm_sinf(float x) { float s, c; dm_sincosf(x, &s, &c); return s; }
TRC_HD float dm_cosf(float x) { float s, c; dm_sincosf(x, &s, &c); return c; }

/* e^x for MINLOGF <= x <= MAXLOGF (the caller knows: no NaN, no overflow, no underflow to zero) */
TRC_HD float dm_expf_fin(float x) {
#if DM_FAST_DEVICE
    return __expf(x);
#endif
    const float LOG2EF = 1.44269504088896341f;
    const float C1 = 0.693359375f;
    const float C2 = -2.12194440e-4f;

    /* e^x = e^g 2^n */
    float z = floorf(LOG2EF * x + 0.5f);
    x = x - z * C1;
    x = x - z * C2;
    int n = (int)z;

    z = x * x;
    z = (((((1.9875691500E-4f * x + 1.3981999507E-3f) * x + 8.3334519073E-3f) * x
           + 4.1665795894E-2f) * x + 1.6666665459E-1f) * x + 5.0000001201E-1f) * z + x + 1.0f;
    return ldexpf(z, n);
}

TRC_HD float dm_expf(float xx) {
#if DM_FAST_DEVICE
    return __expf(xx);
#endif
    const float MAXLOGF = 88.72283905206835f;
    const float MINLOGF = -103.278929903431851103f;   /* log(2^-149) */
    float x = xx;
    if (x != x) return x;
    if (x > MAXLOGF) return DM_INF_F;
    if (x < MINLOGF) return 0.0f;
    return dm_expf_fin(x);
}

/* log(x) for finite x > 0 (the caller knows) */
TRC_HD float dm_logf_pos(float x) {
#if DM_FAST_DEVICE
    return __logf(x);
#endif
    const float SQRTHF = 0.707106781186547524f;
    int e;
    x = frexpf(x, &e);
    if (x < SQRTHF) { e -= 1; x = x + x - 1.0f; } else { x = x - 1.0f; }
    float z = x * x;
    float y = ((((((((7.0376836292E-2f * x - 1.1514610310E-1f) * x + 1.1676998740E-1f) * x
                    - 1.2420140846E-1f) * x + 1.4249322787E-1f) * x - 1.6668057665E-1f) * x
                 + 2.0000714765E-1f) * x - 2.4999993993E-1f) * x + 3.3333331174E-1f) * x * z;
    float fe = (float)e;
    if (e != 0) y = y + (-2.12194440e-4f * fe);
    y = y + (-0.5f * z);
    z = x + y;
    if (e != 0) z = z + 0.693359375f * fe;
    return z;
}

TRC_HD float dm_logf(float xx) {
#if DM_FAST_DEVICE
    return __logf(xx);
#endif
    float x = xx;
    if (x != x) return x;
    if (x <= 0.0f) return (x == 0.0f) ? -DM_INF_F : DM_NAN_F;
    if (x == DM_INF_F) return x;
    return dm_logf_pos(x);
}

/* x^y for x >= 0 (the path only raises [0,1] bases to positive powers) */
TRC_HD float dm_powf(float x, float y) {
#if DM_FAST_DEVICE
    return __powf(x, y);
#endif
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : DM_INF_F;
    return dm_expf(y * dm_logf(x));
}

/* asin and acos are written without early returns: every operand runs the SAME instruction stream (a wavefront whose lanes
 * fall into different branches would otherwise run them one after the other -- acos used to inline asin three times), and the
 * special operands are patched in at the end.  Per operand the operations and their order are those of the Cephes routines. */
TRC_HD float dm_asinf(float xx) {
    const float a = fabsf(xx);
    const int big = a > 0.5f;
    float z = big ? 0.5f * (1.0f - a) : a * a;
    const float x = big ? DM_SQRTF(z) : a;
    z = ((((4.2163199048E-2f * z + 2.4181311049E-2f) * z + 4.5470025998E-2f) * z
          + 7.4953002686E-2f) * z + 1.6666752422E-1f) * z * x + x;
    if (big) { z = z + z; z = DM_PIO2_F - z; }
    float r = (xx < 0.0f) ? -z : z;
    if (a < 1.0e-4f) r = xx;
    if (a != a) r = a;
    if (a > 1.0f) r = DM_NAN_F;
    return r;
}

TRC_HD float dm_acosf(float x) {
    const int lo = x < -0.5f, hi = x > 0.5f;
    const float t = lo ? x : -x;                       /* 0.5 (1 + x) below -0.5, 0.5 (1 - x) above 0.5 */
    const float s = dm_asinf((lo || hi) ? DM_SQRTF(0.5f * (1.0f + t)) : x);
    if (lo) return DM_PI_F - 2.0f * s;
    if (hi) return 2.0f * s;
    return DM_PIO2_F - s;
}

TRC_HD float dm_atanf(float xx) {
    float x = fabsf(xx);
    float y;
    if (x != x) return x;
    if (x > 2.414213562373095f) { y = DM_PIO2_F; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = DM_PIO4_F; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y = y + ((((8.05374449538e-2f * z - 1.38776856032E-1f) * z + 1.99777106478E-1f) * z
              - 3.33329491539E-1f) * z * x + x);
    return (xx < 0.0f) ? -y : y;
}

TRC_HD float dm_atan2f(float y, float x) {
    if (x != x || y != y) return DM_NAN_F;
    if (x == 0.0f) {
        if (y > 0.0f) return DM_PIO2_F;
        if (y < 0.0f) return -DM_PIO2_F;
        return 0.0f;
    }
    if (y == 0.0f) return (x > 0.0f) ? 0.0f : DM_PI_F;
    float z = dm_atanf(y / x);
    if (x < 0.0f) z = (y < 0.0f) ? (z - DM_PI_F) : (z + DM_PI_F);
    return z;
}

#endif /* TRC_DETMATH_H */

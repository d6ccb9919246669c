/*
 * tracer_test_hooks.h -- entry points that exist ONLY in libtracer_amd_hooks.so (the sources of libtracer_amd.so compiled
 * with -DTRC_TEST_HOOKS).  They hold device-side arithmetic of the render / SPPM kernels up for inspection by tests/ and
 * tools/; a host never needs them and the product libraries do not export them (tests/test_abi.py checks both lists).
 * Everything a host calls -- including the Scene::hit hook trc_trace_rays, SURVEY 8(b) -- is in tracer_abi.h.
 */
#ifndef TRACER_TEST_HOOKS_H
#define TRACER_TEST_HOOKS_H

#include "tracer_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* developer diagnostic (divergence / cycle profile of the instrumented kernels), per site i since the
 * last trc_reset_stats: out[3*i] = lanes, out[3*i+1] = wavefronts that executed the site, out[3*i+2] =
 * shader-clock cycles those wavefronts spent inside it; sites: 0 loop iteration, 1 box step, 2 square,
 * 3 sphere, 4 cube, 5 triangle, 6 shade, 7 cosine lobe, 8 Metal, 9 Beckmann sampling,
 * 10 Beckmann lobe evaluation (Plastic specular + Glass), 11 path end */
trc_status trc_debug_profile(trc_ctx* ctx, uint64_t* out, uint32_t n_sites);
/* test hook like trc_trace_rays: `hash()` of Photon.hh:71-89 for n cell indices (3 floats each) at one hash scale, as
 * the hashing and refine passes evaluate it (the index into the 512 x 512 grid, before the -1 shift) */
trc_status trc_sppm_hash_cells(trc_ctx* ctx, const float* cells /* n*3 */, size_t n, float hash_scale, float* out /* n */);

/* test hook: the guarded shared-divisor division of tracer_amd/csrc/dev_vec.hpp (one refined reciprocal per divisor, the
 * compiler's own two fused corrections per quotient, plain `/` outside [2^-60, 2^60]) against the plain division, for n
 * operand pairs: fast / plain receive 3 quotients per pair (a / b, -a / b, (0.75 a) / b).  They must agree bit for bit. */
trc_status trc_div_by_test(trc_ctx* ctx, const float* a, const float* b, size_t n, float* fast /* 3 n */, float* plain /* 3 n */);
/* test hook: the render kernels' guard-free reciprocal / square root / reciprocal square root (dev_vec.hpp: rcp_cr, sqrt_cr, rsqrt_cr;
 * op 0 / 1 / 2) against the compiler's correctly rounded 1.0f / x, sqrtf(x), 1.0f / sqrtf(x) on the `count` operands whose bit patterns
 * start at `first_bits` (count = 2^32 covers every float): the number of operands whose results differ, and the smallest one.
 * op 3 .. 6: x / c for the divisors known when the kernels are written (pi, 0.01^2, 0.02^2, 0.1^2), computed as the product with
 * RN(1 / c) and one residual correction, against the compiler's x / c */
trc_status trc_unary_test(trc_ctx* ctx, uint32_t op, uint32_t first_bits, uint64_t count, uint64_t* n_mismatch, uint32_t* first_mismatch);

#ifdef __cplusplus
}
#endif
#endif /* TRACER_TEST_HOOKS_H */

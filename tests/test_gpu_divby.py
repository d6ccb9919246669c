"""The guarded shared-divisor division (tracer_amd/csrc/dev_vec.hpp::GuardedDivBy) against the compiler's correctly rounded
`/`, bit for bit, over operand pairs chosen to break it: every exponent against every exponent, denormals, the largest and
smallest normals, zeros of both signs, infinities, NaNs, operands at the edges of the guarded range, quotients that round
up / down / to even.  The SPPM hash uses the unguarded form on operands it knows (test_gpu_sppm.py); this is the general
one.  It is NOT wired into the render kernels: with the guards it saves 2-4 of 22 instructions per pair of
quotients (docs/HISTORY.md section 9); profiles/r05/exp_removed_variants.patch + -DTRC_DIVBY_RENDER=1 is the variant that was measured."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pairs():
    rs = np.random.RandomState(20)
    f32 = np.float32
    special = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, np.finfo(f32).max, -np.finfo(f32).max, np.finfo(f32).tiny,
                        -np.finfo(f32).tiny, 1e-45, -1e-45, 1.1754942e-38, 2 ** -60, -(2 ** -60), 2 ** 60, 2 ** -61, 2 ** 61,
                        np.nextafter(f32(2 ** -60), f32(0)), np.nextafter(f32(2 ** 60), f32(np.inf)), 3.0, 1 / 3, 0.1, 16777216.0, 16777217.0], f32)
    a1, b1 = np.meshgrid(special, special)                                      # every special against every special
    e = np.arange(-149, 128)                                                    # every exponent against every exponent, random mantissas
    ea, eb = np.meshgrid(e, e)
    man = lambda n: (1 + rs.randint(0, 2 ** 23, n) / 2.0 ** 23)
    a2 = np.ldexp(man(ea.size), ea.ravel()) * rs.choice([-1, 1], ea.size)
    b2 = np.ldexp(man(eb.size), eb.ravel()) * rs.choice([-1, 1], eb.size)
    n = 400000                                                                  # the range the render kernels live in, densely
    a3 = rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-12, 6, n)
    b3 = rs.uniform(-1, 1, n) * 10.0 ** rs.uniform(-12, 6, n)
    b3[::97] = rs.randint(1, 1 << 20, len(b3[::97]))                            # small integers (the running mean's frame + 1)
    # quotients near ties: a = q * b for q with trailing 1-bit patterns
    q = (1 + (rs.randint(0, 2 ** 12, 50000) * 2 + 1) / 2.0 ** 24)
    b4 = rs.uniform(0.5, 2.0, 50000).astype(f32).astype(np.float64)
    a4 = q * b4
    with np.errstate(over="ignore"):
        a = np.concatenate([a1.ravel(), a2, a3, a4]).astype(f32)
        b = np.concatenate([b1.ravel(), b2, b3, b4]).astype(f32)
    return a, b


def test_guarded_division_equals_plain_division_bit_for_bit(gpu_hooks):
    a, b = _pairs()
    fast, plain = gpu_hooks.div_by_test(a, b)
    fb, pb = fast.view(np.uint32), plain.view(np.uint32)
    same = (fb == pb) | (np.isnan(fast) & np.isnan(plain))
    bad = np.argwhere(~same)
    assert len(bad) == 0, (len(bad), [(a[i], b[i], fast[i, j], plain[i, j]) for i, j in bad[:5]])
    # and the plain division is the correctly rounded one (float64 quotient rounded once is exact enough to tell for
    # operands whose quotient is a normal number)
    with np.errstate(all="ignore"):
        want = (a.astype(np.float64) / b.astype(np.float64)).astype(np.float32)
    ok = np.isfinite(want) & (np.abs(want) > 1e-37) & np.isfinite(a) & np.isfinite(b) & (b != 0)
    assert (plain[ok, 0] == want[ok]).mean() > 0.9999          # double rounding can differ on a handful of ties
    assert same.all() and len(a) > 500000

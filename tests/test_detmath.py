"""include/trc_detmath.h against float64 numpy: error bounds on the ranges the path uses, and
special values.  (Both the HIP kernels and the oracle use these functions, so they must be ACCURATE,
not only deterministic.)"""
import numpy as np
import pytest

from oracle import pyoracle as po

SIN, COS, EXP, LOG, POW, ASIN, ACOS, ATAN2 = range(8)


def dm(fn, a, b=None):
    L = po.lib()
    a = np.asarray(a, dtype=np.float32)
    b = np.zeros_like(a) if b is None else np.asarray(b, dtype=np.float32)
    return np.array([L.orc_math(fn, float(x), float(y)) for x, y in zip(a, b)], dtype=np.float32)


def ulp_err(got, want64):
    want32 = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(np.finfo(np.float32).tiny))
    return np.abs(got.astype(np.float64) - want64) / ulp


RS = np.random.RandomState(0)


def test_sin_cos_on_path_range():
    x = np.concatenate([RS.uniform(-8 * np.pi, 8 * np.pi, 4000), np.linspace(-np.pi, np.pi, 257)]).astype(np.float32)
    for fn, ref in ((SIN, np.sin), (COS, np.cos)):
        got = dm(fn, x)
        want = ref(x.astype(np.float64))
        # absolute error near zeros of the function, ulp error elsewhere
        err = np.abs(got - want)
        assert (err < 2.5e-7).all(), err.max()
        big = np.abs(want) > 0.1
        assert ulp_err(got[big], want[big]).max() < 2.5


def test_exp_log_pow():
    x = RS.uniform(-87, 20, 3000).astype(np.float32)
    assert ulp_err(dm(EXP, x), np.exp(x.astype(np.float64))).max() < 2.5
    y = np.concatenate([RS.uniform(1e-30, 1, 2000), RS.uniform(1, 1e6, 1000)]).astype(np.float32)
    got, want = dm(LOG, y), np.log(y.astype(np.float64))
    far = np.abs(want) > 0.05
    assert ulp_err(got[far], want[far]).max() < 2.5 and np.abs(got - want)[~far].max() < 1e-7
    base = RS.uniform(1e-6, 1, 2000).astype(np.float32)
    ex = RS.uniform(0.3, 1.2, 2000).astype(np.float32)    # BeckmannSample11: pow(1 - sample_x, fit)
    want = np.power(base.astype(np.float64), ex.astype(np.float64))
    assert (np.abs(dm(POW, base, ex) - want) / want).max() < 2e-6


def test_inverse_trig():
    x = np.concatenate([RS.uniform(-1, 1, 3000), [-1, 1, 0, 0.5, -0.5, 1e-5]]).astype(np.float32)
    assert np.abs(dm(ASIN, x) - np.arcsin(x.astype(np.float64))).max() < 4e-7
    assert np.abs(dm(ACOS, x) - np.arccos(x.astype(np.float64))).max() < 6e-7
    a, b = RS.normal(size=3000).astype(np.float32), RS.normal(size=3000).astype(np.float32)
    assert np.abs(dm(ATAN2, a, b) - np.arctan2(a.astype(np.float64), b.astype(np.float64))).max() < 6e-7


def test_special_values():
    L = po.lib()
    assert L.orc_math(EXP, -200.0, 0) == 0.0 and np.isinf(L.orc_math(EXP, 100.0, 0))
    assert L.orc_math(EXP, 0.0, 0) == 1.0
    assert np.isneginf(L.orc_math(LOG, 0.0, 0)) and np.isnan(L.orc_math(LOG, -1.0, 0)) and L.orc_math(LOG, 1.0, 0) == 0.0
    assert L.orc_math(POW, 0.0, 0.7) == 0.0 and L.orc_math(POW, 0.3, 0.0) == 1.0
    assert np.isnan(L.orc_math(ASIN, 1.5, 0))
    assert L.orc_math(ATAN2, 0.0, -1.0) == pytest.approx(np.pi, abs=1e-6)
    assert L.orc_math(SIN, 0.0, 0) == 0.0 and L.orc_math(COS, 0.0, 0) == 1.0


def test_libm_variant_is_really_libm():
    assert po.lib().orc_uses_libm() == 0 and po.lib(libm=True).orc_uses_libm() == 1


def test_asin_acos_keep_the_bits_of_the_branching_form():
    """asin / acos were rewritten without early returns (one instruction stream for all lanes of a wavefront); the committed
    CRCs come from the header BEFORE that (tests/golden/make_detmath_crc.py): every 10001st float, all exponents and specials."""
    import json, os, zlib
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "detmath_crc.json")))
    L = po.lib()
    bits = np.arange(0, 1 << 32, gold["stride"], dtype=np.uint64).astype(np.uint32)
    x = bits.view(np.float32)
    x = x[~np.isnan(x)]
    assert len(x) == gold["operands"]
    for fn, key in ((ASIN, "asin_crc32"), (ACOS, "acos_crc32")):
        out = np.fromiter((L.orc_math(fn, float(v), 0.0) for v in x), dtype=np.float32, count=len(x))
        assert f"{zlib.crc32(out.tobytes()) & 0xFFFFFFFF:08x}" == gold[key]

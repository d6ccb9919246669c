"""PCG32 (Random.metal:3-26 == RT_Metal/Tracer/pcg_basic.c:42-72): the one executable pin of the oracle."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import host

KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "pcg32_kat.json")))


def _oracle_stream(initstate, initseq, n):
    L = po.lib()
    s, inc = C.c_uint64(), C.c_uint64()
    L.orc_pcg32_srandom(C.byref(s), C.byref(inc), initstate, initseq)
    s0 = s.value
    return s0, inc.value, [L.orc_pcg32_random(C.byref(s), inc.value) for _ in range(n)], s.value


def test_survey_known_answer():
    # pcg32_srandom_r(42, 54)
    _, inc, outs, _ = _oracle_stream(42, 54, 6)
    assert [hex(x) for x in outs] == ["0xa15c02b7", "0x7b47f409", "0xba1d3330", "0x83d2f293", "0xbfa4784b", "0xcbed606e"]
    assert inc == 0x6D


@pytest.mark.parametrize("case", KAT["srandom"], ids=lambda c: f"{c['initstate']:x}-{c['initseq']:x}")
def test_oracle_matches_reference_golden_vectors(case):
    s0, inc, outs, s1 = _oracle_stream(case["initstate"], case["initseq"], len(case["outputs"]))
    assert (s0, inc, outs, s1) == (case["state_after_seed"], case["inc"], case["outputs"], case["state_after_outputs"])


@pytest.mark.parametrize("case", KAT["raw"], ids=lambda c: f"{c['state']:x}")
def test_oracle_raw_stepping_including_even_inc(case):
    L = po.lib()
    s = C.c_uint64(case["state"])
    outs = [L.orc_pcg32_random(C.byref(s), case["inc"]) for _ in range(len(case["outputs"]))]
    assert outs == case["outputs"] and s.value == case["state_after"]


def test_oracle_matches_live_reference_build_when_present():
    ref_path = os.path.join(ROOT, "oracle", "_ref", "libpcg_ref.so")
    if not os.path.exists(ref_path):
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    ref = C.CDLL(ref_path)

    class Pcg(C.Structure):
        _fields_ = [("state", C.c_uint64), ("inc", C.c_uint64)]
    ref.pcg32_srandom_r.argtypes = [C.POINTER(Pcg), C.c_uint64, C.c_uint64]
    ref.pcg32_random_r.argtypes = [C.POINTER(Pcg)]
    ref.pcg32_random_r.restype = C.c_uint32
    rs = np.random.RandomState(3)
    for _ in range(200):
        a, b = (int(x) for x in rs.randint(0, 2**63, 2, dtype=np.int64))
        r = Pcg()
        ref.pcg32_srandom_r(C.byref(r), a, b)
        want = [ref.pcg32_random_r(C.byref(r)) for _ in range(5)]
        assert _oracle_stream(a, b, 5)[2] == want


def test_randomF_is_ldexp_of_u32_and_can_reach_one():
    L = po.lib()
    s = C.c_uint64(0x0123456789ABCDEF)
    s2 = C.c_uint64(s.value)
    inc = 0xDA3E39CB94B95BDB
    for _ in range(100):
        u = L.orc_pcg32_random(C.byref(s), inc)
        f = L.orc_randomF(C.byref(s2), inc)
        assert np.float32(f) == np.ldexp(np.float32(u), -32)
    assert np.ldexp(np.float32(0xFFFFFFFF), -32) == np.float32(1.0)   # SURVEY B-4


def test_host_fill_rng_is_per_pixel_srandom():
    W, H, seed = 37, 11, 0x5EED0000
    rng = host.fill_rng(seed, W, H)
    for (x, y) in [(0, 0), (36, 10), (5, 7)]:
        _, _, outs, _ = _oracle_stream(seed, y * W + x, 4)
        assert list(rng[y, x]) == outs

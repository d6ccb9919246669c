"""pbrt::SobolSampler (SobolSampler.hh:26-167, wired at Render.metal:529-530): the generated tables of
include/trc_sobol.h against the reference's own (tests/golden/sobol_tables.json, made from Sobolmatrices.metal by
tests/golden/make_sobol_fixture.py), the oracle's sampler against values computed on the reference's tables, and the
properties the construction promises."""
import json
import os
import zlib

import numpy as np
import pytest

from oracle import pyoracle
from tracer_amd import abi, host

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sobol_tables.json")))


def test_generator_matrices_equal_the_reference_table():
    m = host.sobol_matrices32()
    assert m.shape == (40, 52)
    for d in range(abi.SOBOL_DIMS):
        assert zlib.crc32(m[d].astype("<u4").tobytes()) == GOLD["matrices32_crc32"][d], f"dimension {d}"
    assert m[0, 0] == 0x80000000 and m[0, 31] == 1 and not m[0, 32:].any()         # van der Corput
    assert list(m[1, :4]) == [0x80000000, 0xC0000000, 0xA0000000, 0xF0000000]       # Sobol' dimension 2


def test_interval_tables_equal_the_reference_tables():
    assert len(GOLD["vdc_crc32"]) == 25 and len(GOLD["inv_crc32"]) == 26
    for m in range(1, 27):
        vdc, inv = host.sobol_interval_tables(m)
        if m <= 25:
            assert zlib.crc32(vdc.astype("<u8").tobytes()) == GOLD["vdc_crc32"][m - 1], f"VdC m={m}"
        else:
            assert not vdc.any()                       # 2m = 52: no index bits above the pixel block
        assert zlib.crc32(inv.astype("<u8").tobytes()) == GOLD["inv_crc32"][m - 1], f"Inv m={m}"
    with pytest.raises(Exception):
        host.sobol_interval_tables(27)


def test_oracle_sampler_against_values_from_the_reference_tables():
    L = pyoracle.lib()
    for index, dim, v in GOLD["sample_u32"]:
        want = min(np.float32(np.uint32(v)) * np.float32(2.3283064365386963e-10), np.float32(1.0) - np.float32(np.finfo(np.float32).eps))
        assert np.float32(L.orc_sobol_sample_float(index, dim)) == np.float32(want), (index, dim)
    for m, s, x, y, want in GOLD["interval_to_index"]:
        assert L.orc_sobol_interval_to_index(m, s, x, y) == want, (m, s, x, y)


def test_first_points_are_the_published_sequence():
    L = pyoracle.lib()
    # van der Corput and the second Sobol' dimension: 0, 1/2, 1/4, 3/4, ... and 0, 1/2, 3/4, 1/4, 5/8, 1/8, ...
    assert [L.orc_sobol_sample_float(i, 0) for i in range(8)] == [0, 0.5, 0.25, 0.75, 0.125, 0.625, 0.375, 0.875]
    assert [L.orc_sobol_sample_float(i, 1) for i in range(8)] == [0, 0.5, 0.75, 0.25, 0.625, 0.125, 0.375, 0.875]
    # every dimension is a (0,1)-sequence: any 2^k consecutive-from-zero points hit every interval of width 2^-k once
    for d in range(abi.SOBOL_DIMS):
        pts = np.array([L.orc_sobol_sample_float(i, d) for i in range(64)])
        assert sorted(np.floor(pts * 64).astype(int)) == list(range(64)), f"dimension {d}"


@pytest.mark.parametrize("m", [1, 3, 5])
def test_interval_to_index_enumerates_each_pixel(m):
    """The s-th sample of pixel (x, y) of a 2^m x 2^m grid lies in that pixel, and the indices are distinct
    (Gruenschloss et al.: the point of SobolIntervalToIndex)."""
    L = pyoracle.lib()
    res, seen = 1 << m, set()
    for s in range(4):
        for x in range(res):
            for y in range(res):
                i = L.orc_sobol_interval_to_index(m, s, x, y)
                assert i >> (2 * m) == s
                assert int(L.orc_sobol_sample_float(i, 0) * res) == x and int(L.orc_sobol_sample_float(i, 1) * res) == y
                seen.add(i)
    assert len(seen) == 4 * res * res


def test_pixel_dimensions_are_offsets_inside_the_pixel():
    L = pyoracle.lib()
    W, H = 100, 60                                  # resolution 128, log2 7
    for frame in (0, 1, 17):
        for x, y in ((0, 0), (99, 59), (37, 12)):
            for d in (0, 1):
                v = L.orc_sobol_sample_dimension(frame, x, y, W, H, d)
                assert 0.0 <= v < 1.0
            assert L.orc_sobol_sample_dimension(frame, x, y, W, H, abi.SOBOL_DIMS) == 0.0    # past the table: 0
    assert L.orc_sobol_sample_dimension(5, 0, 0, 1, 1, 2) == 0.0     # log2Resolution 0: index 0 whatever the frame (:130)


def _render(sobol, integrator=abi.INTEGRATOR_PATH, spp=4, seed=11, frame0=0, W=48, H=32):
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    rng = host.fill_rng(seed, W, H)
    acc, st = pyoracle.render(sc.view, host.prepare_camera(W, H), W, H, rng, spp=spp, integrator=integrator, frame0=frame0,
                              sobol=sobol)
    return acc, rng, st


def test_oracle_render_with_the_sobol_sampler():
    a0, r0, _ = _render(False)
    a1, r1, s1 = _render(True)
    a2, r2, _ = _render(True)
    assert np.array_equal(a1, a2) and np.array_equal(r1, r2)             # deterministic
    assert not np.array_equal(a0, a1)
    assert np.isfinite(a1).all() and s1.rays > 48 * 32 * 4
    # the texel keeps the stream as castRay left it (the sampler owns a copy): independent of what the path consumed
    _, r_mis, _ = _render(True, integrator=abi.INTEGRATOR_MIS)
    assert np.array_equal(r1, r_mis) and not np.array_equal(r0, r1)
    # both estimate the same image
    big0, _, _ = _render(False, spp=256)
    big1, _, _ = _render(True, spp=256)
    assert abs(big0[..., :3].mean() - big1[..., :3].mean()) < 0.03 * big0[..., :3].mean()   # seed-to-seed sigma: 1 %


def test_oracle_reproduces_committed_sobol_frames():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_sobol_frames as fx
    frames = np.load(os.path.join(os.path.dirname(__file__), "golden", "sobol_frames.npz"))
    for name in fx.CASES:
        acc, rng, st = fx.render_case(name)
        assert np.array_equal(rng, frames[name + "_rng"]), name
        assert np.array_equal(acc.view(np.uint32), frames[name + "_accum"].view(np.uint32)), name
        assert [st.paths, st.rays, st.shaded] == list(frames[name + "_counts"]), name

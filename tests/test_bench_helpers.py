"""bench.py pieces that do not need a GPU: the algorithmic-bytes formula of SURVEY.md 8(d) on exact oracle
counters, and the JSON contract of the CPU baseline legs."""
import importlib.util
import os

import numpy as np

from conftest import ROOT
from oracle import pyoracle as po
from tracer_amd import abi, host

spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_algorithmic_bytes_formula():
    W, H, spp = 64, 36, 4
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    _, st = po.render(scene.view, host.prepare_camera(W, H), W, H, host.fill_rng(1, W, H), spp=spp)
    got = bench.algorithmic_bytes(st, W * H)
    want = (88 * st.n_descend + 24 * st.n_return + 20 * st.n_leaf_sphere + 32 * st.n_leaf_square +
            100 * st.n_leaf_cube + 128 * st.n_hit_cube + 64 * st.shaded + 64 * W * H)
    assert got == want and got > 0
    # closest-hit identity of the reference's stackless walk: every visited interior node but the root is left
    # once, every leaf test returns to its parent once  =>  N_return = N_leaf + N_descend - #rays that entered the root
    n_leaf = st.n_leaf_sphere + st.n_leaf_square + st.n_leaf_cube + st.n_leaf_triangle
    assert st.n_descend <= st.n_return + st.rays and st.n_return >= n_leaf
    per_ray = got / st.rays
    assert 300 < per_ray < 3000          # "Cornell ~ 1.1 KB/ray"


def test_cpu_rt_weekend_leg_schema():
    leg = bench.cpu_rt_weekend()
    assert leg is not None and leg["unit"] == "Mrays/s" and leg["kind"] == "port"
    assert leg["value"] > 0 and leg["cores"] >= 1 and "400x400x16" in leg["sample"]


def test_pmc_traffic_is_read_from_profiles():
    t = bench.pmc_traffic(1)
    assert t is None or (isinstance(t, int) and 50e6 < t < 500e6)     # ~32 B/pixel read + 32 B written at 1080p
    assert bench.pmc_traffic(8) is None

"""bench.py pieces that do not need a GPU: the algorithmic-bytes formula of SURVEY.md 8(d) on exact oracle
counters, and the JSON contract of the CPU baseline legs."""
import importlib.util
import os

import numpy as np
import pytest

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


@pytest.mark.gpu
def test_two_rank_bench_path_without_rccl():
    """bench.py launched the way the driver launches N > 1 (torch.distributed.run), two ranks sharing cuda:0 with the
    collective switched off (TRC_BENCH_NO_RCCL=1): rendezvous, stacked two-view workload, tile ownership, max-over-ranks
    timing and the single JSON line on rank 0."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ, TRC_BENCH_NO_RCCL="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29517", os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert "cpu_baseline" not in line                                   # rank 0 at N = 1 only
    assert line["config"]["rays_per_step"] > 4.3e8                      # two views' worth of rays
    assert "2 such views" in line["config"]["workload"]

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


def test_roofline_only_from_a_summary_of_the_running_library(tmp_path, monkeypatch):
    """The line's `roofline` (the binding VALU-issue bound, with traffic / l2_miss_traffic_frac_of_hbm_peak) comes from a committed PMC
    summary and ONLY if it was collected on the library that is running (source hash + kernel name); a stale or missing
    summary yields an object without numbers that says why.  frac is a fraction of a real peak: never above 1."""
    mine = bench.lib_source_hash()
    good = {"kernel": "k_render<true, false, 0, false>", "lib_source_hash": mine, "hbm_bytes_per_launch": 140000000,
            "GRBM_GUI_ACTIVE": 8 * 52.7e6, "SQ_INSTS_VALU": 17.1e9, "SQ_THREAD_CYCLES_VALU": 17.1e9 * 16, "kernel_ms": 22.0}
    monkeypatch.setattr(bench, "pmc_summary", lambda: dict(good, _file="profiles/rXX/pmc_config2.json"))
    r = bench.roofline_from_pmc(22.0)
    assert r["bound"] == "valu-issue" and r["traffic"] == 140000000 and abs(r["l2_miss_traffic_frac_of_hbm_peak"] - 140e6 / 22e-3 / 8e12) < 1e-6
    assert abs(r["frac"] - 17.1e9 * 2 / (1024 * 52.7e6)) < 1e-3 and abs(r["lane_utilisation"] - 0.25) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["frac"] <= 1.0
    assert abs(r["useful_lane_frac"] - r["frac"] * 0.25) < 1e-3 and r["lib_source_hash"] == mine
    monkeypatch.setattr(bench, "pmc_summary", lambda: dict(good, lib_source_hash="0123", _file="x.json"))
    r = bench.roofline_from_pmc(22.0)
    assert r["frac"] is None and r["traffic"] is None and "stale" in r and r["bound"] == "valu-issue"
    monkeypatch.setattr(bench, "pmc_summary", lambda: None)
    r = bench.roofline_from_pmc(22.0)
    assert r["frac"] is None and "stale" in r


def test_pmc_summary_is_chosen_by_round_number(tmp_path, monkeypatch):
    import json
    for rnd in ("r02", "r09", "r10"):
        os.makedirs(tmp_path / "profiles" / rnd)
        (tmp_path / "profiles" / rnd / "pmc_config2.json").write_text(json.dumps({"round": rnd}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.pmc_summary()["round"] == "r10"


def test_self_launch_spawns_the_ranks_and_relays_rank0(tmp_path):
    """`python bench.py --gpus 2` with WORLD_SIZE unset: bench.py starts the two ranks itself (fresh children with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_*), they rendezvous over TCP sockets (tracer_amd/socket_group.py: no PyTorch in any
    rank) and rank 0's single JSON line comes back on stdout.
    --rendezvous-only stops before the first GPU call, so this runs here; the rendering part of the same path runs in
    test_two_rank_bench_path_without_rccl on the GPU box."""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only",
                          "--scaling", "weak"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line == {"rendezvous_only": True, "n_gpus": 2, "ranks": [0, 1], "sum": 3.0, "scaling": "weak", "torch_imported": False}


def test_ranks_started_the_driver_s_way_rendezvous_without_torch():
    """the driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`: MASTER_PORT then belongs to torchrun's own store, and bench.py's ranks must meet
    elsewhere (a port rank 0 publishes in a file keyed by MASTER_PORT and the common parent) -- without importing torch"""
    import json, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
                          "--master-port", "29519", os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rendezvous-only"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout
    assert json.loads(lines[0]) == {"rendezvous_only": True, "n_gpus": 3, "ranks": [0, 1, 2], "sum": 6.0, "scaling": "strong",
                                    "torch_imported": False}


def test_self_launch_does_not_hang_when_a_rank_dies():
    """rank 1 exits at once; rank 0 would wait for it in the rendezvous for ever -- the launcher stops it and reports"""
    import subprocess, sys, time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only", "--fail-rank", "1"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 3 and time.time() - t0 < 60
    assert "a rank failed" in out.stderr and not out.stdout.strip()


def test_the_committed_goldens_cover_every_frame_bench_holds_itself_to():
    """tests/golden/bench_goldens.json (the oracle at the named sizes, make_bench_goldens.py): one entry per frame bench.py checks --
    the headline, the sample-sharded and stacked-view composes for N = 2, 4, 8, the other configurations -- and check_against's verdicts"""
    g = bench.load_goldens()
    for k in ["config2", "config3", "config4", "volume", "config5_sppm"] + [f"config2_samples_S{n}" for n in (2, 4, 8)] + [f"config2_weak_N{n}" for n in (2, 4, 8)]:
        assert k in g and isinstance(g[k]["crc_accum"], int) and 0 <= g[k]["crc_accum"] < 2**32, k
    assert g["config2"]["rays"] == 219978393 and g["config3"]["rays"] == 1677111408 and g["config4"]["rays"] == 938746169 and g["volume"]["rays"] == 445116398
    assert g["config2_weak_N2"]["rays"] > 2 * 0.99 * g["config2"]["rays"] and g["config5_sppm"]["frame_count"] == 64
    assert len({g[f"config2_samples_S{n}"]["crc_accum"] for n in (2, 4, 8)} | {g["config2"]["crc_accum"]}) == 4       # four different sample sets
    ok = bench.check_against(g, "config2", crc=g["config2"]["crc_accum"], rays=219978393)
    assert ok == {"golden": "config2", "crc_ok": True, "rays_ok": True}
    bad = bench.check_against(g, "config2", crc=g["config2"]["crc_accum"] ^ 1, rays=219978392)
    assert bad["crc_ok"] is False and bad["rays_ok"] is False
    none = bench.check_against(g, "config2_samples_S3", crc=1)
    assert none["golden"] is None and "crc_ok" not in none and "make_bench_goldens" in none["missing"]
    s5 = bench.check_against(g, "config5_sppm", crc=g["config5_sppm"]["crc_accum"], extra={"totalPhotonSum": g["config5_sppm"]["totalPhotonSum"], "frame_count": 63})
    assert s5["crc_ok"] and s5["totalPhotonSum_ok"] is True and s5["frame_count_ok"] is False
    assert bench.frame_crc(np.zeros((2, 2, 4), np.float32)) == 0x758D6336                       # zlib.crc32 of 64 zero bytes


def test_other_config_rooflines_follow_the_source_hash_rule(monkeypatch):
    """a leg's `roofline` comes from profiles/rNN/pmc_<config>.json only when that summary was collected on the running library"""
    roof, pm = bench.config_roofline("config4", "k_render_pwg<0, false>")
    mine = bench.lib_source_hash()
    if pm is not None:
        assert pm["lib_source_hash"] == mine and 0 < roof["frac"] <= 1 and roof["bound"] == "valu-issue" and "spp" in roof["pmc_launch"]
    else:
        assert roof["frac"] is None and "stale" in roof
    monkeypatch.setattr(bench, "lib_source_hash", lambda: "0123456789abcdef")
    roof, pm = bench.config_roofline("config4", "k_render_pwg<0, false>")
    assert pm is None and roof["frac"] is None and "was collected on library" in roof["stale"]
    roof, pm = bench.config_roofline("no_such_config", "k")
    assert pm is None and "no profiles" in roof["stale"]


def test_gpu_count_probe_does_not_touch_hip():
    assert isinstance(bench.visible_gpus(), int) and bench.visible_gpus() >= 0


def test_transport_is_decided_from_the_devices_the_ranks_hold():
    """`world > visible_gpus()` took a launcher that isolates every rank with its own
    HIP_VISIBLE_DEVICES=<one id> for "8 ranks on 1 GPU" and sent a real 8-GPU node over TCP sockets.  The decision is now made
    from the PCI bus ids the ranks publish.  Three layouts of 8 ranks on a node whose GPUs sit at these bus ids:"""
    node = [f"0000:{b:02x}:00.0" for b in (0x05, 0x15, 0x65, 0x75, 0x85, 0x95, 0xe5, 0xf5)]

    def ranks(visible_of_rank):
        # what each rank's context reports: the bus id of device pick_device(local_rank, #visible) among the GPUs it can see
        return [node[vis[bench.pick_device(r, len(vis))]] for r, vis in enumerate(visible_of_rank)]
    all_visible = ranks([list(range(8))] * 8)                        # torchrun as the driver starts it: every rank sees 8 devices
    isolated = ranks([[r] for r in range(8)])                        # one HIP_VISIBLE_DEVICES id per rank: every rank sees ONE device
    shared = ranks([[0]] * 8)                                        # 8 ranks on a 1-GPU box
    assert all_visible == node and isolated == node and shared == [node[0]] * 8
    assert bench.compose_transport(all_visible) == "rccl"
    assert bench.compose_transport(isolated) == "rccl"               # the case the environment count got wrong
    assert bench.compose_transport(shared) == "table"
    assert bench.compose_transport(node[:3] + [node[1].upper()]) == "table"      # two ranks on one GPU among distinct ones; case-insensitive
    assert bench.compose_transport(["", node[0]]) == "table" and bench.compose_transport([]) == "table"   # unknown devices: never RCCL blindly
    assert bench.compose_transport([node[0]]) == "rccl"              # a 1-rank group (--force-group) is distinct by definition


@pytest.mark.gpu
@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_rank_bench_path_on_one_gpu(launcher):
    """bench.py with two ranks sharing cuda:0 (more ranks than GPUs: the compose goes through trc_group_set_collectives
    with the socket table instead of RCCL, and the line is marked plumbing), started by bench.py itself and the way the driver
    starts N > 1 (torch.distributed.run): rendezvous, BOTH workloads (the named frame tile-sharded = strong, stacked views =
    weak), tile ownership, the pipelined compose, max-over-ranks timing, one JSON line."""
    import json, subprocess, sys
    env = dict(os.environ, TRC_BENCH_NO_RCCL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    if launcher == "self":
        cmd = [sys.executable] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29517"] + tail
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2
    assert line["metric"] == "Mrays/s at 1920x1080x64spp"
    assert "cpu_baseline" not in line                                   # rank 0 at N = 1 only
    assert line["config"]["rays_per_step"] == 219978393                 # the ONE named frame, whatever N
    assert len(line["per_rank"]) == 2 and sum(r["rays_per_step"] for r in line["per_rank"]) == 219978393
    weak = line["other_scaling"]["weak"]
    assert weak["mode"] == "weak" and "2 stacked" in weak["metric"] and weak["frame"] == [1920, 2160]
    assert weak["rays_per_step"] > 4.3e8                                # two views' worth of rays
    # sample sharding: the same 64 samples per pixel, 32 from each of two seeds; every rank renders the whole frame
    smp = line["other_scaling"]["samples"]
    assert smp["mode"] == "samples" and "split over 2 seeds" in smp["metric"] and smp["frame"] == [1920, 1080]
    assert smp["sample_groups"] == 2 and smp["tile_ranks"] == 1 and [r["spp"] for r in smp["per_rank"]] == [32, 32]
    assert abs(smp["rays_per_step"] / 219978393 - 1) < 0.01             # another sample set of the same frame: the same work to 1 %
    assert all(r["compose_ms"] is not None for r in smp["per_rank"])
    assert line["plumbing"] is True and "PLUMBING" in line["config"]["compose"]
    # every workload's composed frame was held to the oracle's committed CRC32 (tests/golden/bench_goldens.json)
    assert line["composed_crc_ok"] is True and line["oracle_check"]["golden"] == "config2"       # tiles: the one-GPU frame, bit for bit
    assert weak["composed_crc_ok"] is True and weak["oracle_check"]["golden"] == "config2_weak_N2"
    assert smp["composed_crc_ok"] is True and smp["oracle_check"]["golden"] == "config2_samples_S2"
    assert line["transport"] == "table" and len(line["devices"]) == 2 and line["devices"][0] == line["devices"][1]   # one GPU, seen by both ranks
    assert all(r["compose_ms"] is not None for r in line["per_rank"])   # the reduce really ran (host-staged)
    # the cold leg and the parity-imposed bounds ride along for every rank
    assert line["cold"]["settle_launches"] == 0 and line["first_launch_ms"] == line["cold"]["first_launch_ms"] > 0
    assert line["config"]["settle_launches"] == 8
    for r in line["per_rank"]:
        assert 0 < r["longest_chain_ms"] and 0 < r["work_over_slots_ms"] and r["launch_entries"] >= 32400 // 2


def _run_bench(args, env_extra=None, timeout=900):
    import json, subprocess, sys
    env = dict(os.environ, TRC_BENCH_NO_RCCL="1", **(env_extra or {}))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stderr[-2000:]
    return out.returncode, json.loads(lines[0]), out.stderr


@pytest.mark.gpu
def test_a_wrong_fold_order_fails_the_run():
    """bench.py validates what it timed: 4 ranks on one GPU, sample shards.  Folded in rank order the composed frame has the CRC32 the
    oracle's statement of the definition has (render_sample_sharded, committed); the same four shards folded in the reverse order
    (test hook: rank r renders group 3 - r) is another frame -- float addition does not associate -- and the run exits non-zero."""
    rc, line, err = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0", "--scaling", "samples", "--no-other-scaling", "--no-cold"])
    assert rc == 0 and line["composed_crc_ok"] is True and line["oracle_check"]["golden"] == "config2_samples_S4", err[-1500:]
    rc, line, err = _run_bench(["--gpus", "4", "--steps", "1", "--warmup", "0", "--scaling", "samples", "--no-other-scaling", "--no-cold"],
                               {"TRC_BENCH_REVERSE_GROUPS": "1"})
    assert rc == 4 and line["composed_crc_ok"] is False and "does NOT match" in err
    # ... and a frame held to the wrong golden fails likewise (the tile split against the sample-sharded CRC)
    rc, line, err = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-other-scaling", "--no-cold"], {"TRC_BENCH_EXPECT_GOLDEN": "config2_samples_S2"})
    assert rc == 4 and line["composed_crc_ok"] is False


@pytest.mark.gpu
def test_the_one_gpu_line_witnesses_every_baseline_config():
    """N = 1: the headline frame and every other BASELINE configuration are held to the oracle's committed ray counts and CRC32s, and
    the HBM operating point says that north_star's 40 % is not met"""
    rc, line, err = _run_bench(["--steps", "2", "--warmup", "0", "--no-cpu-baseline", "--no-fast-math", "--no-cold"], timeout=1500)
    assert rc == 0, err[-2000:]
    assert line["composed_crc_ok"] is True and line["oracle_check"]["rays_ok"] is True
    oc = line["other_configs"]
    assert set(oc) == {"config3", "config4", "config5_sppm", "volume", "hbm_point"}
    for k in ("config3", "config4", "volume"):
        assert "skipped" not in oc[k], oc[k]
        assert oc[k]["oracle_check"]["crc_ok"] is True and oc[k]["oracle_check"]["rays_ok"] is True, (k, oc[k]["oracle_check"])
        assert oc[k]["value"] > 1000 and oc[k]["steps"] == 2
    s5 = oc["config5_sppm"]
    assert s5["oracle_check"]["crc_ok"] is True and s5["oracle_check"]["totalPhotonSum_ok"] is True and s5["frames_per_step"] == 64
    assert line["hbm_point"]["target_frac"] == 0.40 and line["hbm_point"]["mrays"] > 1000
    assert line["hbm_point"]["target_met"] in (False, None)


@pytest.mark.gpu
def test_an_optional_leg_that_never_finishes_does_not_take_the_headline_with_it():
    """the other_scaling legs run under a watchdog: when one does not finish in time (here: at once -- a stand-in for a collective
    that never completes) rank 0 still prints the ONE line with the finished primary measurement, the leg marked skipped, and
    every rank exits 0"""
    import json, subprocess, sys
    env = dict(os.environ, TRC_BENCH_NO_RCCL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--other-timeout", "0.05"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["rays_per_step"] == 219978393 and line["value"] > 0
    assert "did not finish" in line["other_scaling"]["weak"]["skipped"] and "samples" not in line["other_scaling"]


@pytest.mark.gpu
def test_a_failing_rccl_communicator_does_not_end_the_run():
    """The RCCL compose path has never run with N > 1 on hardware.  Here it FAILS for real (two ranks forced onto RCCL with one
    device: ncclCommInitRank refuses): every rank hears of it, all compose through the socket table, the line says so."""
    import json, subprocess, sys
    env = dict(os.environ, TRC_BENCH_FORCE_RCCL="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TRC_BENCH_NO_RCCL"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-other-scaling"],
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert "ncclCommInitRank" in line["compose_fallback"] and "plumbing" not in line
    assert line["n_gpus"] == 2 and line["config"]["rays_per_step"] == 219978393
    assert all(r["compose_ms"] is not None for r in line["per_rank"])


def test_source_hash_ignores_comments_and_white_space_only():
    a = 'int f(int x) { // add one\n    return x + 1; /* really */ }\nconst char* s = "// not a comment";\n'
    b = 'int f(int x) {\n  return x + 1;\n}\n\nconst char* s = "// not a comment";  // trailing\n'
    c = 'int f(int x) { return x + 2; }\nconst char* s = "// not a comment";\n'
    d = 'int f(int x) { return x + 1; }\nconst char* s = "// not b comment";\n'
    assert bench._code_only(a) == bench._code_only(b)
    assert bench._code_only(a) != bench._code_only(c) and bench._code_only(a) != bench._code_only(d)
    assert len(bench.lib_source_hash()) == 16

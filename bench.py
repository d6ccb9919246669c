#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path on BASELINE config 2.

Workload (BASELINE.json configs[1]): RT_Metal Cornell box + the 12 spheres, 1920x1080, 64 spp,
tracePath, depth 8, synthetic (scene from the reference's constants, per-pixel PCG32 seeds).

A "step" = one pass of the hot path over one batch: re-seed the RNG texture (so every step is the SAME
work), render all 64 samples of this rank's pixel tiles, and -- for N > 1 -- compose the frame with one
RCCL reduce of the accumulation buffer to rank 0.  Inputs are resident in HBM before the timed region.
`rays` = Scene::hit invocations, counted exactly by the kernel.

N = 1: one 1920x1080 view.  N > 1 (weak scaling, the unit that shards is the 16x16 pixel tile): the batch
is N such views stacked into one 1920 x (1080 N) frame (trc_params.view_height = 1080: every view has the
same camera and its own RNG texels), tiles owned round-robin (tx + ty) % N, so every rank renders one
view's worth of tiles drawn evenly from all views -- the per-GPU work of the N = 1 run -- and the composed
N-view frame ends up on rank 0.  (Strong scaling of a single 1080p x 64 spp frame is latency-bound: a pixel's
64 samples are a sequential RNG chain)

  python bench.py [--gpus N --steps K --warmup W]           (N = 1)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the roofline definition).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP, DEPTH = 1920, 1080, 64, 8
SEED = 0x5EED0000
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(st, n_pixels):
    """bytes the reference's algorithm moves for this work (NOT cache-line traffic).
    per ray 88*N_descend + 24*N_return + leaf bytes; per shaded hit 64 B material; pixel state 64 B."""
    leaf = (20 * st.n_leaf_sphere + 32 * st.n_leaf_square + 100 * st.n_leaf_cube + 128 * st.n_hit_cube +
            48 * st.n_leaf_triangle + 60 * st.n_hit_triangle)
    return 88 * st.n_descend + 24 * st.n_return + leaf + 64 * st.shaded + 64 * n_pixels


def cpu_baseline(scene, cam, budget_s=20.0):
    """The CPU oracle (a port of the same algorithm) on a bounded tile subsample of the SAME workload."""
    import numpy as np
    from oracle import pyoracle as po
    from tracer_amd import abi, host
    cores = os.cpu_count() or 1
    rng = host.fill_rng(SEED, W, H)
    # calibrate on 1 tile in 256, then pick the subsample that fits the budget
    t0 = time.perf_counter()
    _, st = po.render(scene.view, cam, W, H, rng, spp=SPP, max_depth=DEPTH, tile_rank=0, tile_nranks=256,
                      n_threads=cores)
    dt = time.perf_counter() - t0
    rate = st.rays / dt
    full_rays_est = st.rays * 256
    nranks = 256
    while nranks > 1 and (full_rays_est / (nranks // 2)) / rate < budget_s:
        nranks //= 2
    if nranks != 256:
        rng = host.fill_rng(SEED, W, H)
        t0 = time.perf_counter()
        _, st = po.render(scene.view, cam, W, H, rng, spp=SPP, max_depth=DEPTH, tile_rank=0, tile_nranks=nranks,
                          n_threads=cores)
        dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"oracle/liboracle.so (scalar C++ restatement), tiles (tx+ty)%{nranks}==0 of the same "
                      f"1920x1080x64spp frame: {st.paths} paths, {st.rays} rays in {dt:.2f} s on {cores} threads"}


def cpu_rt_weekend():
    """BASELINE config 1: the reference's RT_Weekend CPU path (C++ restatement, oracle/cpu_baseline.cpp)
    on the Cornell box at 400x400x16 spp, all host cores, row bands like main.swift:77-87."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "cpu_baseline")
    if not os.path.exists(exe):
        return None
    out = subprocess.check_output([exe, "--scene", "cornell", "--width", "400", "--height", "400", "--spp", "16"],
                                  text=True, timeout=600)
    r = json.loads(out.strip().splitlines()[-1])
    return {"value": r["mrays_per_s"], "unit": "Mrays/s", "cores": r["threads"], "kind": "port",
            "sample": f"RT_Weekend recursive tracer restated in C++ (linear-scan HittableList), RT_Nextweek Cornell "
                      f"box 400x400x16spp: {r['rays']} rays in {r['seconds']:.2f} s, {r['mpaths_per_s']} Mpaths/s"}


def pmc_traffic(world):
    """HBM bytes per launch from rocprofv3 PMC passes (profiles/*/traffic.json, produced by
    tools/tools_pmc.sh + tools/tools_traffic.py with the gfx950 FETCH_SIZE x2 correction); None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files or world != 1:
        return None
    t = json.load(open(files[-1]))
    return t.get("hbm_bytes_per_launch")


def _with_c_stdout_on_stderr(fn):
    """Run fn() with file descriptor 1 pointing at stderr, flushing C stdio before restoring it."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        fn()
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-group", action="store_true",
                    help="exercise the gloo rendezvous + RCCL compose path even with one rank (plumbing check)")
    args = ap.parse_args()

    import torch  # plumbing only: rendezvous / barrier / max-over-ranks; loaded first so ONE HIP runtime is used
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(1, torch.cuda.device_count()))        # barrier()'s torch.cuda.synchronize() must not touch GPU 0 from every rank
    dist = None
    grouped = world > 1 or args.force_group
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        _with_c_stdout_on_stderr(lambda: dist.init_process_group("gloo", rank=rank, world_size=world))

    from tracer_amd import abi, host
    from tracer_amd.device import Tracer, group_unique_id

    # plumbing check on boxes with fewer GPUs than ranks (no RCCL between ranks sharing a GPU): every other part of the
    # N-rank path -- rendezvous, stacked workload, tile ownership, reductions of the timings -- runs as usual
    no_rccl = os.environ.get("TRC_BENCH_NO_RCCL") == "1"
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    trc = Tracer(0 if no_rccl else local_rank)
    trc.upload_scene(scene.view)
    trc.set_camera(cam)
    trc.set_environment((0.0, 0.0, 0.0))
    FH = H * world                      # N stacked views
    trc.resize(W, FH)
    if grouped and not no_rccl:
        ids = [group_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)

        def init_comm():
            trc.group_init(ids[0], world, rank)
            trc.clear_accum()
            trc.group_reduce_accum(0)        # first collective: RCCL finishes its lazy set-up here
            trc.synchronize()
        # RCCL prints a version banner through C stdio on fd 1; keep stdout clean for the single JSON line
        _with_c_stdout_on_stderr(init_comm)

    def barrier():
        trc.synchronize()
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def step(collect_stats=False):
        if grouped:
            trc.clear_accum()            # non-owned tiles must be zero for the sum-compose
        trc.seed(SEED)
        trc.render(spp=SPP, max_depth=DEPTH, integrator=abi.INTEGRATOR_PATH, frame0=0, tile_rank=rank,
                   tile_nranks=world, collect_stats=collect_stats, view_height=H)
        if grouped and not no_rccl:
            trc.group_reduce_accum_async(0)   # overlaps with the next step's render (second accumulator + stream)

    # exact algorithmic work of ONE step on this rank (instrumented kernel, untimed)
    trc.reset_stats()
    step(collect_stats=True)
    trc.synchronize()
    st1 = trc.stats()
    own_pixels = st1.paths // SPP
    bytes_per_launch = algorithmic_bytes(st1, own_pixels)
    rays_per_launch = st1.rays

    for _ in range(args.warmup):
        step()
    barrier()
    trc.reset_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    st = trc.stats()
    assert st.rays == rays_per_launch * args.steps, "steps are not identical work"

    rays_total, dt_max = float(st.rays), dt
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_max = float(t.item())
        r = torch.tensor([rays_total], dtype=torch.float64)
        dist.all_reduce(r, op=dist.ReduceOp.SUM)
        rays_total = float(r.item())

    if rank == 0:
        kernel_ms = st.kernel_ms / max(1, st.launches)          # HIP events on the render stream
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        info = trc.device_info()
        line = {
            "metric": "Mrays/s at 1920x1080x64spp", "value": round(rays_total / dt_max / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_max / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RT_Metal Cornell box + 12 spheres (BASELINE config 2), 1920x1080x64spp, "
                                   "tracePath depth 8, 21 leaves / 41 BVH nodes" +
                                   (f"; {world} such views stacked into one 1920x{FH} frame, one view's worth of "
                                    f"tiles per GPU" if world > 1 else ""),
                       "integrator": "tracePath", "rays_per_step": int(rays_total / args.steps),
                       "paths_per_step": W * FH * SPP, "mpaths_per_s": round(W * FH * SPP * args.steps / dt_max / 1e6, 2),
                       "tiles": f"16x16 px, owner (tx+ty)%{world}", "device": info["name"],
                       "compose": f"ncclReduce(sum) of the {W}x{FH} RGBA32F frame to rank 0 on a second stream, overlapped with the next step" if grouped else "none"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic(world),
                         "kernel": "k_render", "kernel_ms": round(kernel_ms, 3),
                         "algorithmic_bytes_per_launch": int(bytes_per_launch),
                         "bytes_per_ray": round(bytes_per_launch / max(1, rays_per_launch), 1)},
        }
        if not args.no_cpu_baseline and world == 1:      # CPU baselines: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(scene, cam)
            line["cpu_rt_weekend"] = cpu_rt_weekend()
        print(json.dumps(line), flush=True)
    sys.stdout.flush()

    def teardown():
        if grouped:
            if not no_rccl:
                trc.group_finalize()
            dist.barrier()
            dist.destroy_process_group()
        trc.close()
    _with_c_stdout_on_stderr(teardown)
    os.dup2(2, 1)       # anything native code still prints at exit goes to stderr, after the JSON line


if __name__ == "__main__":
    main()

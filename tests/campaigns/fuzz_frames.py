#!/usr/bin/env python3
"""Differential campaign beyond tests/test_gpu_fuzz.py's small frames: generated scenes (its generator) at frame sizes, sample counts
and launch sequences that bring the adaptive machinery in -- cost-ordered launches, the split plan, persistent workgroups with
overflow stacks, coalesced 1-sample calls, progressive accumulation over several calls, the Sobol' sampler, one tile rank of several,
stacked views, equirectangular environment maps, random scheduling knobs and launch flags -- GPU against the oracle, bit for bit.
    python3 tests/campaigns/fuzz_frames.py <first seed> <last seed>"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("TRC_FUZZ_SEEDS", "1:2")
import test_gpu_fuzz as tf
from oracle import pyoracle as po
from tracer_amd import abi, host
from tracer_amd.device import Tracer

gpu = Tracer(0)
VOLUME = os.environ.get("TRC_FUZZ_VOLUME") == "1"
BIG = os.environ.get("TRC_FUZZ_BIG") == "1"        # TRC_FUZZ_BIG=1: 1200..1920 x 700..1080 frames, knobs on every case
bad = 0
a, b = int(sys.argv[1]), int(sys.argv[2])
for seed in range(a, b):
    rs = np.random.RandomState(50000 + seed)
    big = seed % 2 == 0
    sv, keep = tf.random_scene(rs, n_spheres=int(rs.randint(3, 20)), n_cubes=int(rs.randint(1, 6)),
                               n_tris=int(rs.randint(900, 4000)) if big else int(rs.randint(5, 60)))
    W, H = int(rs.randint(40, 330)), int(rs.randint(40, 210))
    if BIG: W, H = int(rs.randint(1200, 1921)), int(rs.randint(700, 1081))      # frames with more blocks than wavefront slots: dense kernels, persistent workgroups, split plans
    integ = int(rs.randint(3))
    calls = [int(rs.choice([1, 1, 2, 3, 8, 9, 16, 24, 33])) for _ in range(int(rs.randint(1, 6)))]
    if BIG: calls = [int(rs.choice([1, 3, 8, 9, 16])) for _ in range(int(rs.randint(1, 4)))]
    depth = int(rs.randint(1, 9))
    # every fifth case through the Sobol' sampler (tracePath / traceMIS), every seventh as one tile rank of 2..5, every ninth as stacked views
    sobol = seed % 5 == 4 and integ != 2 and not VOLUME
    nranks = int(rs.randint(2, 6)) if seed % 7 == 6 else 1
    trank = int(rs.randint(nranks))
    views = int(rs.randint(2, 4)) if seed % 9 == 8 else 1
    if views > 1: H = (H // views) * views
    vh = H // views
    look_from = rs.uniform(-150, 150, 3); look_from[2] = -170.0
    if seed % 4 == 3: look_from = rs.uniform(-35, 35, 3)            # a camera INSIDE the scene: rays start within boxes, media, next to surfaces
    cam = host.make_camera(tuple(look_from), tuple(rs.uniform(-10, 10, 3)), (0, 1, 0), float(rs.uniform(0.0, 3.0)), W / vh, math.radians(55), 170.0)
    grid = rs.rand(6, 7, 8).astype(np.float32) * (rs.rand(6, 7, 8) > 0.4)
    info = host.density_info(np.ascontiguousarray(grid), sigma_a=0.02, sigma_s=0.05, g=0.3)
    if VOLUME:          # TRC_FUZZ_VOLUME=1: traceVolume only, under hostile media -- grids of 1..12 cells a side (sparse, empty, spiked, negative, NaN, -0),
        integ = 2       # coefficients from 0 to large, forward and backward phase functions
        dims = tuple(int(x) for x in rs.randint(1, 13, 3))
        grid = (rs.rand(*dims) * (rs.rand(*dims) > rs.uniform(0.1, 0.95))).astype(np.float32)
        kind = seed % 6
        if kind == 1: grid[:] = 0
        if kind == 2: grid[tuple(rs.randint(0, d) for d in dims)] = 1e3
        if kind == 3: grid.reshape(-1)[rs.randint(0, grid.size, max(1, grid.size // 10))] = rs.choice(np.array([-1.0, -0.0, np.nan, np.inf], np.float32), max(1, grid.size // 10))
        sa, ss = [(0.02, 0.05), (0.0, 0.3), (0.5, 0.0), (2.0, 8.0), (1e-4, 1e-4), (0.0, 0.0)][int(rs.randint(6))]
        info = host.density_info(np.ascontiguousarray(grid), sigma_a=sa, sigma_s=ss, g=float(rs.uniform(-0.9, 0.9)))
    env = (0.3, 0.4, 0.6) if seed % 3 else (0.0, 0.0, 0.0)
    # every fourth case (offset 1) under an equirectangular environment map of odd sizes, 1 x 1 included (the open scenes let most rays out)
    envmap = rs.uniform(0.0, 3.0, (int(rs.randint(1, 40)), int(rs.randint(1, 70)), 3)).astype(np.float32) if seed % 4 == 1 else None
    gpu.set_environment_map(envmap); po.set_environment_map(envmap)
    # every third case (offset 2) under a random handful of the library's scheduling knobs and launch flags: none of them may change a pixel
    knobs = {}
    if seed % 3 == 2 or BIG:
        pool = {"no_lds_fit": [1], "stack_lds_levels": [1, 2, 3, 6, 12], "strip_len": [1, 2, 3, 5], "no_pwg": [1], "no_split": [1], "no_cost_filter": [1],
                "no_cold_probe": [1], "probe_spp": [8, 16], "no_plan_reuse": [1], "no_coalesce": [1], "no_dense": [1], "head_stages": [1, 2], "descend_min": [1, 3, 24, 64]}
        for k in rs.choice(sorted(pool), int(rs.randint(1, 5)), replace=False): knobs[str(k)] = int(rs.choice(pool[str(k)]))
    fixed = bool(seed % 11 == 10); small = [None, True, False][int(rs.randint(3))] if seed % 3 == 2 else None
    gpu.upload_scene(sv); gpu.set_camera(cam); gpu.set_environment(env); gpu.resize(W, H)
    for k, v in knobs.items(): gpu.debug_set(k, v)
    gpu.upload_density(info, np.ascontiguousarray(grid)); po.set_density(info, np.ascontiguousarray(grid))
    ok = True
    for rep in range(2):                     # the second pass runs on the first one's block costs (cost order, split plan)
        rng = host.fill_rng(70 + seed, W, H)
        gpu.upload_rng(rng); gpu.clear_accum(); gpu.reset_stats()
        f0 = 0
        for c in calls:
            gpu.render(spp=c, integrator=integ, max_depth=depth, frame0=f0, sobol=sobol, tile_rank=trank, tile_nranks=nranks, view_height=vh if views > 1 else 0,
                       fixed_order=fixed, small_blocks=small); f0 += c
        got, got_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
        if rep == 0:
            ref_rng = rng.copy()
            ref, rst = po.render(sv, cam, W, H, ref_rng, spp=sum(calls), integrator=integ, max_depth=depth, env=env, sobol=sobol,
                                  tile_rank=trank, tile_nranks=nranks, view_height=vh if views > 1 else 0)
        if not (np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and np.array_equal(got_rng, ref_rng) and st.rays == rst.rays):
            ok = False
            print(f"MISMATCH seed {seed} pass {rep}: {W}x{H} integrator {integ} calls {calls} depth {depth} big {big} sobol {sobol} rank {trank}/{nranks} views {views} knobs {knobs} fixed {fixed} small {small}: "
                  f"{int((got.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())} pixels, rays {st.rays} / {rst.rays}", flush=True)
    bad += not ok
    po.set_density(None, None); gpu.upload_density(None, None)
    for k in knobs: gpu.debug_set(k, 0)
    if envmap is not None: gpu.set_environment_map(None); po.set_environment_map(None)
print(f"seeds {a}..{b - 1}: {b - a - bad} passed, {bad} FAILED")

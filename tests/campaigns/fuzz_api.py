#!/usr/bin/env python3
"""API-sequence campaign: random programs over one context -- trc_render calls of 1..40 samples (the 1..7-sample ones are kept and coalesced by
the library), camera moves, environment changes, reseeding, clearing, RNG / accumulator uploads and downloads, tone mapping, stats, scene
swaps, resizes -- mirrored step by step on the oracle.  Every download along the way and the final accumulator, RNG texture and ray count must be
the oracle's, bit for bit: a kept launch must be flushed by exactly the calls that would observe or invalidate it, with the state it was
issued under.          python3 tests/campaigns/fuzz_api.py <a> <b>"""
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
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    rs = np.random.RandomState(120000 + seed)
    scenes = [tf.random_scene(rs, n_spheres=int(rs.randint(3, 12)), n_cubes=int(rs.randint(1, 4)),
                              n_tris=int(rs.randint(900, 2000)) if rs.rand() < 0.4 else int(rs.randint(5, 60))) for _ in range(2)]
    def new_cam(W, H):
        lf = rs.uniform(-150, 150, 3); lf[2] = -170.0
        return host.make_camera(tuple(lf), tuple(rs.uniform(-10, 10, 3)), (0, 1, 0), float(rs.uniform(0.0, 2.0)), W / H, math.radians(55), 170.0)
    W, H = int(rs.randint(24, 200)), int(rs.randint(24, 140))
    cur = 0; sv = scenes[0][0]
    cam = new_cam(W, H); env = (0.3, 0.4, 0.6)
    gpu.upload_scene(sv); gpu.set_camera(cam); gpu.set_environment(env); gpu.resize(W, H)
    s0 = int(rs.randint(1, 1 << 30)); gpu.seed(s0); gpu.clear_accum(); gpu.reset_stats()
    rng = host.fill_rng(s0, W, H); acc = np.zeros((H, W, 4), np.float32); rays = 0
    frame = 0; integ = int(rs.randint(2)); depth = int(rs.randint(2, 9))
    log = []; why = None
    def check(tag):
        global why
        g_acc, g_rng = gpu.download_accum(), gpu.download_rng()
        if not np.array_equal(g_acc.view(np.uint32), acc.view(np.uint32)): why = f"{tag}: accumulator, {int((g_acc.view(np.uint32) != acc.view(np.uint32)).any(axis=2).sum())} pixels"
        elif not np.array_equal(g_rng, rng): why = f"{tag}: rng texture"
    for step in range(int(rs.randint(6, 40))):
        op = rs.choice(["render"] * 8 + ["camera", "env", "seed", "clear", "check", "tonemap", "stats", "uprng", "upacc", "scene", "resize", "integ"])
        log.append(op)
        if op == "render":
            k = int(rs.choice([1, 1, 1, 2, 3, 5, 7, 8, 9, 16, 40])); log[-1] = f"render{k}"
            gpu.render(spp=k, integrator=integ, max_depth=depth, frame0=frame)
            _, st = po.render(sv, cam, W, H, rng, accum=acc, spp=k, integrator=integ, max_depth=depth, frame0=frame, env=env)
            rays += st.rays; frame += k
        elif op == "camera": cam = new_cam(W, H); gpu.set_camera(cam)
        elif op == "env": env = tuple(float(x) for x in rs.uniform(0, 1, 3)); gpu.set_environment(env)
        elif op == "seed": s = int(rs.randint(1, 1 << 30)); gpu.seed(s); rng = host.fill_rng(s, W, H)
        elif op == "clear": gpu.clear_accum(); acc[:] = 0; frame = 0
        elif op == "check": check(f"step {step}")
        elif op == "tonemap":
            got, e = gpu.tonemap(); want, e_ref = po.tonemap(acc)
            if e != e_ref or not np.array_equal(got, want): why = f"step {step}: tonemap"
        elif op == "stats":
            if gpu.stats().rays != rays: why = f"step {step}: rays {gpu.stats().rays} / {rays}"
        elif op == "uprng": rng = host.fill_rng(int(rs.randint(1, 1 << 30)), W, H); gpu.upload_rng(rng)
        elif op == "upacc": acc = rs.uniform(0, 2, (H, W, 4)).astype(np.float32); acc[..., 3] = 1.0; gpu.upload_accum(acc); frame = int(rs.randint(1, 50))
        elif op == "scene": cur ^= 1; sv = scenes[cur][0]; gpu.upload_scene(sv)
        elif op == "integ": integ = int(rs.randint(2)); depth = int(rs.randint(2, 9))
        elif op == "resize":
            W, H = int(rs.randint(24, 200)), int(rs.randint(24, 140)); cam = new_cam(W, H)
            gpu.resize(W, H); gpu.set_camera(cam)
            s = int(rs.randint(1, 1 << 30)); gpu.seed(s); gpu.clear_accum()
            rng = host.fill_rng(s, W, H); acc = np.zeros((H, W, 4), np.float32); frame = 0
        if why: break
    if not why: check("end")
    if not why and gpu.stats().rays != rays: why = f"end: rays {gpu.stats().rays} / {rays}"
    if why:
        bad += 1
        print(f"MISMATCH seed {seed}: {why}; program: {' '.join(log)}", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad} passed, {bad} FAILED")

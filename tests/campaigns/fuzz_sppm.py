#!/usr/bin/env python3
"""SPPM campaign beyond tests/test_gpu_fuzz.py::test_generated_scene_sppm (64x48, 3 frames in one call): generated scenes at canvas sizes
32..200 x 24..140, 1..11 frames split over 1..3 trc_sppm_frames calls (the camera pass of an odd frame runs a frame ahead on its own
stream WITHIN a call: the call boundaries move where that happens), GPU against the oracle -- accumulator, canvas RNG, photon and
camera records, hash grids, the Complex block -- bit for bit.      python3 tests/campaigns/fuzz_sppm.py <first seed> <last seed>"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("TRC_FUZZ_SEEDS", "1:2")
import test_gpu_fuzz as tf
from oracle import pyoracle as po
from tracer_amd import host
from tracer_amd.device import Tracer

gpu = Tracer(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    rs = np.random.RandomState(90000 + seed)
    sv, keep = tf.random_scene(rs, n_spheres=int(rs.randint(3, 14)), n_cubes=int(rs.randint(1, 5)),
                               n_tris=int(rs.randint(900, 2500)) if seed % 4 == 0 else int(rs.randint(5, 80)))
    W, H = int(rs.randint(32, 201)), int(rs.randint(24, 141))
    calls = [int(rs.randint(1, 5)) for _ in range(int(rs.randint(1, 4)))]
    if seed % 13 == 12: W, H = int(rs.randint(300, 500)), int(rs.randint(200, 320)); calls = [int(rs.randint(1, 8)) for _ in range(int(rs.randint(1, 4)))]
    inside = seed % 5 == 4
    eye = tuple(rs.uniform(-30, 30, 3)) if inside else (rs.uniform(-100, 100), rs.uniform(-60, 60), -160.0)
    cam = host.make_camera(eye, (0, 0, 0), (0, 1, 0), 0.0, W / H, math.radians(50), 160.0)
    env = tuple(float(x) for x in rs.uniform(0, 1, 3)) if seed % 3 == 1 else (0.0, 0.0, 0.0)          # every third case under a constant environment
    serial = int(seed % 7 == 3)                                                                       # knob: the camera pass without its frame of lead
    gpu.upload_scene(sv); gpu.set_camera(cam); gpu.set_environment(env); gpu.resize(W, H); gpu.debug_set("sppm_serial_camera", serial)
    gpu.seed(8 + seed); gpu.clear_accum(); gpu.sppm_init(40 + seed)
    for c in calls:
        gpu.sppm_frames(c)
    dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
    dacc, drng = gpu.download_accum(), gpu.download_rng()
    rng = host.fill_rng(8 + seed, W, H); acc = np.zeros((H, W, 4), np.float32)
    o = po.Sppm(W, H, 40 + seed); o.frames(sv, cam, rng, acc, sum(calls), env=env)
    ocam, opho, omark, ocount, ocx = o.download()
    why = []
    if not np.array_equal(drng, rng): why.append("canvas rng")
    if not np.array_equal(dacc.view(np.uint32), acc.view(np.uint32)): why.append("accumulator")
    if not (np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)): why.append("hash grids")
    for what, got, ref in (("photon", dpho, opho), ("camera", dcam, ocam)):
        for f in got.dtype.names:
            if f.startswith("_"): continue
            x, y = np.ascontiguousarray(got[f]), np.ascontiguousarray(ref[f])
            if bytes(memoryview(x)) != bytes(memoryview(y)): why.append(f"{what}.{f}")
    if dcx.frame_count != ocx.frame_count or dcx.totalPhotonSum != ocx.totalPhotonSum: why.append("complex")
    if why:
        bad += 1
        print(f"MISMATCH seed {seed}: {W}x{H} calls {calls} inside {inside} env {env} serial {serial}: {why}", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad} passed, {bad} FAILED")

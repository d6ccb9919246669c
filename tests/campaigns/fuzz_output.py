#!/usr/bin/env python3
"""Output stage campaign: random accumulators -- ordinary radiance, zeros, denormals, huge values, negatives, NaN, +-inf, in frames of
1 x 1 .. 400 x 300 -- through trc_upload_accum + trc_tonemap against oracle/pyoracle.tonemap (fragmentShader's auto-exposure from exact
fixed-point sums + ACES, Render.metal:29-75): the exposure's bits and every output byte.      python3 tests/campaigns/fuzz_output.py <a> <b>"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po
from tracer_amd import abi, host
from tracer_amd.device import Tracer

gpu = Tracer(0)
sc = host.HostScene(abi.SCENE_CORNELL)
gpu.upload_scene(sc.view)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
for seed in range(a, b):
    rs = np.random.RandomState(seed)
    W, H = int(rs.randint(1, 401)), int(rs.randint(1, 301))
    mode = seed % 6
    acc = rs.uniform(0, [1.0, 10.0, 0.01, 1e4, 1.0, 1e-30][mode], (H, W, 4)).astype(np.float32)
    if mode == 4: acc = np.exp(rs.normal(0, 6, (H, W, 4))).astype(np.float32)
    if mode == 5: acc = (acc * np.float32(1e-10)).astype(np.float32)           # denormals
    n = H * W * 4
    flat = acc.reshape(-1)
    if seed % 3 == 1:                          # sprinkle the nasties
        for v in (np.nan, -np.nan, np.inf, -np.inf, -1.0, -0.0, 3.4e38, 1e-45):
            flat[rs.randint(0, n, max(1, n // 200))] = np.float32(v)
    if seed % 10 == 9: flat[:] = 0
    acc[..., 3] = 1.0
    gpu.set_camera(host.prepare_camera(W, H)); gpu.resize(W, H); gpu.upload_accum(acc)
    got, e = gpu.tonemap()
    want, e_ref = po.tonemap(acc)
    if bits(e) != bits(e_ref) or not np.array_equal(got, want):
        bad += 1
        print(f"MISMATCH seed {seed}: {W}x{H} mode {mode}: exposure {e!r} / {e_ref!r}, {int((got != want).any(axis=2).sum())} pixels differ", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad} passed, {bad} FAILED")

#!/usr/bin/env python3
"""Throughput of config 2 against the number of samples fused into one launch (the reference launches 1 spp per
frame): the same 64 frames as 64 x 1, 16 x 4, 4 x 16 and 1 x 64 samples per trc_render."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
for spp, coalesce in ((1, False), (1, True), (4, False), (4, True), (16, True), (64, True)):
    t.debug_set("no_coalesce", 0 if coalesce else 1)
    for rep in range(2):                    # second pass: adaptive launch order warmed up
        t.seed(0x5EED0000); t.clear_accum(); t.reset_stats(); t.synchronize()
        t0 = time.perf_counter()
        for i in range(64 // spp):
            t.render(spp=spp, frame0=spp * i)
        t.synchronize(); wall = (time.perf_counter() - t0) * 1e3
        s = t.stats()
    print(f"{64 // spp:2d} calls x {spp:2d} spp{' (launched one by one: knob no_coalesce)' if not coalesce and spp < 8 else ''}: kernels {s.kernel_ms:7.2f} ms, wall {wall:7.2f} ms, {s.rays / wall / 1e3:7.1f} Mrays/s (wall)")

#!/usr/bin/env python3
"""config 5: SPPM, Cornell + spheres, 1920x1080, 512^2 photons per frame, 64 frames (for rocprofv3 --kernel-trace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H); t.seed(1); t.sppm_init(2)
t.sppm_frames(2); t.synchronize()
t0 = time.perf_counter(); t.sppm_frames(64); t.synchronize(); dt = time.perf_counter() - t0
print(f"SPPM 64 frames: {dt / 64 * 1e3:.3f} ms/frame")

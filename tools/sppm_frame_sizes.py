#!/usr/bin/env python3
"""SPPM frame time against the frame size (one frame per call: even / odd frames; odd frames re-run the camera pass).
A pass bound by its slowest wavefront would cost the same on a 64x36 frame; the camera pass did not (docs/HISTORY.md section 9)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
for W, H in ((64, 36), (256, 144), (960, 540), (1920, 1080)):
    t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.set_environment((0, 0, 0)); t.resize(W, H)
    t.seed(1); t.sppm_init(7); t.sppm_frames(4); t.synchronize()
    res = []
    for n in (1, 1, 1, 1):      # frames 4, 5, 6, 7: even, odd, even, odd -- one frame per call: camera beside its own photon pass
        t0 = time.time(); t.sppm_frames(1); t.synchronize(); res.append((time.time() - t0) * 1e3)
    print(W, H, " ".join(f"{r:.3f}" for r in res), "ms per single-frame call (even, odd, even, odd)")
    t.close()

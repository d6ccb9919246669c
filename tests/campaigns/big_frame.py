#!/usr/bin/env python3
"""One frame far beyond 1080p (default 7680 x 4320: 33 M pixels, 518 400 blocks) through the coalesced, strip and fused kernels against the oracle:
index arithmetic, launch lists and cost arrays at a size the suite never reaches.      python3 tests/campaigns/big_frame.py [W H]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pyoracle as po
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7680, 4320)
gpu = Tracer(0)
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
cam = host.prepare_camera(W, H)
gpu.upload_scene(sc.view); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
ok = True
for integ, calls in ((abi.INTEGRATOR_PATH, [1, 9]), (abi.INTEGRATOR_MIS, [3])):
    gpu.seed(77); gpu.clear_accum(); gpu.reset_stats()
    f0 = 0
    for c in calls:
        gpu.render(spp=c, integrator=integ, frame0=f0); f0 += c
    got, got_rng, st = gpu.download_accum(), gpu.download_rng(), gpu.stats()
    rng = host.fill_rng(77, W, H)
    ref, rst = po.render(sc.view, cam, W, H, rng, spp=sum(calls), integrator=integ)
    same = np.array_equal(got.view(np.uint32), ref.view(np.uint32)) and np.array_equal(got_rng, rng) and st.rays == rst.rays
    ok &= same
    print(f"{W}x{H} integrator {integ} calls {calls}: {'equal' if same else 'MISMATCH'} ({st.rays} rays, kernels {st.kernel_ms:.1f} ms)", flush=True)
# ... and the SPPM pass on the same canvas: 3 frames (visible points, bound, hash, refine) -- every record, both grids, the frame
gpu.seed(5); gpu.clear_accum(); gpu.sppm_init(6); gpu.sppm_frames(3)
dcam, dpho, dmark, dcount, dcx = gpu.sppm_download()
dacc, drng = gpu.download_accum(), gpu.download_rng()
rng = host.fill_rng(5, W, H); acc = np.zeros((H, W, 4), np.float32)
o = po.Sppm(W, H, 6); o.frames(sc.view, cam, rng, acc, 3)
ocam, opho, omark, ocount, ocx = o.download()
same = (np.array_equal(drng, rng) and np.array_equal(dacc.view(np.uint32), acc.view(np.uint32)) and np.array_equal(dcount, ocount) and np.array_equal(dmark, omark)
        and all(bytes(memoryview(np.ascontiguousarray(dpho[f]))) == bytes(memoryview(np.ascontiguousarray(opho[f]))) for f in ("flux", "normal", "position", "direction", "step", "active"))
        and all(bytes(memoryview(np.ascontiguousarray(dcam[f]))) == bytes(memoryview(np.ascontiguousarray(ocam[f]))) for f in ("ratio", "position", "direction", "valid", "alternative", "flux", "radius", "photonCount")))
ok &= same
print(f"{W}x{H} SPPM, 3 frames: {'equal' if same else 'MISMATCH'}", flush=True)
sys.exit(0 if ok else 1)

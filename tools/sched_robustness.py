#!/usr/bin/env python3
"""The feedback scheduler (launch order, split plan, head + rest, plan reuse) outside the two scenes it was tuned on:
  (a) a camera that moves every 4 launches (config 2 and config 4 at 32 spp): kernel ms launch by launch -- a moved camera
      invalidates the block costs, so its first launch runs as an 8-sample head + the rest;
  (b) a pbrt-loaded scene (tests' Cornell-like file: spheres, checkerboard cylinder + disk, a PLY wedge) rendered at 1080p,
      tracePath and traceMIS: convergence launch by launch, against the same launches in fixed row-major order.
    python3 tools/sched_robustness.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import workloads as wlmod
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080


def launch(t, seed, **kw):
    t.seed(seed); t.clear_accum(); t.reset_stats(); t.render(**kw); t.synchronize()
    return t.stats().kernel_ms


print("(a) camera moving every 4 launches (orbit in steps of 2 degrees); kernel ms per launch, * = first launch of a view")
import math
for config, spp in (("2", 64), ("4", 32)):
    wl = wlmod.make(config)
    with Tracer(0) as t:
        wlmod.setup(t, wl)
        out = []
        for k in range(16):
            if k % 4 == 0:
                a = math.radians(2.0 * (k // 4))
                eye = (278 + 800 * math.sin(a), 278, 278 - 1078 * math.cos(a))       # the reference's camera distance, orbiting the box centre
                t.set_camera(host.make_camera(eye, (278, 278, 278), (0, 1, 0), 0.0, W / H, math.radians(40), 10.0))
            out.append(("*" if k % 4 == 0 else "") + f"{launch(t, 100 + k, spp=spp, integrator=wl['integrator']):.2f}")
        fixed = []
        for k in range(4):
            fixed.append(f"{launch(t, 200 + k, spp=spp, integrator=wl['integrator'], fixed_order=True, small_blocks=False):.2f}")
    print(f"config {config}, {spp} spp: " + " ".join(out) + "   | last view in fixed row-major order: " + " ".join(fixed))

print("(b) pbrt-loaded scene at 1080p, 32 spp; kernel ms per launch (new seed each)")
from test_pbrt_scene import CORNELL, WEDGE_PLY
with tempfile.TemporaryDirectory() as d:
    open(os.path.join(d, "c.pbrt"), "w").write(CORNELL); open(os.path.join(d, "wedge.ply"), "w").write(WEDGE_PLY)
    scene, cam, info, shapes = host.HostScene.from_pbrt(os.path.join(d, "c.pbrt"))
cam = host.make_camera((cam.lookFrom.x, cam.lookFrom.y, cam.lookFrom.z), (cam.lookAt.x, cam.lookAt.y, cam.lookAt.z),
                       (cam.viewUp.x, cam.viewUp.y, cam.viewUp.z), 0.0, W / H, cam.vfov, cam.focus_dist)     # the file's view at 16:9
for integ, name in ((abi.INTEGRATOR_PATH, "tracePath"), (abi.INTEGRATOR_MIS, "traceMIS")):
    with Tracer(0) as t:
        t.upload_scene(scene.view); t.set_camera(cam); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        ad = [f"{launch(t, 300 + k, spp=32, integrator=integ):.2f}" for k in range(12)]
        fx = [f"{launch(t, 300 + k, spp=32, integrator=integ, fixed_order=True, small_blocks=False):.2f}" for k in range(3)]
    print(f"{name}: adaptive " + " ".join(ad) + "   | fixed row-major order: " + " ".join(fx))

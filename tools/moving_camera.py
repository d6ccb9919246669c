#!/usr/bin/env python3
"""A camera that moves EVERY frame (the reference resets its accumulation on every drag, AAPLRenderer.mm:1112-1132): one launch per
view, the view orbiting in steps of 0.5 degrees; kernel ms per launch against the settled launch of a camera at rest, under the
policies of trc_set_camera (knob camera_policy):
  1  the recorded costs are forgotten; the launch runs as an 8-sample head ordered by the previous view's launch order + the rest
     planned from the head (shipped up to round 5)
  2  the previous view's filtered costs order and plan the launch as if the camera had not moved (no head)
  3  as 2, with the previous launch's RAW durations as the costs (the filter follows a block that got heavier by 1.6 % per launch)
  0  the shipped rule: 3 for a camera that moved a little (<= 5 degrees, <= 5 % of the scene's diagonal), 1 for one that jumped
    python3 tools/moving_camera.py [--config 2,3,4] [--frames 24] [--step 0.5]"""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd import host
from tracer_amd.device import Tracer
W, H = 1920, 1080
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2,3,4"); ap.add_argument("--frames", type=int, default=24); ap.add_argument("--step", type=float, default=0.5)
a = ap.parse_args()


def view(t, deg):
    r = math.radians(deg)
    eye = (278 + 1078 * math.sin(r), 278, 278 - 1078 * math.cos(r))           # the reference's camera distance, orbiting the box centre
    t.set_camera(host.make_camera(eye, (278, 278, 278), (0, 1, 0), 0.0, W / H, math.radians(45), 10.0))


def launch(t, seed, **kw):
    t.seed(seed); t.clear_accum(); t.reset_stats(); t.render(**kw); t.synchronize()
    return t.stats().kernel_ms


for config in a.config.split(","):
    wl = wlmod.make(config)
    spp = {"2": 64, "3": 32, "4": 32}[config]
    print(f"{wl['what']}, {spp} spp per view, {a.frames} views {a.step} degrees apart (kernel ms)")
    for policy in (1, 2, 3, 0):
        with Tracer(0) as t:
            wlmod.setup(t, wl)
            t.debug_set("camera_policy", policy)
            view(t, 0.0)
            rest = [launch(t, 10 + k, spp=spp, integrator=wl["integrator"]) for k in range(10)]
            settled = sum(rest[-4:]) / 4
            moving = []
            for k in range(a.frames):
                view(t, a.step * (k + 1))
                moving.append(launch(t, 100 + k, spp=spp, integrator=wl["integrator"]))
            tail = moving[4:]
            print(f"  policy {policy}: settled {settled:.2f}   moving mean {sum(tail) / len(tail):.2f} ({sum(tail) / len(tail) / settled:.3f} x)  "
                  f"worst {max(tail):.2f}   first views " + " ".join(f"{m:.2f}" for m in moving[:6]), flush=True)

#!/usr/bin/env python3
"""How fast is Scene::hit alone?  Primary + two generations of cosine-scattered secondary rays through
trc_trace_rays (k_trace); run under `rocprofv3 --kernel-trace --stats` to read the k_trace durations."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
from tracer_amd.dtypes import make_rays
W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else "spheres"
if kind == "spheres":
    sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
else:
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4))
cam = host.prepare_camera(W, H)
t = Tracer(0); t.upload_scene(sc.view)
ys, xs = np.mgrid[0:H, 0:W]
# tile order (16x16) so that a wavefront holds an 8x8 pixel block like the render kernel
ty, tx = ys // 16, xs // 16
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), ((xs % 16) // 8).ravel(), ((ys % 16) // 8).ravel(), tx.ravel(), ty.ravel()))
u = (xs.ravel()[order].astype(np.float32) / np.float32(W)); v = (ys.ravel()[order].astype(np.float32) / np.float32(H))
f = lambda a: np.array([a.x, a.y, a.z], dtype=np.float32)
sample = f(cam.cornerLowLeft)[None] + f(cam.horizontal)[None] * u[:, None] + f(cam.vertical)[None] * v[:, None]
o = np.repeat(f(cam.lookFrom)[None], len(u), 0)
rays = make_rays(o, (sample - o).astype(np.float32))
rng = np.random.default_rng(1)
for gen in range(3):
    hits = t.trace_rays(rays); hits = t.trace_rays(rays); hits = t.trace_rays(rays)
    ok = hits["hit"] != 0
    print(f"gen {gen}: {len(rays)} rays, {ok.sum()} hit, descend/ray {hits['n_descend'].mean():.2f} leaf/ray {hits['n_leaf'].mean():.2f}")
    steps = (hits["n_descend"] + hits["n_leaf"]).astype(np.int64); m = len(steps) // 64 * 64
    wmax = steps[:m].reshape(-1, 64).max(1)
    print(f"   per-wave: mean steps {steps.mean():.1f}, mean of wave max {wmax.mean():.1f}, max {wmax.max()}, p99 of wave max {np.percentile(wmax, 99):.0f}; SIMD eff {steps.mean()/wmax.mean():.2f}")
    hits = hits[ok]
    n = hits["sn"].astype(np.float32)
    r1, r2 = rng.random(len(n), dtype=np.float32), rng.random(len(n), dtype=np.float32)
    phi = 2 * np.pi * r1; rr = np.sqrt(r2)
    loc = np.stack([rr * np.cos(phi), rr * np.sin(phi), np.sqrt(1 - r2)], 1).astype(np.float32)
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tt = np.cross(n, a); tt /= np.linalg.norm(tt, axis=1, keepdims=True); bb = np.cross(n, tt)
    d = (loc[:, :1] * tt + loc[:, 1:2] * bb + loc[:, 2:3] * n).astype(np.float32)
    rays = make_rays((hits["p"] + 1e-3 * n).astype(np.float32), d)

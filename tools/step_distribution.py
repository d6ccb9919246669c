"""Distribution of the box steps per ray (exact counters of the Scene::hit hook) through the host SAH tree and the
device-built tree on the 1.04 M-triangle scene: mean, tail, and what a 64-wide wavefront pays (the maximum)."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
from tracer_amd.dtypes import make_rays
mesh = host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)
sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
t = Tracer(0)
rng = np.random.default_rng(3)
N = 1 << 20
O = rng.uniform(20, 535, size=(N, 3)).astype(np.float32)
D = rng.normal(size=(N, 3)).astype(np.float32); D /= np.linalg.norm(D, axis=1, keepdims=True)
rays = make_rays(O, D)
def stats(label):
    h = t.trace_rays(rays)
    d = h["n_descend"].astype(np.float64)
    g = d.reshape(-1, 64)
    print(f"{label}: mean {d.mean():.2f}  p99 {np.percentile(d, 99):.0f}  max {d.max():.0f}  mean of 64-ray maxima {g.max(axis=1).mean():.1f}  (lane efficiency {d.mean() / g.max(axis=1).mean():.3f})")
t.upload_scene(sc.view); stats("SAH tree   ")
t.upload_scene_lbvh(sc.leaves_view()); stats("device tree")

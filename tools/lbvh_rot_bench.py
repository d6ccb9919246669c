"""Render time and exact per-ray traversal counters of the 1.04 M-triangle scene through the host SAH tree and through
the tree trc_upload_scene_lbvh builds on the device (Morton order + two sweeps of tree rotations)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
mesh = host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)
sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
t = Tracer(0); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
def run(label):
    out = []
    for rep in range(3):
        t.seed(1 + rep); t.clear_accum(); t.reset_stats(); t.render(spp=16); t.synchronize(); s = t.stats(); out.append(s.kernel_ms)
    t.seed(1); t.clear_accum(); t.reset_stats(); t.render(spp=2, collect_stats=True); t.synchronize(); c = t.stats()
    print(f"{label}: 16 spp {min(out):.2f} ms, {s.rays / min(out) / 1e3:.1f} Mrays/s | per ray: descend {c.n_descend / c.rays:.2f} return {c.n_return / c.rays:.2f} tri tests {c.n_leaf_triangle / c.rays:.2f}")
t.upload_scene(sc.view); run("SAH tree ")
t.upload_scene_lbvh(sc.leaves_view()); n, h, ms = t.lbvh_info()
run(f"device tree (build {ms:.2f} ms, height {h})")

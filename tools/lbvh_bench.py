#!/usr/bin/env python3
"""LBVH vs host SAH on config 4's scene (teapot.obj x 64 = 1.0 M triangles): build / upload times and render throughput
through either tree."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
mesh = host.Mesh.golden("teapot").replicate(8, 80.0)
t0 = time.time(); sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh); host_s = time.time() - t0
t = Tracer(0); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
def run(label):
    t.seed(1); t.render(spp=1); t.synchronize()
    t.seed(1); t.reset_stats(); t.render(spp=16); t.synchronize(); s = t.stats()
    print(f"{label}: 16 spp {s.kernel_ms:.2f} ms, {s.rays / s.kernel_ms / 1e3:.1f} Mrays/s")
t0 = time.time(); t.upload_scene(sc.view); up_s = time.time() - t0
print(f"host scene + SAH build {host_s:.3f} s (all host cores); trc_upload_scene (validate + repack + copy) {up_s:.3f} s")
run("SAH tree ")
for i in range(3):
    t0 = time.time(); t.upload_scene_lbvh(sc.leaves_view()); up_s = time.time() - t0
    n, h, ms = t.lbvh_info()
    print(f"trc_upload_scene_lbvh {up_s:.3f} s wall, GPU build {ms:.2f} ms, {n} nodes, height {h}")
run("LBVH tree")

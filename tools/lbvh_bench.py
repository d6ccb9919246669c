#!/usr/bin/env python3
"""LBVH and the device-built SAH tree vs the host SAH tree on config 4's scene (teapot.obj x 64 = 1.0 M triangles): build / upload times and render throughput
through either tree.  Both trees are measured the same way: a first launch (head + rest: trc_render's cold path), then SETTLE
launches with a new seed each so that order and split plan settle, then the mean of K timed launches at 32 spp."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H, SPP, SETTLE, K = 1920, 1080, 32, 8, 6
mesh = host.Mesh.golden("teapot").replicate(8, 80.0)
t0 = time.time(); sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh); host_s = time.time() - t0
t = Tracer(0); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)


def run(label):
    t.seed(1); t.clear_accum(); t.reset_stats(); t.render(spp=SPP); t.synchronize()
    first = t.stats().kernel_ms
    for i in range(SETTLE):
        t.seed(2 + i); t.clear_accum(); t.render(spp=SPP)
    t.synchronize(); t.reset_stats()
    for i in range(K):
        t.seed(100 + i); t.clear_accum(); t.render(spp=SPP)
    t.synchronize(); s = t.stats()
    print(f"{label}: {SPP} spp first launch {first:.2f} ms, settled {s.kernel_ms / K:.2f} ms per launch, {s.rays / s.kernel_ms / 1e3:.1f} Mrays/s")
    return s.kernel_ms / K


t0 = time.time(); t.upload_scene(sc.view); up_s = time.time() - t0
print(f"host scene + SAH build {host_s:.3f} s (all host cores); trc_upload_scene (validate + repack + copy) {up_s:.3f} s")
sah = run("SAH tree ")
for i in range(3):
    t0 = time.time(); t.upload_scene_lbvh(sc.leaves_view()); up_s = time.time() - t0
    n, h, ms = t.lbvh_info()
    print(f"trc_upload_scene_lbvh {up_s:.3f} s wall, GPU build {ms:.2f} ms, {n} nodes, height {h}")
lb = run("LBVH tree")
print(f"LBVH / SAH render time: {lb / sah:.3f}")
for i in range(3):
    t0 = time.time(); t.upload_scene_sah(sc.leaves_view()); up_s = time.time() - t0
    n, h, ms = t.lbvh_info()
    print(f"trc_upload_scene_sah {up_s:.3f} s wall, GPU build {ms:.2f} ms, {n} nodes, height {h}")
ds = run("device SAH tree")
print(f"device SAH / host SAH render time: {ds / sah:.3f}")

#!/usr/bin/env python3
"""config 4 stand-in (1.04 M triangles, tracePath) as a small repeatable workload for rocprofv3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
mesh = host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)
sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
t.seed(1); t.reset_stats(); t.render(spp=16, collect_stats=True); t.synchronize(); s1 = t.stats()
leaf = 48 * s1.n_leaf_triangle + 60 * s1.n_hit_triangle + 32 * s1.n_leaf_square + 100 * s1.n_leaf_cube + 128 * s1.n_hit_cube
bytes16 = 88 * s1.n_descend + 24 * s1.n_return + leaf + 64 * s1.shaded + 64 * W * H
for i in range(3):
    t.seed(1); t.reset_stats(); t.render(spp=16); t.synchronize()
s = t.stats()
print(f"config4 16spp: {s.kernel_ms:.2f} ms {s.rays/s.kernel_ms/1e3:.1f} Mrays/s; algorithmic {bytes16/s1.rays:.0f} B/ray -> "
      f"{bytes16/s.kernel_ms/1e6:.1f} GB/s = {bytes16/s.kernel_ms/1e6/8000:.3f} of 8 TB/s; descend/ray {s1.n_descend/s1.rays:.1f} "
      f"tri tests/ray {s1.n_leaf_triangle/s1.rays:.2f}")

#!/usr/bin/env python3
"""Quick Mrays/s of the mesh scenes (BASELINE configs 3 / 4 stand-ins) -- exploration tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer

W, H = 1920, 1080
cases = [("ball 46.8k tris, MIS", host.Mesh.ball(153, 153, 0.08), abi.INTEGRATOR_MIS, 16),
         ("ball 46.8k tris, PATH", host.Mesh.ball(153, 153, 0.08), abi.INTEGRATOR_PATH, 16),
         ("grid 12x12 of 7.2k-tri balls = 1.04M tris, PATH", host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4), abi.INTEGRATOR_PATH, 16)]
t = Tracer(0)
for name, mesh, integ, spp in cases:
    t0 = time.time()
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    tb = time.time() - t0
    t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H); t.seed(1)
    t.render(spp=1, integrator=integ); t.synchronize()
    t.reset_stats()
    t.render(spp=spp, integrator=integ, collect_stats=True); t.synchronize()
    s1 = t.stats()
    t.reset_stats(); t.seed(1)
    t.render(spp=spp, integrator=integ); t.synchronize()
    s = t.stats()
    print(f"{name}: tris {mesh.n_triangles} nodes {sc.view.n_bvh} depth {sc.tree_depth()} build {tb:.2f}s | "
          f"{s.rays/s.kernel_ms/1e3:.1f} Mrays/s ({s.kernel_ms:.1f} ms, {s.rays} rays, {s.rays/s.paths:.2f} rays/path) "
          f"descend/ray {s1.n_descend/s1.rays:.1f} tri tests/ray {s1.n_leaf_triangle/s1.rays:.2f}")
    for k, (l, w, u) in t.debug_profile().items():
        if w: print(f"    {k:10s} util {u:.3f} waves {w}")

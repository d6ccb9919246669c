#!/usr/bin/env python3
"""Throughput across the memory hierarchy's boundaries: the same Cornell box with a mesh of 0 ... 4 M triangles, both integrators,
1920x1080, 32 spp per launch.  Where the scene lives decides what a traversal step costs:
  LDS          the whole tree staged per workgroup (Cornell + 12 spheres: 3 KB)
  L2           4 MiB per XCD
  MALL         the 256 MiB Infinity Cache (config 4's 1.0 M triangles: 242 MB)
  HBM          beyond it (teapot.obj x 256 = 4.0 M triangles: ~0.97 GB) -- the operating point where "HBM GB/s" means HBM
Scenes are built on the device (trc_upload_scene_device: SAH tree + one leaf per triangle), so the 4 M-triangle row costs
milliseconds of build, not seconds.
    python3 tools/size_sweep.py [--spp 32] [--steps 3] [--k 0,1,2,4,8,16] [--integrators path,mis]
    python3 tools/size_sweep.py --k 16 --integrators path --json      one point, one JSON line (the PMC passes' workload)"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd import abi, host
from tracer_amd.device import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--spp", type=int, default=32); ap.add_argument("--steps", type=int, default=3); ap.add_argument("--settle", type=int, default=4)
ap.add_argument("--k", default="0,1,2,4,8,16"); ap.add_argument("--integrators", default="path,mis")
ap.add_argument("--mesh", default="teapot"); ap.add_argument("--json", action="store_true")
a = ap.parse_args()
W, H = wlmod.W, wlmod.H
INTEG = {"path": abi.INTEGRATOR_PATH, "mis": abi.INTEGRATOR_MIS}
t = Tracer(0)
if not a.json:
    print(f"# 1920x1080, {a.spp} spp per launch, settled ({a.settle} launches), mean of {a.steps}; mesh = {a.mesh}.obj on a k x k grid; "
          f"scene bytes = 64 B fat nodes + 48 B positions + 64 B attributes per triangle")
    print("# k  triangles  scene_MB  lives_in  integrator  kernel_ms  Mrays/s  rays/launch  build_ms")
for k in [int(x) for x in a.k.split(",")]:
    if k == 0:
        scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
        t.upload_scene(scene.view); build_ms = 0.0
    else:
        mesh = host.Mesh.golden(a.mesh)
        if k > 1:
            mesh = mesh.replicate(k, 80.0)
        scene = host.HostScene(abi.SCENE_CORNELL_MESH, mesh, analytic_leaves_only=True)
        t0 = time.perf_counter()
        t.upload_scene_device(scene.view, abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES); t.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
    tris = scene.view.n_index // 3
    n_leaves = tris + 21 if k == 0 else tris + scene.view.n_bvh
    mb = (64 * max(n_leaves - 1, 1) + 112 * tris) / 1e6
    where = "LDS" if k == 0 else "L2" if mb < 24 else "MALL" if mb < 256 else "HBM"
    t.set_camera(host.prepare_camera(W, H)); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
    for name in a.integrators.split(","):
        integ = INTEG[name]
        for i in range(a.settle):
            t.seed(0x5EED0000 + i); t.clear_accum(); t.render(spp=a.spp, integrator=integ)
        t.synchronize(); t.reset_stats()
        for i in range(a.steps):
            t.seed(0x5EED0100 + i); t.clear_accum(); t.render(spp=a.spp, integrator=integ)
        t.synchronize()
        s = t.stats()
        ms = s.kernel_ms / a.steps
        row = {"k": k, "triangles": tris, "scene_mb": round(mb, 1), "lives_in": where, "integrator": name, "spp": a.spp,
               "kernel_ms": round(ms, 3), "mrays_per_s": round(s.rays / a.steps / ms / 1e3, 1), "rays_per_launch": int(s.rays // a.steps),
               "build_ms": round(build_ms, 1)}
        if a.json:
            print(json.dumps(row), flush=True)
        else:
            print(f"{k:3d} {tris:9d} {mb:9.1f}  {where:5s} {name:5s} {ms:9.3f} {row['mrays_per_s']:9.1f} {row['rays_per_launch']:12d} {build_ms:8.1f}", flush=True)
t.close()

#!/usr/bin/env python3
"""Throughput of BASELINE configs 3, 4 (mesh scenes, 256 spp) and 5 (SPPM, 64 frames) at 1920x1080 on one GPU.
The asset files of the reference do not travel to the GPU box; procedural stand-ins of the same size are used:
config 3 = Cornell + a 46.8k-triangle ball (coatball.obj has 46.8k triangles), traceMIS;
config 4 = Cornell + a 12x12 grid of balls = 1.04 M triangles, tracePath."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer

W, H = 1920, 1080
t = Tracer(0)
cam = host.prepare_camera(W, H)
out = []
for name, mesh, integ, spp in [
        ("config3 coatball stand-in 46.8k tris, traceMIS, 256spp", host.Mesh.ball(153, 153, 0.08), abi.INTEGRATOR_MIS, 256),
        ("config4 1.04M tris, tracePath, 256spp", host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4), abi.INTEGRATOR_PATH, 256)]:
    t0 = time.time()
    sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
    build_s = time.time() - t0
    t.upload_scene(sc.view); t.set_camera(cam); t.resize(W, H); t.seed(1)
    t.render(spp=1, integrator=integ); t.synchronize(); t.seed(1); t.reset_stats()
    t.render(spp=spp, integrator=integ); t.synchronize()
    s = t.stats()
    out.append({"config": name, "triangles": mesh.n_triangles, "bvh_nodes": sc.view.n_bvh, "depth": sc.tree_depth(),
                "host_bvh_build_s": round(build_s, 3), "kernel_ms": round(s.kernel_ms, 2), "rays": s.rays,
                "mrays_per_s": round(s.rays / s.kernel_ms / 1e3, 1), "mpaths_per_s": round(s.paths / s.kernel_ms / 1e3, 1)})
# traceVolume: cloud container (procedural 100x100x40 grid) + the config-3 mesh as glass filled with the
# homogeneous medium, 64 spp
sc = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.ball(153, 153, 0.08))
cloud = host.make_cloud()
t.upload_scene(sc.view); t.upload_density(host.density_info(cloud), cloud); t.set_camera(cam); t.resize(W, H)
for name, integ in (("traceVolume", abi.INTEGRATOR_VOLUME), ("traceMIS on the same scene", abi.INTEGRATOR_MIS)):
    t.seed(1); t.render(spp=1, integrator=integ); t.synchronize(); t.seed(1); t.reset_stats()
    t.render(spp=64, integrator=integ); t.synchronize()
    s = t.stats()
    out.append({"config": f"volume scene, {name}, 64spp", "kernel_ms": round(s.kernel_ms, 2), "rays": s.rays,
                "mrays_per_s": round(s.rays / s.kernel_ms / 1e3, 1), "mpaths_per_s": round(s.paths / s.kernel_ms / 1e3, 1)})
t.upload_density(None, None)
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
t.upload_scene(sc.view); t.set_camera(cam); t.resize(W, H); t.seed(1); t.sppm_init(2)
t.sppm_frames(1); t.synchronize(); t.reset_stats()
t0 = time.time(); t.sppm_frames(64); t.synchronize(); dt = time.time() - t0
s = t.stats()
out.append({"config": "config5 SPPM, Cornell+spheres, 512^2 photons/frame, 64 frames", "wall_ms_per_frame": round(dt / 64 * 1e3, 3),
            "rays": s.rays, "mrays_per_s": round(s.rays / dt / 1e6, 1)})
for o in out: print(json.dumps(o))

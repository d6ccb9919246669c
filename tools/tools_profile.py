#!/usr/bin/env python3
"""Divergence + cycle attribution of the instrumented render kernel: lanes vs wavefronts per code site and the
share of wavefront-resident cycles spent in it.  usage: tools_profile.py [spheres|mesh|mesh1m] [path|mis]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else "spheres"
integ = 1 if (len(sys.argv) > 2 and sys.argv[2] == "mis") else 0
if kind == "spheres": sc, spp = host.HostScene(abi.SCENE_CORNELL_SPHERES), 64
elif kind == "mesh": sc, spp = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(152, 154, 1.0)), 16
else: sc, spp = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)), 16
t = Tracer(0, hooks=True); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H); t.seed(0x5EED0000)
t.reset_stats(); t.render(spp=spp, integrator=integ, collect_stats=True); t.synchronize()
st = t.stats(); prof = t.debug_profile(); total = prof["loop"][3]
print(f"# {kind} integrator {integ} {spp} spp: instrumented kernel {st.kernel_ms:.1f} ms, {st.rays} rays")
print("# site        lanes      wave-executions  lanes/(64*exec)   cycles/exec  share of loop cycles")
for k, (l, w, u, c) in prof.items():
    print(f"{k:10s} {l:12d} {w:12d}   {u:.3f}   {c / max(w, 1):9.0f}   {c / max(total, 1):.3f}")

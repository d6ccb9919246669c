import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
mesh = host.Mesh.ball(60, 60, 0.08).replicate(12, 2.4)
sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh)
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
t.seed(1); t.render(spp=1); t.synchronize()
for spp in (64, 32, 16):
    t.seed(1)
    print(f"progressive, {spp} spp per launch:")
    for i in range(256 // spp):
        t.reset_stats(); t.render(spp=spp, frame0=spp*i); t.synchronize(); s = t.stats()
        print(f"  frames {spp*i:3d}..{spp*i+spp-1:3d}: {s.kernel_ms:7.1f} ms  {s.rays/s.kernel_ms/1e3:7.1f} Mrays/s")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W,H=1920,1080
sc=host.HostScene(abi.SCENE_CORNELL_SPHERES); cam=host.prepare_camera(W,H)
t=Tracer(0); t.upload_scene(sc.view); t.set_camera(cam); t.resize(W,H); t.seed(0x5EED0000)
t.reset_stats(); t.render(spp=64); t.synchronize()
s=t.stats(); print("kernel ms", s.kernel_ms, "rays", s.rays)
names=["T","L","R","M"]
prof=list(t.debug_profile().values())
tot=0
for n,(l,w,u) in zip(names,prof):
    print(f"{n:8s} lanes {l:12d} batches {w:10d} avg {l/max(1,w):5.1f}"); tot+=w
print("total batches", tot, "per wave", tot/ (1024*2))

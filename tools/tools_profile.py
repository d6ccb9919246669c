import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W,H=1920,1080
sc=host.HostScene(abi.SCENE_CORNELL_SPHERES); cam=host.prepare_camera(W,H)
t=Tracer(0); t.upload_scene(sc.view); t.set_camera(cam); t.resize(W,H); t.seed(0x5EED0000)
t.reset_stats(); t.render(spp=64, collect_stats=True); t.synchronize()
for k,(l,w,u) in t.debug_profile().items(): print(f"{k:10s} lanes {l:12d} waves {w:11d} util {u:.3f}")

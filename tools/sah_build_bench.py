#!/usr/bin/env python3
"""Device SAH build (trc_upload_scene_sah) of config 4's scene (teapot.obj x 64 = 1.0 M triangles): GPU ms of K builds and
the host builder beside it.  Under `rocprofv3 --kernel-trace --stats` the k_sah_* / k_lbvh_* rows are the per-kernel split."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mesh = host.Mesh.golden("teapot").replicate(8, 80.0)
t0 = time.time(); sc = host.HostScene(abi.SCENE_CORNELL_MESH, mesh); host_s = time.time() - t0
t = Tracer(0)
print(f"{sc.n_leaves} leaves; host scene + SAH build {host_s * 1e3:.0f} ms (all host cores)")
for i in range(K):
    t0 = time.time(); t.upload_scene_sah(sc.leaves_view()); wall = time.time() - t0
    n, h, ms = t.lbvh_info()
    print(f"trc_upload_scene_sah: {wall * 1e3:.1f} ms wall (leaf upload + triangle repack included), GPU build {ms:.2f} ms, depth {h}")
# the whole way from a mesh to a scene on the GPU, host work included: the host that does everything / the host that hands over
# its analytic leaves and the mesh (trc_host_scene_create_leaves + trc_upload_scene_device)
for i in range(3):
    t0 = time.time(); full = host.HostScene(abi.SCENE_CORNELL_MESH, mesh); t1 = time.time(); t.upload_scene(full.view); t.synchronize(); t2 = time.time()
    lean = host.HostScene(abi.SCENE_CORNELL_MESH, mesh, analytic_leaves_only=True); t3 = time.time()
    t.upload_scene_device(lean.view, abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES); t.synchronize(); t4 = time.time()
    print(f"mesh -> scene on the GPU: host leaves + host tree {1e3 * (t1 - t0):.0f} ms + trc_upload_scene {1e3 * (t2 - t1):.0f} ms = {1e3 * (t2 - t0):.0f} ms;  "
          f"analytic leaves {1e3 * (t3 - t2):.0f} ms + trc_upload_scene_device {1e3 * (t4 - t3):.0f} ms = {1e3 * (t4 - t2):.0f} ms (GPU build {t.lbvh_info()[2]:.2f} ms)")
for i in range(2):
    t0 = time.time(); t.upload_scene_lbvh(sc.leaves_view()); wall = time.time() - t0
    n, h, ms = t.lbvh_info()
    print(f"trc_upload_scene_lbvh: {wall * 1e3:.1f} ms wall, GPU build {ms:.2f} ms, depth {h}")

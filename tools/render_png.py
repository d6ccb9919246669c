#!/usr/bin/env python3
"""Render a frame on the GPU and write a tone-mapped PNG (developer eyeballing aid, not part of the path).
usage: render_png.py <spheres|mesh|volume> <path|mis|volume> <spp> <out.png> [W H]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
kind, integ, spp, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
W, H = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (640, 360)
integ = {"path": 0, "mis": 1, "volume": 2}[integ]
if kind == "spheres": sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
elif kind == "mesh": sc = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.ball(60, 60, 0.08))
else: sc = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.ball(60, 60, 0.08))
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H); t.seed(7)
if kind == "volume":
    cloud = host.make_cloud(); t.upload_density(host.density_info(cloud), cloud)
t.render(spp=spp, integrator=integ); t.synchronize()
img = t.download_accum()[..., :3]
img = img / (1 + img)                      # Reinhard
img = np.clip(img, 0, 1) ** (1 / 2.2)
from PIL import Image
Image.fromarray((img[::-1] * 255 + 0.5).astype(np.uint8)).save(out)
print("wrote", out, "mean", float(img.mean()), "kernel ms", t.stats().kernel_ms)

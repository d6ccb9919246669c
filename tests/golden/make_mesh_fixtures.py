#!/usr/bin/env python3
"""tests/golden/meshes.npz: the vertex / index arrays of the reference's own mesh assets.

BASELINE configs 3 and 4 name RT_Metal/coatball/coatball.obj and RT_Metal/meshes/teapot.obj (loaded by ModelIO at
AAPLRenderer.mm:474-511, placed at :513-572).  /root/reference does not exist on the GPU box, so the geometry is
committed as DATA: the 32-byte vertices {position, normal, uv} and u32 triangle indices that the host library's OBJ
reader (tracer_amd/host/mesh.cpp, standing in for ModelIO) produces from those files -- object space, untransformed;
the placement is applied by trc_host_scene_create like the reference applies it.  Run in the build container only:

    python tests/golden/make_mesh_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tracer_amd import host  # noqa: E402

REF = "/root/reference/RT_Metal"
out = {}
for name, rel in (("coatball", "coatball/coatball.obj"), ("teapot", "meshes/teapot.obj")):
    m = host.Mesh.load_obj(os.path.join(REF, rel))
    out[name + "_vertices"] = m.vertices().copy()
    out[name + "_indices"] = m.indices().copy()
    print(name, m.n_vertices, "vertices", m.n_triangles, "triangles")
np.savez_compressed(os.path.join(HERE, "meshes.npz"), **out)
print(os.path.getsize(os.path.join(HERE, "meshes.npz")), "bytes")

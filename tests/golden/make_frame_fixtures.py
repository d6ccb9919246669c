#!/usr/bin/env python3
"""Generates tests/golden/frames.npz: small frames rendered by the CPU oracle (oracle/liboracle.so).

These are REGRESSION fixtures of our own restatement (the reference cannot be executed off macOS and
holds no golden images as data -- SURVEY.md section 4); they pin the oracle against drift, and the GPU
tests compare the HIP path with the same arrays on the GPU box.
Run in the build container:  make oracle host && python tests/golden/make_frame_fixtures.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po          # noqa: E402
from tracer_amd import abi, host           # noqa: E402

CASES = {   # name: (scene kind, integrator, W, H, spp, seed, env)
    "cornell_path": (abi.SCENE_CORNELL, abi.INTEGRATOR_PATH, 48, 32, 8, 1, (0.0, 0.0, 0.0)),
    "spheres_path": (abi.SCENE_CORNELL_SPHERES, abi.INTEGRATOR_PATH, 48, 32, 8, 2, (0.0, 0.0, 0.0)),
    "spheres_mis": (abi.SCENE_CORNELL_SPHERES, abi.INTEGRATOR_MIS, 48, 32, 4, 3, (0.0, 0.0, 0.0)),
    "spheres_sky": (abi.SCENE_CORNELL_SPHERES, abi.INTEGRATOR_PATH, 40, 24, 4, 4, (0.5, 0.7, 1.0)),
}


def render_case(name):
    kind, integ, W, H, spp, seed, env = CASES[name]
    scene = host.HostScene(kind)
    cam = host.prepare_camera(W, H)
    rng = host.fill_rng(seed, W, H)
    acc, st = po.render(scene.view, cam, W, H, rng, spp=spp, integrator=integ, env=env)
    return acc, rng, st


if __name__ == "__main__":
    out = {}
    for name in CASES:
        acc, rng, st = render_case(name)
        out[name + "_accum"] = acc
        out[name + "_rng"] = rng
        out[name + "_counts"] = np.array([st.paths, st.rays, st.shaded, st.n_descend, st.n_return, st.n_leaf_sphere,
                                          st.n_leaf_square, st.n_leaf_cube], dtype=np.uint64)
        print(name, "rays", st.rays, "mean", acc[..., :3].mean())
    np.savez_compressed(os.path.join(HERE, "frames.npz"), **out)

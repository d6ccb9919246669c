#!/usr/bin/env python3
"""Generates tests/golden/sobol_frames.npz: small frames rendered by the CPU oracle with TRC_FLAG_SOBOL (regression
fixtures of our own restatement, like frames.npz; the Sobol' tables behind them are pinned to the reference's by
sobol_tables.json).  Run in the build container:  make oracle host && python tests/golden/make_sobol_frames.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import pyoracle as po          # noqa: E402
from tracer_amd import abi, host           # noqa: E402

CASES = {   # name: (scene kind, integrator, W, H, spp, seed, frame0)
    "sobol_spheres_path": (abi.SCENE_CORNELL_SPHERES, abi.INTEGRATOR_PATH, 48, 32, 8, 5, 0),
    "sobol_spheres_mis": (abi.SCENE_CORNELL_SPHERES, abi.INTEGRATOR_MIS, 44, 30, 4, 6, 9),
}


def render_case(name):
    kind, integ, W, H, spp, seed, frame0 = CASES[name]
    scene = host.HostScene(kind)
    rng = host.fill_rng(seed, W, H)
    acc, st = po.render(scene.view, host.prepare_camera(W, H), W, H, rng, spp=spp, integrator=integ, frame0=frame0, sobol=True)
    return acc, rng, st


if __name__ == "__main__":
    out = {}
    for name in CASES:
        acc, rng, st = render_case(name)
        out[name + "_accum"], out[name + "_rng"] = acc, rng
        out[name + "_counts"] = np.array([st.paths, st.rays, st.shaded], dtype=np.uint64)
        print(name, "rays", st.rays, "mean", acc[..., :3].mean())
    np.savez_compressed(os.path.join(HERE, "sobol_frames.npz"), **out)

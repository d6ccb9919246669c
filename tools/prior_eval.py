#!/usr/bin/env python3
"""How well does a STATIC prior -- what the primary rays of a block hit -- predict the block's settled cost?  """
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import workloads as wlmod
from tracer_amd import abi, host
from tracer_amd.device import Tracer
from tracer_amd.dtypes import make_rays
W, H = wlmod.W, wlmod.H


def camera_rays(cam, step):
    ys, xs = np.mgrid[step // 2:H:step, step // 2:W:step]
    u = (xs.astype(np.float32) / np.float32(W)).ravel(); v = (ys.astype(np.float32) / np.float32(H)).ravel()
    f = lambda a: np.array([a.x, a.y, a.z], dtype=np.float32)
    sample = f(cam.cornerLowLeft)[None] + f(cam.horizontal)[None] * u[:, None] + f(cam.vertical)[None] * v[:, None]
    o = np.repeat(f(cam.lookFrom)[None], len(u), 0)
    return make_rays(o, (sample - o).astype(np.float32)), xs.ravel(), ys.ravel()


def spearman(a, b):
    ra = np.argsort(np.argsort(a)).astype(np.float64); rb = np.argsort(np.argsort(b)).astype(np.float64)
    return np.corrcoef(ra, rb)[0, 1]


for config in sys.argv[1:] or ["2", "3", "4"]:
    wl = wlmod.make(config)
    with Tracer(0) as t:
        wlmod.setup(t, wl)
        spp = 64 if config == "2" else 32
        for i in range(6):
            t.seed(0x5EED0000 + i); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"])
        t.synchronize()
        tiles, costs, shift = t.block_costs()
        cam = host.prepare_camera(W, H)
        rays, xs, ys = camera_rays(cam, 2)
        hits = t.trace_rays(rays)
        mats = np.array([wl["scene"].view.materials[i].type for i in range(wl["scene"].view.n_material)])
        mtype = np.where(hits["hit"] > 0, mats[np.minimum(hits["material"], len(mats) - 1)], -1)
        tri = hits["pType"] == abi.PRIM_TRIANGLE
        w = np.full(len(mtype), 0.3)
        for mt, wt in ((abi.MAT_DIFFUSE, 0.3), (abi.MAT_LAMBERT, 1.0), (abi.MAT_PLASTIC, 1.5), (abi.MAT_METAL, 2.5), (abi.MAT_GLASS, 5.0)):
            w[mtype == mt] = wt
        w[tri] += 3.0
        bx, by = xs >> shift, ys >> shift
        nbx = (W + (1 << shift) - 1) >> shift
        prior = np.zeros(((H + (1 << shift) - 1) >> shift) * nbx)
        np.add.at(prior, by * nbx + bx, w)
        tx, ty = tiles & 0xFFFF, tiles >> 16
        c = (costs & 0xFFFFFF).astype(np.float64)
        p = prior[ty * nbx + tx]
        order_true = np.argsort(-c); order_prior = np.argsort(-p)
        top = len(c) // 10
        overlap = len(set(order_true[:top]) & set(order_prior[:top])) / top
        print(f"config {config}: {len(c)} blocks, Spearman(prior, settled cost) = {spearman(p, c):.3f}; the costliest 10 % of the blocks: {overlap:.0%} are in the prior's top 10 %; "
              f"cost max/median {c.max() / np.median(c):.1f}")

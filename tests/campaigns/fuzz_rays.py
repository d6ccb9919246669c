#!/usr/bin/env python3
"""Scene::hit campaign with rays the random batches of tests/ never contain: axis-parallel directions (one or two components exactly +-0),
origins exactly ON surfaces (the hit points of a first batch) and on box faces, directions at triangle vertices and along edges, unnormalised /
tiny / huge directions, NaN and +-inf components, tmax of 0 / tiny / inf / NaN / negative -- closest and any-hit with counters, and the
production walks, GPU against the oracle, every field bit for bit (NaN fields: both NaN).      python3 tests/campaigns/fuzz_rays.py <a> <b>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("TRC_FUZZ_SEEDS", "1:2")
import test_gpu_fuzz as tf
from oracle import pyoracle as po
from tracer_amd.device import Tracer
from tracer_amd.dtypes import make_rays
F32 = np.float32
gpu = Tracer(0)
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    rs = np.random.RandomState(70000 + seed)
    sv, keep = tf.random_scene(rs, n_spheres=int(rs.randint(3, 20)), n_cubes=int(rs.randint(1, 6)),
                               n_tris=int(rs.randint(900, 3000)) if seed % 3 == 0 else int(rs.randint(5, 60)))
    gpu.upload_scene(sv)
    n = 4000
    o = rs.uniform(-90, 90, (n, 3)).astype(F32); d = rs.normal(size=(n, 3)).astype(F32)
    first = po.trace_rays(sv, make_rays(o, d))
    onsurf = first["p"][first["hit"] != 0]
    if len(onsurf) == 0: onsurf = o
    tri = np.array([[v_ for v_ in sv.triList[i].v] for i in range(min(int(sv.n_vertex), 600))], F32)
    O, D, T = [], [], []
    def add(oo, dd, tt=None):
        O.append(np.asarray(oo, F32)); D.append(np.asarray(dd, F32)); T.append(np.full(len(oo), np.finfo(F32).max, F32) if tt is None else np.asarray(tt, F32))
    m = 1500
    # axis-parallel: zero one or two components, with both signs of zero
    dd = rs.normal(size=(m, 3)).astype(F32); z = rs.randint(0, 3, m); dd[np.arange(m), z] = np.where(rs.rand(m) < 0.5, F32(0.0), F32(-0.0))
    two = rs.rand(m) < 0.3; dd[two, (z[two] + 1) % 3] = 0.0
    add(rs.uniform(-90, 90, (m, 3)), dd)
    # ... from grid-aligned origins (box faces of the generated squares / cubes sit on round numbers rarely; integers do happen)
    add(np.round(rs.uniform(-60, 60, (m, 3))), dd)
    # origins exactly on surfaces, any direction / the same direction again (self-intersection range_t.x = FLT_MIN)
    k = rs.randint(0, len(onsurf), m)
    add(onsurf[k], rs.normal(size=(m, 3)))
    # at triangle vertices and along edges
    if len(tri) >= 3:
        v = tri[rs.randint(0, len(tri), m)]; oo = rs.uniform(-90, 90, (m, 3)).astype(F32)
        add(oo, v - oo)
        t3 = (len(tri) // 3) * 3; e0 = tri[0:t3:3][rs.randint(0, t3 // 3, m)]; e1 = tri[1:t3:3][rs.randint(0, t3 // 3, m)]
        add(e0, e1 - e0)
    # magnitudes: unnormalised tiny / huge directions
    add(rs.uniform(-90, 90, (m, 3)), rs.normal(size=(m, 3)) * np.array([1e-30, 1e-20, 1e20, 1e30])[rs.randint(0, 4, m)][:, None])
    # non-finite components in origin or direction
    oo = rs.uniform(-90, 90, (m, 3)).astype(F32); dd = rs.normal(size=(m, 3)).astype(F32)
    nasty = np.array([np.nan, -np.nan, np.inf, -np.inf, 0.0, -0.0], F32)
    which = rs.rand(m) < 0.5
    oo[which, rs.randint(0, 3, which.sum())] = nasty[rs.randint(0, 6, which.sum())]
    dd[~which, rs.randint(0, 3, (~which).sum())] = nasty[rs.randint(0, 6, (~which).sum())]
    add(oo, dd)
    # tmax values
    tm = np.array([0.0, 1e-38, 1e-3, 1.0, 50.0, np.inf, np.nan, -1.0, -0.0], F32)[rs.randint(0, 9, m)]
    add(rs.uniform(-90, 90, (m, 3)), rs.normal(size=(m, 3)), tm)
    rays = make_rays(np.concatenate(O), np.concatenate(D), np.concatenate(T))
    why = []
    with np.errstate(all="ignore"):
        for any_hit in (False, True):
            ref = po.trace_rays(sv, rays, any_hit=any_hit)
            for production in (False, True):
                got = gpu.trace_rays(rays, any_hit=any_hit, production=production)
                names = ("hit",) if (any_hit and production) else ("hit", "n_descend", "n_return", "n_leaf") if any_hit else \
                        [f for f in ref.dtype.names if not (production and f.startswith("n_"))]
                for f in names:
                    x, y = got[f], ref[f]
                    if f not in ("hit",) and not any_hit:               # fields of a miss are unspecified
                        x, y = x[ref["hit"] != 0], y[ref["hit"] != 0]
                    if not np.array_equal(x.view(np.uint32), y.view(np.uint32)):
                        nb = int((x.view(np.uint32) != y.view(np.uint32)).reshape(len(x), -1).any(axis=1).sum())
                        why.append(f"{'any' if any_hit else 'closest'}{'/production' if production else ''}.{f}: {nb}")
    if why:
        bad += 1
        print(f"MISMATCH seed {seed}: {why[:6]}", flush=True)
print(f"seeds {a}..{b - 1}: {b - a - bad} passed, {bad} FAILED")

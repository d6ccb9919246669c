#!/usr/bin/env python3
"""What a COMPACTED walk would do on config 4's scene: Scene::hit alone (k_trace, the production closest-hit walk, one ray per lane, every
lane of a wavefront busy) on 1080p primary rays and two generations of cosine-scattered secondary rays -- the rays a path tracer's bounces
are made of -- against the render kernel's rays per second on the same scene (where 18 % of the lanes are active: DESIGN section 8).
Run under the kernel trace and read the k_trace rows:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/walk -- python3 tools/walk_potential.py
    python3 tools/walk_potential.py --report gpurun_out/walk"""
import csv, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*_kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_trace" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        n = [int(x) for x in open(os.path.join(sys.argv[2], "rays.txt")).read().split()]
        per = len(rows) // len(n)
        for g, cnt in enumerate(n):
            d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[g * per:(g + 1) * per]][1:]      # first call of a generation: cold
            ms = sum(d) / len(d) / 1e6
            print(f"generation {g}: {cnt} rays, k_trace {ms:.3f} ms -> {cnt / ms / 1e3:.0f} Mrays/s (walk + record only, every lane busy)")
    sys.exit(0)
from tracer_amd import abi, host
from tracer_amd.device import Tracer
from tracer_amd.dtypes import make_rays
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
W, H = 1920, 1080
wl = wlmod.make("4")
t = Tracer(0); wlmod.setup(t, wl)
cam = host.prepare_camera(W, H)
ys, xs = np.mgrid[0:H, 0:W]
ty, tx = ys // 8, xs // 8                          # 8x8 pixel blocks per wavefront, like the render kernels
order = np.lexsort(((xs % 8).ravel(), (ys % 8).ravel(), tx.ravel(), ty.ravel()))
u = xs.ravel()[order].astype(np.float32) / np.float32(W); v = ys.ravel()[order].astype(np.float32) / np.float32(H)
f = lambda a: np.array([a.x, a.y, a.z], dtype=np.float32)
sample = f(cam.cornerLowLeft)[None] + f(cam.horizontal)[None] * u[:, None] + f(cam.vertical)[None] * v[:, None]
o = np.repeat(f(cam.lookFrom)[None], len(u), 0)
rays = make_rays(o, (sample - o).astype(np.float32))
rng = np.random.default_rng(1)
counts = []
for gen in range(3):
    for _ in range(4):
        hits = t.trace_rays(rays, production=True)
    counts.append(len(rays))
    ok = hits["hit"] != 0
    print(f"gen {gen}: {len(rays)} rays, {int(ok.sum())} hit", flush=True)
    hits = hits[ok]
    n = hits["sn"].astype(np.float32); n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-20)
    r1, r2 = rng.random(len(n), dtype=np.float32), rng.random(len(n), dtype=np.float32)
    phi = 2 * np.pi * r1; rr = np.sqrt(r2)
    loc = np.stack([rr * np.cos(phi), rr * np.sin(phi), np.sqrt(1 - r2)], 1).astype(np.float32)
    a = np.where(np.abs(n[:, :1]) > 0.9, np.array([[0, 1, 0]], np.float32), np.array([[1, 0, 0]], np.float32))
    tt = np.cross(n, a); tt /= np.linalg.norm(tt, axis=1, keepdims=True); bb = np.cross(n, tt)
    d = (loc[:, :1] * tt + loc[:, 1:2] * bb + loc[:, 2:3] * n).astype(np.float32)
    rays = make_rays((hits["p"] + 1e-2 * n).astype(np.float32), d)
out = os.environ.get("WALK_OUT", "gpurun_out/walk")
os.makedirs(out, exist_ok=True)
open(os.path.join(out, "rays.txt"), "w").write(" ".join(str(c) for c in counts))

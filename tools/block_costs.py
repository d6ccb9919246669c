#!/usr/bin/env python3
"""Where a launch's tail comes from: per-block durations of one launch (trc_debug_block_costs) and the kernel time of the
same pixels cut into 8x8 / 4x4 / 2x2 / 1x1 pixel blocks per wavefront (knob force_blk_shift), for rank 0 of N.

    python3 tools/block_costs.py --config 2|3|4 [--spp S] [--ranks 1,8]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--ranks", default="1,8")
ap.add_argument("--shifts", default="3,2,1,0")
a = ap.parse_args()
wl = wlmod.make(a.config)
spp = a.spp or wl["spp"]
t = Tracer(0)
wlmod.setup(t, wl)
print(f"{wl['what']}, {wlmod.W}x{wlmod.H}x{spp}spp")
for N in [int(x) for x in a.ranks.split(",")]:
    for shift in [int(x) for x in a.shifts.split(",")]:
        t.debug_set("force_blk_shift", shift + 1)
        for rep in range(2):          # second launch: adaptive order
            t.seed(0x5EED0000); t.reset_stats(); t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N); t.synchronize()
        ms = t.stats().kernel_ms
        tiles, costs, bs = t.block_costs()
        c = costs.astype(np.float64)
        scale = ms / c.max() if c.max() > 0 else 0.0          # the longest block cannot last longer than the launch
        q = np.percentile(c, [50, 90, 99, 99.9, 100]) * scale
        slots = 256 * 16
        print(f"N={N} rank 0, {1 << shift}x{1 << shift} blocks: {len(c)} blocks, kernel {ms:.2f} ms; block duration (ms, longest := kernel) "
              f"p50 {q[0]:.3f} p90 {q[1]:.3f} p99 {q[2]:.3f} p99.9 {q[3]:.3f} max {q[4]:.3f}; sum/4096 slots {c.sum() * scale / slots:.2f} ms")
t.debug_set("force_blk_shift", 0)

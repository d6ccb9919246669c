#!/usr/bin/env python3
"""Launch by launch: kernel ms and how many 8x8 blocks the cost-adaptive plan ran as quarters (rank 0 of N).
    python3 tools/split_trace.py --config 2 --ranks 1,4,8 [--launches 8]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2"); ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--ranks", default="1,4,8"); ap.add_argument("--launches", type=int, default=8)
ap.add_argument("--vary-seed", action="store_true")
a = ap.parse_args()
wl = wlmod.make(a.config); spp = a.spp or wl["spp"]
t = Tracer(0); wlmod.setup(t, wl)
print(f"{wl['what']}, {spp} spp")
for N in [int(x) for x in a.ranks.split(",")]:
    for mode in ("adaptive", "no_split"):
        t.debug_set("no_split", 1 if mode == "no_split" else 0)
        out = []
        for i in range(a.launches):
            t.seed(0x5EED0000 + (i if a.vary_seed else 0)); t.reset_stats()
            t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N); t.synchronize()
            _, costs, _ = t.block_costs()
            sh = t.launch_shape()
            out.append(f"{t.stats().kernel_ms:.2f}/{int((costs >> 31).sum())}/{int(((costs >> 30) & 1).sum())}/{int(((costs >> 29) & 1).sum())}"
                       f"[{sh['entries']}:{sh['longest_entry_ms']:.1f}]")
        print(f"N={N} {mode}: {len(costs)} blocks; kernel ms / blocks run in parts / of those with 2x2 sixteenths / with single pixels [items : longest item ms], per launch: " + " ".join(out))

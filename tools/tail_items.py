#!/usr/bin/env python3
"""What a settled launch ends on: the blocks whose slowest item (whole block, 4x4 quarter or 2x2 sixteenth) lasts nearly as long
as the launch (trc_debug_block_costs + trc_debug_launch_shape).  python3 tools/tail_items.py --config 3|4 [--spp S]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="4")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--launches", type=int, default=7)
a = ap.parse_args()
wl = wlmod.make(a.config)
spp = a.spp or wl["spp"]
t = Tracer(0)
wlmod.setup(t, wl)
for i in range(a.launches):
    t.seed(0x5EED0000); t.reset_stats(); t.render(spp=spp, integrator=wl["integrator"]); t.synchronize()
ms = t.stats().kernel_ms
shape = t.launch_shape()
tiles, costs, bs = t.block_costs()
c = (costs & 0xFFFFFF).astype(np.float64)          # bits 31 / 30 / 29: ran in parts / some as 2x2 sixteenths / some as single pixels
parts, deep = (costs >> 31) & 1, (costs >> 30) & 1
scale = shape["longest_entry_ms"] / c.max()
d = c * scale
print(f"{wl['what']}, {spp} spp: kernel {ms:.1f} ms, chain bound {shape['longest_entry_ms']:.1f}, work bound {shape['work_over_slots_ms']:.1f}, "
      f"{len(c)} blocks, {int(parts.sum())} in parts, {int(deep.sum())} with sixteenths")
for frac in (0.98, 0.95, 0.9, 0.85, 0.8, 0.7):
    sel = d >= frac * d.max()
    print(f"  slowest item >= {frac:.2f} x the longest ({frac * d.max():7.1f} ms): {int(sel.sum()):5d} blocks ({int((sel & (deep == 1)).sum())} of them already down to 2x2, "
          f"{int((sel & (parts == 1) & (deep == 0)).sum())} at 4x4, {int((sel & (parts == 0)).sum())} whole)")

#!/usr/bin/env python3
"""A rank's share of a frame against the two bounds no schedule can beat (trc_debug_launch_shape): its slowest wavefront-sized
item (a pixel's samples are one chain) and its summed durations over the GPU's wavefront slots.
    python3 tools/share_bounds.py --config 2|3|4 [--spp N] [--ranks 1,2,4,8] [--settle 10]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2"); ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--ranks", default="1,2,4,8"); ap.add_argument("--settle", type=int, default=10)
a = ap.parse_args()
wl = wlmod.make(a.config); spp = a.spp or wl["spp"]
print(f"{wl['what']}, {spp} spp; rank 0's share, settled over {a.settle} launches")
print("# N  kernel_ms  longest_item_ms  work/slots_ms  items  slots  ideal(N=1 kernel / N)")
base = None
with Tracer(0) as t:
    wlmod.setup(t, wl)
    for N in [int(x) for x in a.ranks.split(",")]:
        for i in range(a.settle):
            t.seed(0x5EED0000 + i); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N)
        t.synchronize(); t.reset_stats()
        K = 4
        for i in range(K):
            t.seed(0x5EED0100 + i); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N)
        t.synchronize()
        ms = t.stats().kernel_ms / K
        sh = t.launch_shape()
        if base is None: base = ms * N
        print(f"{N:3d}  {ms:8.3f}  {sh['longest_entry_ms']:8.3f}  {sh['work_over_slots_ms']:8.3f}  {sh['entries']:6d}  {sh['wave_slots']:5d}  {base / N:8.3f}")

#!/usr/bin/env python3
"""A rank's share of a frame against the two bounds no schedule can beat (trc_debug_launch_shape): its slowest wavefront-sized
item (a pixel's samples are one chain) and its summed durations over the GPU's wavefront slots.
Both splits of an N-GPU frame, rank 0's share of each (one GPU emulates one rank: the other ranks do the same amount of work):
  tiles    the rank's 16x16 tiles, ALL samples of their pixels (the 1-GPU-identical frame): chains stay whole
  samples  the WHOLE frame, spp / N samples from seed trc_shard_seed(seed, 0) (sample sharding, tracer_abi.h): chains shrink
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
from tracer_amd import abi
print("# split    N  kernel_ms  longest_item_ms  work/slots_ms  items  slots  ideal(N=1 kernel / N)  efficiency")
base = None
with Tracer(0) as t:
    wlmod.setup(t, wl)
    for split in ("tiles", "samples"):
        for N in [int(x) for x in a.ranks.split(",")]:
            if split == "samples" and (N == 1 or spp % N):
                continue
            kw = dict(spp=spp, tile_rank=0, tile_nranks=N) if split == "tiles" else dict(spp=spp // N)

            def launch(seed):
                t.seed(abi.shard_seed(seed, 0)); t.clear_accum(); t.render(integrator=wl["integrator"], **kw)
            for i in range(a.settle):
                launch(0x5EED0000 + i)
            t.synchronize(); t.reset_stats()
            K = 4
            for i in range(K):
                launch(0x5EED0100 + i)
            t.synchronize()
            ms = t.stats().kernel_ms / K
            sh = t.launch_shape()
            if base is None: base = ms * N
            print(f"{split:8s} {N:3d}  {ms:8.3f}  {sh['longest_entry_ms']:8.3f}  {sh['work_over_slots_ms']:8.3f}  {sh['entries']:6d}  {sh['wave_slots']:5d}  {base / N:8.3f}  {base / N / ms:6.3f}")

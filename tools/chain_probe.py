#!/usr/bin/env python3
"""Would a third split level (single pixels) shorten the chains a strong-scaled share ends on?  Rank 0 of 8 of configs 2 / 4 / 3
with the WHOLE list cut into 8x8 (the adaptive plan), 4x4, 2x2 and 1x1 pixel blocks (knob force_blk_shift), longest-first:
the longest item of each launch (trc_debug_launch_shape) is the chain no schedule of that block size can beat."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer

for cfg, N in (("2", 8), ("4", 8), ("3", 8)):
    wl = wlmod.make(cfg); t = Tracer(0); wlmod.setup(t, wl)
    spp = wl["spp"]
    for shift in (3, 2, 1, 0):
        t.debug_set("force_blk_shift", shift + 1)
        for i in range(5):
            t.seed(0x5EED0000); t.reset_stats(); t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N); t.synchronize()
        s = t.launch_shape(); ms = t.stats().kernel_ms
        print(f"config {cfg} rank 0 of {N}, {1 << shift}x{1 << shift} blocks: kernel {ms:.2f} ms, longest item {s['longest_entry_ms']:.2f}, "
              f"work/slots {s['work_over_slots_ms']:.2f}, entries {s['entries']}")
    t.debug_set("force_blk_shift", 0)
    del t

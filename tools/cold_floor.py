#!/usr/bin/env python3
"""The floor of a head + rest first launch: on SETTLED block costs, the named frame as 8 samples + the rest by hand (two ordered launches), beside the
first launch of a fresh context under the head knobs (probe_spp, head_stages).  What separates the two is the head's own disorder and the rest being
planned on 8 cold samples.       python3 tools/cold_floor.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer
for cfg in ("2", "3", "4"):
    wl = wlmod.make(cfg); spp = wl["spp"]
    def run(t, k, f0=0, seed=0x5EED0000, fresh=True):
        if fresh: t.seed(seed); t.clear_accum()
        t.reset_stats(); t.render(spp=k, integrator=wl["integrator"], frame0=f0); t.synchronize(); return t.stats().kernel_ms
    with Tracer(0) as t:
        wlmod.setup(t, wl)
        ms = [run(t, spp) for _ in range(10)]
        settled = sum(ms[-3:]) / 3
        # settled costs: the same frame as head + rest by hand (no_coalesce irrelevant: 8 >= 8)
        h = run(t, 8); r = run(t, spp - 8, f0=8, fresh=False)
        h2 = run(t, 8); r2 = run(t, spp - 8, f0=8, fresh=False)
    print(f"config {cfg}: first {ms[0]:.2f}  settled {settled:.2f}  | on settled costs: head(8) {h:.2f} + rest({spp-8}) {r:.2f} = {h+r:.2f}; again {h2:.2f} + {r2:.2f} = {h2+r2:.2f}", flush=True)
    for knob, val in (("probe_spp", 16), ("probe_spp", 12), ("head_stages", 1), ("head_stages", 2)):
        with Tracer(0) as t:
            wlmod.setup(t, wl); t.debug_set(knob, val)
            ms = [run(t, spp) for _ in range(3)]
        print(f"   fresh context, {knob}={val}: " + " ".join(f"{m:.2f}" for m in ms), flush=True)

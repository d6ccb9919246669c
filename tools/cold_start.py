#!/usr/bin/env python3
"""What the first launch of a block list costs, and what a cheap probe launch ahead of it would buy.
    python3 tools/cold_start.py --config 2|3|4 [--spp N] [--ranks 1,8]
Per configuration and rank count (fresh context each): kernel ms of launch 1 ... 10 ("cold" first, settled last), and the
same first launch after a probe of 1 / 2 / 4 samples per pixel on the same block list (emulated through the public calls:
render(spp = k) with knob strip_len = 1 leaves per-block durations, the next launch is ordered and planned by them)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer
ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2"); ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--ranks", default="1"); ap.add_argument("--launches", type=int, default=10)
ap.add_argument("--native", action="store_true", help="the library's own probe (on unless knob no_cold_probe) instead of the emulation")
a = ap.parse_args()
wl = wlmod.make(a.config); spp = a.spp or wl["spp"]
print(f"{wl['what']}, {spp} spp")


def one(t, N, seed=0x5EED0000):
    t.seed(seed); t.clear_accum(); t.reset_stats()
    t.render(spp=spp, integrator=wl["integrator"], tile_rank=0, tile_nranks=N); t.synchronize()
    return t.stats().kernel_ms


for N in [int(x) for x in a.ranks.split(",")]:
    with Tracer(0) as t:
        wlmod.setup(t, wl)
        t.debug_set("no_cold_probe", 1)
        ms = [one(t, N) for _ in range(a.launches)]
    print(f"N={N} no probe : " + " ".join(f"{m:.2f}" for m in ms))
    if a.native:
        with Tracer(0) as t:
            wlmod.setup(t, wl)
            ms = [one(t, N) for _ in range(a.launches)]
        print(f"N={N} library's probe (first launch includes it): " + " ".join(f"{m:.2f}" for m in ms))
        continue
    for k in (1, 2, 4):
        with Tracer(0) as t:
            wlmod.setup(t, wl)
            t.debug_set("no_cold_probe", 1)
            t.debug_set("strip_len", 1)
            t.seed(0x5EED0000); t.reset_stats()
            t.render(spp=k, integrator=wl["integrator"], tile_rank=0, tile_nranks=N); t.synchronize()
            probe_ms = t.stats().kernel_ms
            ms = [one(t, N) for _ in range(4)]
        print(f"N={N} probe {k} spp ({probe_ms:.2f} ms): " + " ".join(f"{m:.2f}" for m in ms))

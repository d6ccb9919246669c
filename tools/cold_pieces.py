#!/usr/bin/env python3
"""Where the first launch of config 2 loses its time: the 8-sample head and the 56-sample rest of a fresh context, each against
the same launch once the order has settled (kernel ms from HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer
wl = wlmod.make(sys.argv[1] if len(sys.argv) > 1 else "2")
H8, REST = 8, wl["spp"] - 8


def launch(t, spp, frame0=0, seed=0x5EED0000, clear=True):
    if clear:
        t.seed(seed); t.clear_accum()
    t.reset_stats()
    t.render(spp=spp, integrator=wl["integrator"], frame0=frame0); t.synchronize()
    return t.stats().kernel_ms


with Tracer(0) as t:
    wlmod.setup(t, wl)
    t.debug_set("no_cold_probe", 1)
    a = launch(t, H8); b = launch(t, REST, frame0=H8, clear=False)
    print(f"fresh context, by hand: head {H8} spp {a:.2f} ms (row-major), rest {REST} spp {b:.2f} ms (ordered by the head) = {a + b:.2f}")
    for i in range(10):
        launch(t, wl["spp"], seed=100 + i)
    whole = launch(t, wl["spp"])
    for i in range(6):
        launch(t, H8, seed=200 + i)
    a2 = launch(t, H8)
    for i in range(6):
        launch(t, REST, seed=300 + i)
    b2 = launch(t, REST)
    print(f"settled: whole {wl['spp']} spp {whole:.2f} ms; a settled {H8}-spp launch {a2:.2f} ms, a settled {REST}-spp launch {b2:.2f} ms (sum {a2 + b2:.2f})")
with Tracer(0) as t:
    wlmod.setup(t, wl)
    first = launch(t, wl["spp"])
    print(f"fresh context, library's head + rest: {first:.2f} ms")

# the same with the kernels' code already loaded and run once (a 64 x 64 frame first): what of the first launch is one-time set-up
# of the process (module load, first use of the buffers) and what is the missing order
from tracer_amd import host
with Tracer(0) as t:
    wlmod.setup(t, wl)
    t.resize(64, 64); t.set_camera(host.prepare_camera(64, 64))
    launch(t, wl["spp"])
    t.resize(wlmod.W, wlmod.H); t.set_camera(host.prepare_camera(wlmod.W, wlmod.H))
    first = launch(t, wl["spp"])
    print(f"fresh block list, code warm: library's head + rest: {first:.2f} ms")
with Tracer(0) as t:
    wlmod.setup(t, wl)
    t.debug_set("no_cold_probe", 1)
    t.resize(64, 64); t.set_camera(host.prepare_camera(64, 64))
    launch(t, wl["spp"])
    t.resize(wlmod.W, wlmod.H); t.set_camera(host.prepare_camera(wlmod.W, wlmod.H))
    a = launch(t, H8); b = launch(t, REST, frame0=H8, clear=False)
    print(f"fresh block list, code warm, by hand: head {a:.2f} ms, rest {b:.2f} ms = {a + b:.2f}")

# is it the GPU that is cold, not the block list?  another context keeps the GPU busy right up to the fresh context's first launch
with Tracer(0) as busy:
    wlmod.setup(busy, wl)
    for i in range(12):
        launch(busy, wl["spp"], seed=500 + i)
    with Tracer(0) as t:
        wlmod.setup(t, wl)
        for i in range(3):
            launch(busy, wl["spp"], seed=600 + i)
        first = launch(t, wl["spp"])
        second = launch(t, wl["spp"], seed=7)
        third = launch(t, wl["spp"], seed=8)
        print(f"fresh context on a busy GPU: first launch {first:.2f} ms, then {second:.2f}, {third:.2f}")

#!/usr/bin/env python3
"""Strong-scaling load balance of the tile ownership rule, measured on ONE GPU: render the tiles of every rank
r of N separately and compare kernel times (the N-GPU step takes max over ranks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracer_amd import abi, host
from tracer_amd.device import Tracer
W, H = 1920, 1080
sc = host.HostScene(abi.SCENE_CORNELL_SPHERES)
t = Tracer(0); t.upload_scene(sc.view); t.set_camera(host.prepare_camera(W, H)); t.resize(W, H)
t.seed(0x5EED0000); t.render(spp=64); t.synchronize()
t.seed(0x5EED0000); t.reset_stats(); t.render(spp=64); t.synchronize(); full = t.stats().kernel_ms
print(f"N=1: {full:.2f} ms")
for small in (False, True):
    print("strong scaling of the named frame, " + ("4x4 pixel blocks on 16 lanes (TRC_FLAG_SMALL_BLOCKS)" if small else "8x8 pixel blocks"))
    for N in (1, 2, 4, 8):
        ms = []
        for r in range(N):
            for rep in range(2):            # the second launch of a block list runs in adaptive (expensive-first) order
                t.seed(0x5EED0000); t.reset_stats(); t.render(spp=64, tile_rank=r, tile_nranks=N, small_blocks=small); t.synchronize()
            ms.append(t.stats().kernel_ms)
        print(f"N={N}: per-rank kernel ms {' '.join(f'{m:.2f}' for m in ms)}; max {max(ms):.2f}, ideal {full / N:.2f}, "
              f"render-only efficiency {full / N / max(ms):.3f}")

print("weak-scaling workload of bench.py: N views stacked, one view's worth of tiles per rank")
for N in (2, 8):
    t.resize(W, H * N)
    ms = []
    for r in range(N):
        for rep in range(2):
            t.seed(0x5EED0000); t.reset_stats(); t.render(spp=64, tile_rank=r, tile_nranks=N, view_height=H); t.synchronize()
        ms.append(t.stats().kernel_ms)
    print(f"N={N}: per-rank kernel ms {' '.join(f'{m:.2f}' for m in ms)}; max {max(ms):.2f} vs N=1 {full:.2f}: efficiency {full / max(ms):.3f}")

#!/usr/bin/env python3
"""Per-rank cost of an N-rank split of a BASELINE configuration, measured on ONE GPU: the share of every rank r of N is
run alone and timed with the library's own HIP events (trc_stats.kernel_ms); the N-GPU step lasts as long as its slowest
rank.  No hardware scaling curve exists for this repository (one GPU per box): this is the emulation.

    python3 tools/tile_balance.py --config 2|4|5 [--spp S] [--ranks 2,4,8]

config 2 / 4 (path tracing): rank r renders the tiles (tx + ty) % N == r of the one named frame, every rank --launches times
(from the second launch of a block list on, the adaptive expensive-first order and the cost-adaptive block size apply and
settle within ~8 launches; the mean of the last three is reported).
config 5 (SPPM): rank r runs its share of every pass (camera + refine on its tiles, its photon index range, hash / table
over ALL photons) with the collectives of trc_group_set_collectives served from a 1-rank pass running in lock step in a
second context (bit-identical photons: tests/test_gpu_shared_gpu_ranks.py), timed per frame with the "sppm_timing" knob:
[photon pass] + [hash, table, refine], collectives excluded.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod  # noqa: E402
from tracer_amd import abi  # noqa: E402
from tracer_amd.device import Tracer  # noqa: E402
from tracer_amd.gloo_collectives import ALLGATHER_FN, ALLREDUCE_FN, REDUCE_FN, Collectives  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--ranks", default="2,4,8")
ap.add_argument("--frames", type=int, default=16, help="config 5: SPPM frames")
ap.add_argument("--launches", type=int, default=10, help="configs 2 / 4: launches per rank (the plan settles within ~8); the mean of the last 3 is reported")
ap.add_argument("--small-blocks", action="store_true", help="configs 2 / 4: also the forced 4x4-block variant")
a = ap.parse_args()
W, H = wlmod.W, wlmod.H
ranks = [int(x) for x in a.ranks.split(",")]


def path_traced(cfg):
    wl = wlmod.make(cfg)
    spp = a.spp or wl["spp"]
    t = Tracer(0)
    wlmod.setup(t, wl)
    print(f"{wl['what']}, {W}x{H}x{spp}spp; per-rank kernel ms of an N-rank tile split, emulated on one GPU")
    last = []
    for rep in range(a.launches):                   # the same treatment as every rank below: settled order, mean of the last 3
        t.seed(0x5EED0000); t.reset_stats(); t.render(spp=spp, integrator=wl["integrator"]); t.synchronize()
        last.append(t.stats().kernel_ms)
    full = sum(last[-3:]) / len(last[-3:])
    print(f"N=1: {full:.2f} ms")
    variants = [(None, "default block size (trc_render decides)")] + ([(True, "4x4 pixel blocks forced (TRC_FLAG_SMALL_BLOCKS)")] if a.small_blocks else [])
    for small, label in variants:
        print(label)
        for N in ranks:
            ms, rays = [], []
            for r in range(N):
                last = []
                for rep in range(a.launches):   # launch 1 measures the blocks, later ones run in adaptive order with the plan's parts
                    t.seed(0x5EED0000); t.reset_stats()
                    t.render(spp=spp, integrator=wl["integrator"], tile_rank=r, tile_nranks=N, small_blocks=small); t.synchronize()
                    st = t.stats()
                    last.append(st.kernel_ms)
                ms.append(sum(last[-3:]) / len(last[-3:])); rays.append(st.rays)
            print(f"N={N}: per-rank kernel ms {' '.join(f'{m:.2f}' for m in ms)}; max {max(ms):.2f}, ideal {full / N:.2f}, "
                  f"render-only efficiency {full / N / max(ms):.3f}; rays per rank min/max {min(rays)}/{max(rays)}")
    if cfg == "2":
        print("weak-scaling workload of bench.py: N views stacked into one 1920 x 1080 N frame, one view's worth of tiles per rank")
        for N in (2, 8):
            t.resize(W, H * N)
            ms = []
            for r in range(N):
                last = []
                for rep in range(a.launches):
                    t.seed(0x5EED0000); t.reset_stats(); t.render(spp=spp, integrator=wl["integrator"], tile_rank=r, tile_nranks=N, view_height=wlmod.H); t.synchronize()
                    last.append(t.stats().kernel_ms)
                ms.append(sum(last[-3:]) / len(last[-3:]))
            print(f"N={N}: per-rank kernel ms {' '.join(f'{m:.2f}' for m in ms)}; max {max(ms):.2f} vs N=1 {full:.2f}: efficiency {full / max(ms):.3f}")
    t.close()


class ReplayCollectives:
    """Collectives of ONE emulated rank, served from the state of a 1-rank pass (`full`) that runs in lock step."""

    def __init__(self, full, world, rank):
        self.full, self.world, self.rank = full, world, rank
        self.pho = None
        self.keys = None
        self._cb = (REDUCE_FN(self._reduce), ALLREDUCE_FN(self._allreduce), ALLGATHER_FN(self._allgather))
        self.table = Collectives(None, 1, 0, *self._cb)

    def _reduce(self, user, buf, count, dtype, op, root, stream):
        return 0

    def _allreduce(self, user, buf, count, dtype, op, stream):
        # the bound keys of frame 0: min / max over all ranks == what the 1-rank pass computed; the emulated rank only
        # sees its own tiles, so hand it the full result (self.keys = [min3, max3], order-preserving u32 keys)
        k = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint32)), shape=(count,))
        k[:] = self.keys[0] if op == abi.OP_MIN else self.keys[1]
        return 0

    def _allgather(self, user, buf, bytes_per_rank, stream):
        a8 = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(bytes_per_rank * self.world,))
        mine = slice(self.rank * bytes_per_rank, (self.rank + 1) * bytes_per_rank)
        assert np.array_equal(a8[mine], self.pho[mine]), "the emulated rank's photons differ from the 1-rank pass"
        a8[:] = self.pho
        return 0


def float_key(v):
    """order-preserving u32 key of a float (trc_sppm.hip)"""
    b = np.float32(v).view(np.uint32)
    return np.uint32(~b) if b & 0x80000000 else np.uint32(b | 0x80000000)


def sppm():
    wl = wlmod.make("2")
    print(f"config 5: SPPM on {wl['what']}, {W}x{H}, 512^2 photons per frame, {a.frames} frames; per-rank GPU ms per frame of "
          f"[photon pass] + [hash, table, refine] (collectives excluded), emulated on one GPU")

    def one(world, rank):
        full, t = Tracer(0), Tracer(0)
        for x in (full, t):
            wlmod.setup(x, wl)
            x.clear_accum(); x.seed(0x5EED0050); x.sppm_init(0x5EED0051)
        rc = ReplayCollectives(full, world, rank)
        if world > 1:
            t.set_collectives(rc, world, rank)
        t.debug_set("sppm_timing", 1)
        t.debug_set("sppm_serial_camera", 1)          # one frame per call below: no look-ahead to overlap anyway
        t.reset_stats()
        for f in range(a.frames):
            full.sppm_frames(1)
            cam, pho, _, _, cx = full.sppm_download()
            rc.pho = pho.view(np.uint8).ravel()
            if f == 0:          # the bound of the valid visible points, as keys (what all ranks' min / max all-reduce to)
                pos = cam["position"][cam["valid"] != 0][:, :3]
                rc.keys = (np.array([float_key(v) for v in pos.min(axis=0)], np.uint32),
                           np.array([float_key(v) for v in pos.max(axis=0)], np.uint32))
            t.sppm_frames(1)
            t.synchronize()
        st = t.stats()
        full.close(); t.close()
        return st.kernel_ms / a.frames

    base = one(1, 0)
    print(f"N=1: {base:.3f} ms per frame")
    for N in ranks:
        ms = [one(N, r) for r in range(N)]
        print(f"N={N}: per-rank ms per frame {' '.join(f'{m:.3f}' for m in ms)}; max {max(ms):.3f}, ideal {base / N:.3f}, "
              f"efficiency before the all-gather {base / N / max(ms):.3f}; all-gather per frame: {512 * 512 * 80 / 1e6:.1f} MB in total")


if a.config in ("2", "4", "3"):
    path_traced(a.config)
elif a.config == "5":
    sppm()
else:
    raise SystemExit("config 2, 3, 4 or 5")

#!/usr/bin/env python3
"""What bench.py holds its own frames to: the CPU oracle's ray counts and CRC32s of the frames at the sizes BASELINE.json names.

    python tests/golden/make_bench_goldens.py [--only config2,samples,weak,config3,config4,volume,sppm] [--out FILE]

Every entry is the ORACLE's (oracle/liboracle.so: test infrastructure, the checker) on exactly the calls bench.py makes on the GPU:
same scene, camera, seed, samples.  Needs no reference and no GPU; the mesh configurations are a few 10^9 rays, so it is meant for
a machine with many cores (the GPU box's host: `gpurun -- python tests/golden/make_bench_goldens.py --out gpurun_out/...`).
Merges into tests/golden/bench_goldens.json (entries it does not regenerate stay).

  config2          1920x1080x64 spp tracePath, seed 0x5EED0000: rays, crc of the accumulator -- also what the N-rank TILE split
                   must compose to (bit-identical to one GPU by construction)
  samples S=2,4,8  the sample-sharded frame (pyoracle.render_sample_sharded, include/tracer_abi.h): crc per S
  weak N=2,4,8     N stacked views (trc_params.view_height): crc per N
  config3 / 4 / volume   as named (traceMIS 256 spp / tracePath 256 spp / traceVolume 64 spp): rays, crc
  sppm             config 5: 64 SPPM frames, seeds 1 / 2: totalPhotonSum, crc of the refined frame
"""
import argparse
import json
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle as po          # noqa: E402
from tracer_amd import abi, host           # noqa: E402
import workloads as wlmod                  # noqa: E402

W, H, SEED = 1920, 1080, 0x5EED0000
GOLD = os.path.join(ROOT, "tests", "golden", "bench_goldens.json")


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes()) & 0xFFFFFFFF


def whole(config, out):
    wl = wlmod.make(config)
    cam = host.prepare_camera(W, H)
    if wl["density"] is not None:
        po.set_density(host.density_info(wl["density"]), wl["density"])
    try:
        rng = host.fill_rng(SEED, W, H)
        t0 = time.time()
        acc, st = po.render(wl["scene"].view, cam, W, H, rng, spp=wl["spp"], max_depth=8, integrator=wl["integrator"])
        out["volume" if config == "volume" else "config" + config] = {
            "what": wl["what"], "spp": wl["spp"], "seed": SEED, "rays": int(st.rays), "paths": int(st.paths),
            "crc_accum": crc(acc), "crc_rng": crc(rng), "oracle_s": round(time.time() - t0, 1)}
    finally:
        po.set_density(None, None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="config2,samples,weak,config3,config4,volume,sppm")
    ap.add_argument("--out", default=GOLD)
    a = ap.parse_args()
    todo = a.only.split(",")
    out = json.load(open(GOLD)) if os.path.exists(GOLD) else {}
    scene2 = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    for what in todo:
        t0 = time.time()
        if what in ("config2", "config3", "config4", "volume"):
            whole(what.replace("config", "") if what != "volume" else "volume", out)
        elif what == "samples":
            for S in (2, 4, 8):
                rngs = [host.fill_rng(abi.shard_seed(SEED, g), W, H) for g in range(S)]
                acc, sts = po.render_sample_sharded(scene2.view, cam, W, H, rngs, 64, max_depth=8, integrator=abi.INTEGRATOR_PATH)
                out[f"config2_samples_S{S}"] = {"spp": 64, "seed": SEED, "sample_groups": S, "rays": int(sum(s.rays for s in sts)), "crc_accum": crc(acc)}
        elif what == "weak":
            for N in (2, 4, 8):
                rng = host.fill_rng(SEED, W, H * N)
                acc, st = po.render(scene2.view, cam, W, H * N, rng, spp=64, max_depth=8, integrator=abi.INTEGRATOR_PATH, view_height=H)
                out[f"config2_weak_N{N}"] = {"spp": 64, "seed": SEED, "views": N, "rays": int(st.rays), "crc_accum": crc(acc)}
        elif what == "sppm":
            rng = host.fill_rng(1, W, H)
            acc = np.zeros((H, W, 4), np.float32)
            s = po.Sppm(W, H, 2)
            s.frames(scene2.view, cam, rng, acc, 64)
            cx = s.download()[4]
            out["config5_sppm"] = {"frames": 64, "canvas_seed": 1, "photon_seed": 2, "totalPhotonSum": int(cx.totalPhotonSum),
                                   "frame_count": int(cx.frame_count), "crc_accum": crc(acc), "crc_rng": crc(rng)}
        print(what, f"{time.time() - t0:.1f} s", flush=True)
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()

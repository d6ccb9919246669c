#!/usr/bin/env python3
"""Stress harness for the one unexplained SPPM mismatch on record (profiles/r05/fuzz_campaign.txt: one of five
identical 3 000-scene campaigns under 16 processes on ONE GPU had nine scenes with photon records whose `flux` was 0 where the oracle
has the reset value 1 -- records that look unwritten).

The scenes of tests/test_gpu_fuzz.py::test_generated_scene_sppm (64x48, 3 frames), but the photon records are downloaded after EVERY
frame and held to the oracle's, so a mismatch is caught in the frame it appears in; at the first one the harness dumps what is needed to
place it: frame, photon indices (their wavefront = index // 64, the workgroup of k_sppm_photon that owns them), whether the records are
all-zero ("never written / cleared") or partly written, the neighbours' state, a SECOND download of the same device buffer (a transfer
that lost bytes differs from a kernel that did not store them), and the bytes.

    python3 tests/campaigns/sppm_stress.py <first seed> <last seed> [--procs 16] [--serialize] [--serial-camera]

--procs N        N worker processes on the one GPU, each with the seeds s with s % N == its index (the condition the mismatch was seen under)
--serialize      AMD_SERIALIZE_KERNEL=3 / AMD_SERIALIZE_COPY=3 in the workers: every launch and copy waits for the one before
--serial-camera  knob sppm_serial_camera: the camera pass of an odd frame without its frame of lead on the second stream
"""
import argparse
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(first, last, index, procs, serial_camera):
    import numpy as np
    os.environ.setdefault("TRC_FUZZ_SEEDS", "1:2")
    import test_gpu_fuzz as tf
    from oracle import pyoracle as po
    from tracer_amd import host
    from tracer_amd.device import Tracer
    gpu = Tracer(0)
    gpu.debug_set("sppm_serial_camera", 1 if serial_camera else 0)
    W, H, frames = 64, 48, 3
    bad = done = 0
    for seed in range(first, last):
        if seed % procs != index:
            continue
        rs = np.random.RandomState(3000 + seed)
        sv, keep = tf.random_scene(rs, n_spheres=int(rs.randint(3, 12)), n_cubes=int(rs.randint(1, 4)), n_tris=int(rs.randint(5, 80)))
        cam = host.make_camera((rs.uniform(-100, 100), rs.uniform(-60, 60), -160.0), (0, 0, 0), (0, 1, 0), 0.0, W / H, math.radians(50), 160.0)
        gpu.upload_scene(sv); gpu.set_camera(cam); gpu.set_environment((0.0, 0.0, 0.0)); gpu.resize(W, H)
        gpu.seed(8); gpu.clear_accum(); gpu.sppm_init(40 + seed)
        rng = host.fill_rng(8, W, H); acc = np.zeros((H, W, 4), np.float32)
        o = po.Sppm(W, H, 40 + seed)
        done += 1
        for f in range(frames):
            gpu.sppm_frames(1)
            dpho = gpu.sppm_download()[1]
            o.frames(sv, cam, rng, acc, 1)
            opho = o.download()[1]
            a = dpho.view(np.uint8).reshape(len(dpho), -1)
            b = opho.view(np.uint8).reshape(len(opho), -1)
            diff = np.flatnonzero((a != b).any(axis=1))
            if len(diff) == 0:
                continue
            bad += 1
            again = gpu.sppm_download()[1].view(np.uint8).reshape(len(dpho), -1)
            zero = (a[diff] == 0).all(axis=1)
            print(f"MISMATCH seed {seed} frame {f} (worker {index} of {procs}): {len(diff)} records; indices {diff[:16].tolist()}; wavefronts "
                  f"{sorted(set((diff // 64).tolist()))[:16]}; all-zero on the device: {int(zero.sum())} of {len(diff)}; a second download of the same "
                  f"buffer {'AGREES with the first' if (again[diff] == a[diff]).all() else 'DIFFERS from the first (the transfer, not the kernel)'}; "
                  f"of it matches the oracle: {int((again[diff] == b[diff]).all(axis=1).sum())}", flush=True)
            for i in diff[:4]:
                lo, hi = max(0, i - 1), min(len(dpho), i + 2)
                print(f"   record {i}: gpu {dpho[i]}  oracle {opho[i]}  neighbours differ: {[(int(j), bool((a[j] != b[j]).any())) for j in range(lo, hi)]}", flush=True)
            break
    print(f"worker {index}: {done} scenes, {bad} with a mismatch", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("first", type=int); ap.add_argument("last", type=int)
    ap.add_argument("--procs", type=int, default=16)
    ap.add_argument("--serialize", action="store_true"); ap.add_argument("--serial-camera", action="store_true")
    ap.add_argument("--worker", type=int, default=-1)
    a = ap.parse_args()
    if a.worker >= 0:
        sys.exit(1 if worker(a.first, a.last, a.worker, a.procs, a.serial_camera) else 0)
    env = dict(os.environ)
    if a.serialize:
        env.update(AMD_SERIALIZE_KERNEL="3", AMD_SERIALIZE_COPY="3")
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(a.first), str(a.last), "--procs", str(a.procs), "--worker", str(i)] +
                              (["--serial-camera"] if a.serial_camera else []), env=env) for i in range(a.procs)]
    rc = [p.wait() for p in procs]
    print(f"seeds {a.first}..{a.last - 1} x {3} frames, {a.procs} processes on one GPU, serialize={a.serialize} serial_camera={a.serial_camera}: "
          f"{sum(1 for r in rc if r)} workers saw a mismatch")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""A/B of two (or more) builds of libtracer_amd.so on the same box: box-to-box variance between gpurun calls is
+-2.5 %, more than most kernel changes, so the variants are timed alternately (A B A B ...) in child processes.

    python tools/ab_bench.py [--config 2|3|4|volume] [--rounds 3] libA.so libB.so ...

Each child loads exactly one library (device.lib_path is pointed at it before the first call), warms the adaptive
launch order up, then reports the mean kernel time of `--steps` launches."""
import argparse, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, config, steps, spp_arg=0, warm=3):
    if "@" in path:                        # lib.so@NAME=VALUE[@NAME=VALUE]: the same build under other environment knobs
        path, *envs = path.split("@")
        for e in envs:
            k, v = e.split("=", 1)
            os.environ[k] = v
    os.environ["TRC_AMD_LIB"] = os.path.abspath(path)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import workloads as wlmod
    from tracer_amd import device
    wl = wlmod.make(config)
    spp = spp_arg or {"2": 64, "3": 32, "4": 32, "volume": 16}[config]
    t = device.Tracer(0)
    wlmod.setup(t, wl)
    for i in range(warm):
        t.seed(0x5EED0000 + i); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"])
    t.synchronize(); t.reset_stats()
    for i in range(steps):
        t.seed(0x5EED0100 + i); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"])
    t.synchronize()
    s = t.stats()
    print(f"{s.kernel_ms / steps:.3f} {s.rays / steps:.0f}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2")
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--child", default=None)
    ap.add_argument("--spp", type=int, default=0, help="samples per launch (default: 64 / 32 / 32 / 16 for config 2 / 3 / 4 / volume)")
    ap.add_argument("--warm", type=int, default=3, help="untimed launches before the timed ones (the adaptive order and split plan settle)")
    ap.add_argument("libs", nargs="*")
    a = ap.parse_args()
    if a.child:
        child(a.child, a.config, a.steps, a.spp, a.warm)
        return
    res = {p: [] for p in a.libs}
    for r in range(a.rounds):
        for p in a.libs:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", p, "--config", a.config,
                                  "--steps", str(a.steps), "--spp", str(a.spp), "--warm", str(a.warm)], capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(p, "FAILED", out.stderr[-400:]); continue
            res[p].append(float(out.stdout.split()[0]))
    for p, v in res.items():
        if v:
            print(f"{p}: " + " ".join(f"{x:.3f}" for x in v) + f"  | min {min(v):.3f} mean {sum(v) / len(v):.3f} ms")


if __name__ == "__main__":
    main()

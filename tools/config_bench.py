#!/usr/bin/env python3
"""One BASELINE configuration at 1920x1080 on one GPU: kernel time (HIP events), exact ray count, algorithmic bytes
per ray (instrumented launch) and the three SURVEY 8(d) numbers' inputs, as ONE JSON line.  Also the workload the
rocprofv3 passes of tools/pmc_collect.sh run.

    python3 tools/config_bench.py --config 2|3|4|5|volume [--spp N] [--steps K] [--warmup W]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import workloads as wlmod
from tracer_amd.device import Tracer

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="2")
ap.add_argument("--spp", type=int, default=0)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--frames", type=int, default=64, help="config 5: SPPM frames per step")
ap.add_argument("--no-cold", action="store_true", help="profiling runs: the first launch as ONE pass (knob no_cold_probe), so every launch of the kernel in a trace is a whole one")
a = ap.parse_args()
W, H = wlmod.W, wlmod.H
t = Tracer(0)
if a.config == "5":
    wl = wlmod.make("2")
    wlmod.setup(t, wl)
    t.seed(1); t.sppm_init(2)
    t.sppm_frames(2); t.synchronize(); t.reset_stats()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        t.sppm_frames(a.frames)
    t.synchronize()
    dt = time.perf_counter() - t0
    s = t.stats()
    # rooflines of the SPPM frame: one object per pass, from the newest committed PMC summaries (profiles/rNN/pmc_config5_*.json;
    # numbers only when they were collected on the running library), the dominant pass (refine) as `roofline`
    import glob, re
    sys.path.insert(0, wlmod.ROOT)
    import bench
    mine = bench.lib_source_hash()
    passes = {}
    for k in ("k_sppm_refine", "k_sppm_camera", "k_sppm_photon", "k_sppm_table"):
        files = sorted(glob.glob(os.path.join(wlmod.ROOT, "profiles", "r*", f"pmc_config5_{k}.json")),
                       key=lambda f: int(re.search(r"r(\d+)$", os.path.basename(os.path.dirname(f))).group(1)))
        if not files:
            passes[k] = {"bound": "valu-issue", "frac": None, "stale": "no committed PMC summary of this pass", "lib_source_hash": mine}
            continue
        pm = json.load(open(files[-1])); rel = os.path.relpath(files[-1], wlmod.ROOT)
        if pm.get("lib_source_hash") != mine:
            passes[k] = {"bound": "valu-issue", "frac": None, "lib_source_hash": mine,
                         "stale": f"{rel} was collected on library {pm.get('lib_source_hash')}; running {mine}"}
        else:
            passes[k] = bench.valu_roofline(pm, None, rel, mine)
    print(json.dumps({"config": "config 5: SPPM, Cornell + 12 spheres, 512^2 photons per frame", "frames": a.frames * a.steps,
                      "wall_ms_per_frame": round(dt / (a.frames * a.steps) * 1e3, 4), "rays": s.rays,
                      "mrays_per_s": round(s.rays / dt / 1e6, 1), "kernel": "k_sppm_",
                      "roofline": passes["k_sppm_refine"], "roofline_per_pass": passes}))
    sys.exit(0)
wl = wlmod.make(a.config)
spp = a.spp or wl["spp"]
wlmod.setup(t, wl)
if a.no_cold:
    t.debug_set("no_cold_probe", 1)
t.seed(0x5EED0000); t.reset_stats()
t.render(spp=spp, integrator=wl["integrator"], collect_stats=True); t.synchronize()
s1 = t.stats()
bytes_launch = wlmod.algorithmic_bytes(s1, W * H)
for i in range(a.warmup):
    t.seed(0x5EED0000); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"])
t.synchronize(); t.reset_stats()
for i in range(a.steps):
    t.seed(0x5EED0000); t.clear_accum(); t.render(spp=spp, integrator=wl["integrator"])
t.synchronize()
s = t.stats()
ms = s.kernel_ms / a.steps
n_leaf = s1.n_leaf_sphere + s1.n_leaf_square + s1.n_leaf_cube + s1.n_leaf_triangle


def rooflines():
    """the two objects of bench.py's line for this configuration: `roofline` = the binding VALU-issue bound from the newest
    committed PMC summary of this kernel (profiles/rNN/pmc_config<c>.json: collected on a 32-spp launch of the same kernel;
    only when its source hash is the running library's), `roofline_hbm_algorithmic` = the north_star's figure, live"""
    import glob, re
    sys.path.insert(0, wlmod.ROOT)
    import bench
    alg = {"bound": "hbm (algorithmic bytes, not a physical bound)", "achieved": round(bytes_launch / ms / 1e6, 1), "peak": bench.HBM_PEAK_GBS,
           "unit": "GB/s", "frac": round(bytes_launch / ms / 1e6 / bench.HBM_PEAK_GBS, 4), "bytes_per_ray": round(bytes_launch / s1.rays, 1)}
    files = sorted(glob.glob(os.path.join(wlmod.ROOT, "profiles", "r*", f"pmc_{'volume' if a.config == 'volume' else 'config' + a.config}.json")),
                   key=lambda f: int(re.search(r"r(\d+)$", os.path.basename(os.path.dirname(f))).group(1)))
    mine = bench.lib_source_hash()
    none = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "VALU wave-instructions/launch", "frac": None, "traffic": None, "lib_source_hash": mine}
    if not files:
        return dict(none, stale="no committed PMC summary of this configuration"), alg
    pm = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], wlmod.ROOT)
    if pm.get("lib_source_hash") != mine or not str(pm.get("kernel", "")).startswith(wl["kernel"]):
        return dict(none, stale=f"{rel} was collected on library {pm.get('lib_source_hash')} / kernel {pm.get('kernel')!r}; running {mine} / {wl['kernel']!r}"), alg
    r = bench.valu_roofline(pm, None, rel, mine)
    r["pmc_launch"] = f"{pm.get('workload_line', {}).get('spp')} spp"
    return r, alg


roof, roof_alg = rooflines()
print(json.dumps({
    "config": wl["what"], "spp": spp, "kernel": wl["kernel"], "kernel_ms": round(ms, 3), "rays_per_launch": s1.rays,
    "mrays_per_s": round(s1.rays / ms / 1e3, 1), "mpaths_per_s": round(W * H * spp / ms / 1e3, 1),
    "bvh_nodes": wl["scene"].view.n_bvh, "triangles": wl["scene"].view.n_index // 3,
    "algorithmic_bytes_per_launch": bytes_launch, "bytes_per_ray": round(bytes_launch / s1.rays, 1),
    "algorithmic_gbs": round(bytes_launch / ms / 1e6, 1), "hbm_roofline_frac": round(bytes_launch / ms / 1e6 / 8000.0, 4),
    "descend_per_ray": round(s1.n_descend / s1.rays, 2), "leaf_tests_per_ray": round(n_leaf / s1.rays, 2),
    "roofline": roof, "roofline_hbm_algorithmic": roof_alg}))

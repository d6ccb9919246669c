#!/usr/bin/env python3
"""bench.py -- Mrays/s of the path-tracing hot path on BASELINE config 2.

Workload (BASELINE.json configs[1]): RT_Metal Cornell box + the 12 spheres, 1920x1080, 64 spp,
tracePath, depth 8, synthetic (scene from the reference's constants, per-pixel PCG32 seeds).

A "step" = one pass of the hot path over one batch: re-seed the RNG texture (so every step is the SAME
work), render all 64 samples of this rank's pixel tiles, and -- for N > 1 -- compose the frame with one
RCCL reduce of the accumulation buffer to rank 0.  Inputs are resident in HBM before the timed region.
`rays` = Scene::hit invocations, counted exactly by the kernel.

  python bench.py [--gpus N --steps K --warmup W] [--scaling strong|weak|samples]

N > 1 is launched either by the driver (python -m torch.distributed.run ... bench.py --gpus N: RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or by bench.py itself: with WORLD_SIZE unset, `--gpus N` spawns N fresh
child processes of this script (before anything touches the GPU) and relays rank 0's JSON line.  Either way the ranks
find each other, synchronise and reduce their timings over plain TCP sockets (tracer_amd/socket_group.py): bench.py
imports no PyTorch at any N -- the frame is composed by the library's own RCCL reduce.

Three multi-GPU workloads, ALL measured in every N > 1 run (the primary one fills the top-level fields, the others
are reported under "other_scaling"):
  strong (default; the metric as BASELINE.json names it): the ONE 1920x1080x64spp frame, its 16x16 pixel tiles owned
      round-robin (tx + ty) % N, every rank renders all 64 samples of its tiles, one ncclReduce(sum) composes the
      frame on rank 0.  A pixel's 64 samples are a sequential RNG chain (Render.metal:545-557), so the GPU drains on
      its slowest 8x8 blocks however few blocks it owns: expect well below linear.
  weak: N such views stacked into one 1920 x (1080 N) frame (trc_params.view_height), same ownership rule, i.e. one
      view's worth of tiles per GPU -- the N = 1 work per GPU; metric string says so.
  samples: the split that shortens the chains (include/tracer_abi.h, "sample sharding"): every GPU renders the WHOLE
      1920x1080 frame with 64 / N samples per pixel from its own seed trc_shard_seed(seed, rank); the frame is the
      rank-ordered sum of the N accumulators / N (all-to-all of pixel slices + ordered fold + gather to rank 0:
      trc_group_compose_samples_async) -- 64 samples per pixel like the named frame, but another sample set than one
      GPU's (bit-defined: oracle/pyoracle.py::render_sample_sharded).  --sample-groups S < N makes it S seeds x N / S
      tile ranks.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the roofline definitions).

Self-validation (round 6): the frame the timed steps rendered / composed is downloaded once, OUTSIDE the timed region, and its CRC32 (at
N = 1 also the ray count) is held to the CPU oracle's committed value for that frame (tests/golden/bench_goldens.json, written by
tests/golden/make_bench_goldens.py): `composed_crc_ok`; false ends the run non-zero after the line.  At N = 1 the line also carries
`other_configs` -- BASELINE configs 3, 4, 5, the traceVolume scene and the 4 M-triangle HBM operating point, two timed steps each on a
context of its own, each frame checked the same way, each with the roofline of its kernel's committed PMC summary -- and `hbm_point`
(north_star's ">= 40 % of the HBM roofline" at the one operating point where it can be asked: not met).  Optional legs never take the
headline down: each runs inside try / except and under a watchdog.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, SPP, DEPTH = 1920, 1080, 64, 8
SEED = 0x5EED0000
SETTLE_LAUNCHES = 8          # untimed launches before the W warm-up steps: the adaptive order / split plan of the block list settle
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
N_SIMD = 1024                # 256 CUs x 4 SIMD-32; a wave64 VALU instruction issues over 2 cycles (ibid., line 54)
KERNEL = ("k_render_dense", "k_render<true, false, 0")   # rocprof name prefixes of the production tracePath kernel on an LDS-resident
                                                         # scene: whole-frame launch lists run it at 6 waves/SIMD (k_render_dense), shares at 5


def algorithmic_bytes(st, n_pixels):
    """bytes the reference's algorithm moves for this work (NOT cache-line traffic).
    per ray 88*N_descend + 24*N_return + leaf bytes; per shaded hit 64 B material; pixel state 64 B."""
    leaf = (20 * st.n_leaf_sphere + 32 * st.n_leaf_square + 100 * st.n_leaf_cube + 128 * st.n_hit_cube +
            48 * st.n_leaf_triangle + 60 * st.n_hit_triangle)
    return 88 * st.n_descend + 24 * st.n_return + leaf + 64 * st.shaded + 64 * n_pixels


def cpu_baseline(scene, cam, budget_s=20.0):
    """The CPU oracle (a port of the same algorithm) on a bounded tile subsample of the SAME workload."""
    from oracle import pyoracle as po
    from tracer_amd import host
    cores = os.cpu_count() or 1
    rng = host.fill_rng(SEED, W, H)
    # calibrate on 1 tile in 256, then pick the subsample that fits the budget
    t0 = time.perf_counter()
    _, st = po.render(scene.view, cam, W, H, rng, spp=SPP, max_depth=DEPTH, tile_rank=0, tile_nranks=256,
                      n_threads=cores)
    dt = time.perf_counter() - t0
    rate = st.rays / dt
    full_rays_est = st.rays * 256
    nranks = 256
    while nranks > 1 and (full_rays_est / (nranks // 2)) / rate < budget_s:
        nranks //= 2
    if nranks != 256:
        rng = host.fill_rng(SEED, W, H)
        t0 = time.perf_counter()
        _, st = po.render(scene.view, cam, W, H, rng, spp=SPP, max_depth=DEPTH, tile_rank=0, tile_nranks=nranks,
                          n_threads=cores)
        dt = time.perf_counter() - t0
    return {"value": round(st.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"oracle/liboracle.so (scalar C++ restatement), tiles (tx+ty)%{nranks}==0 of the same "
                      f"1920x1080x64spp frame: {st.paths} paths, {st.rays} rays in {dt:.2f} s on {cores} threads"}


def cpu_rt_weekend():
    """BASELINE config 1: the reference's RT_Weekend CPU path (C++ restatement, oracle/cpu_baseline.cpp)
    on the Cornell box at 400x400x16 spp, all host cores, row bands like main.swift:77-87."""
    exe = os.path.join(ROOT, "oracle", "cpu_baseline")
    if not os.path.exists(exe):
        return None
    out = subprocess.check_output([exe, "--scene", "cornell", "--width", "400", "--height", "400", "--spp", "16"],
                                  text=True, timeout=600)
    r = json.loads(out.strip().splitlines()[-1])
    return {"value": r["mrays_per_s"], "unit": "Mrays/s", "cores": r["threads"], "kind": "port",
            "sample": f"RT_Weekend recursive tracer restated in C++ (linear-scan HittableList), RT_Nextweek Cornell "
                      f"box 400x400x16spp: {r['rays']} rays in {r['seconds']:.2f} s, {r['mpaths_per_s']} Mpaths/s"}


def load_goldens():
    """tests/golden/bench_goldens.json: the ORACLE's ray counts and frame CRC32s at the sizes BASELINE.json names
    (tests/golden/make_bench_goldens.py) -- what this script holds its own frames to, outside the timed regions."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_goldens.json")) as f:
            return json.load(f)
    except OSError:
        return {}


def frame_crc(a):
    import zlib
    import numpy as np
    return zlib.crc32(np.ascontiguousarray(a).view(np.uint8).tobytes()) & 0xFFFFFFFF


def check_against(gold, key, crc=None, rays=None, extra=None):
    """{"golden": key, "crc_ok": ..., "rays_ok": ...}: None where the committed file has no such entry"""
    g = gold.get(key)
    out = {"golden": key if g else None}
    if g is None:
        out["missing"] = f"tests/golden/bench_goldens.json has no '{key}' (tests/golden/make_bench_goldens.py)"
        return out
    if crc is not None:
        out["crc_ok"] = bool(g.get("crc_accum") == crc)
    if rays is not None and "rays" in g:
        out["rays_ok"] = bool(int(g["rays"]) == int(rays))
    for k, v in (extra or {}).items():
        out[k + "_ok"] = bool(g.get(k) == v)
    return out


def config_roofline(name, kernel_prefix, kernel_ms=None):
    """the VALU-issue roofline of another configuration's kernel from its newest committed PMC summary
    (profiles/rNN/pmc_<name>.json), under the same rule as the headline's: only when collected on the running library"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"pmc_{name}.json")),
                   key=lambda f: int(re.search(r"r(\d+)$", os.path.basename(os.path.dirname(f))).group(1)))
    mine = lib_source_hash()
    none = {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "VALU wave-instructions/launch", "frac": None, "traffic": None,
            "lib_source_hash": mine}
    if not files:
        return dict(none, stale=f"no profiles/r*/pmc_{name}.json"), None
    pm = json.load(open(files[-1]))
    rel = os.path.relpath(files[-1], ROOT)
    if pm.get("lib_source_hash") != mine or not str(pm.get("kernel", "")).startswith(kernel_prefix):
        return dict(none, stale=f"{rel} was collected on library {pm.get('lib_source_hash')} / kernel {pm.get('kernel')!r}; running {mine} / {kernel_prefix!r}"), None
    r = valu_roofline(pm, None, rel, mine)
    r["pmc_launch"] = f"{pm.get('workload_line', {}).get('spp')} spp per launch"
    return r, pm


OTHER_CONFIGS = ("config3", "config4", "config5_sppm", "volume", "hbm_point")


def other_config_leg(key, device, steps=2):
    """One of BASELINE's other configurations (or the HBM operating point) on a context of its own: `steps` timed steps after the
    launch order has settled, the frame held to the oracle's committed ray count / CRC (tests/golden/bench_goldens.json)."""
    import numpy as np
    from tracer_amd import abi, host
    from tracer_amd.device import Tracer
    gold = load_goldens()
    cam = host.prepare_camera(W, H)
    t = Tracer(device)
    try:
        if key == "config5_sppm":
            scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
            t.upload_scene(scene.view); t.set_camera(cam); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
            frames, wall, rays = 64, [], 0
            for i in range(steps + 1):                           # the first pass is the warm-up
                t.clear_accum(); t.seed(1); t.sppm_init(2); t.synchronize(); t.reset_stats()
                t0 = time.perf_counter()
                t.sppm_frames(frames); t.synchronize()
                if i:
                    wall.append(time.perf_counter() - t0)
                rays = t.stats().rays
            cx = t.sppm_download()[4]
            ms = sum(wall) / len(wall) * 1e3
            roofs = {k: config_roofline("config5_" + k, k)[0] for k in ("k_sppm_refine", "k_sppm_camera", "k_sppm_photon", "k_sppm_table")}
            return {"config": "BASELINE config 5: SPPM photon pass, Cornell + 12 spheres, 512^2 photons per frame, 64 frames at 1920x1080",
                    "value": round(rays / ms / 1e3, 1), "unit": "Mrays/s", "ms_per_step": round(ms, 3), "ms_per_frame": round(ms / frames, 4),
                    "rays_per_step": int(rays), "steps": steps, "frames_per_step": frames,
                    "oracle_check": check_against(gold, key, crc=frame_crc(t.download_accum()),
                                                  extra={"totalPhotonSum": int(cx.totalPhotonSum), "frame_count": int(cx.frame_count)}),
                    "roofline": roofs["k_sppm_refine"], "roofline_per_pass": roofs}
        density = None
        if key == "config3":
            scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("coatball")), abi.INTEGRATOR_MIS, 256
            what, pmc, kern = "BASELINE config 3: Cornell + coatball.obj (46 816 triangles), traceMIS, 1920x1080x256spp", "config3", "k_render_pwg<1, false>"
        elif key == "config4":
            scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(8, 80.0)), abi.INTEGRATOR_PATH, 256
            what, pmc, kern = "BASELINE config 4: Cornell + teapot.obj x 64 (1 005 056 triangles), tracePath, 1920x1080x256spp", "config4", "k_render_pwg<0, false>"
        elif key == "volume":
            scene, integ, spp = host.HostScene(abi.SCENE_CORNELL_VOLUME, host.Mesh.golden("coatball")), abi.INTEGRATOR_VOLUME, 64
            density = host.make_cloud()
            what, pmc, kern = "traceVolume (SURVEY 8f-3): Cornell + cloud container (100x100x40 grid) + coatball.obj as glass, 1920x1080x64spp", "volume", "k_render_pwg<2, false>"
        else:                                                     # the operating point where HBM is the memory
            scene = host.HostScene(abi.SCENE_CORNELL_MESH, host.Mesh.golden("teapot").replicate(16, 80.0), analytic_leaves_only=True)
            integ, spp = abi.INTEGRATOR_PATH, 32
            what, pmc, kern = "Cornell + teapot.obj x 256 (4 020 224 triangles, 708 MB: beyond the 256 MiB Infinity Cache), tracePath, 1920x1080x32spp", "hbm_point", "k_render_pwg<0, false>"
        if key == "hbm_point":
            t.upload_scene_device(scene.view, abi.TREE_SAH | abi.TREE_TRIANGLE_LEAVES)
        else:
            t.upload_scene(scene.view)
        if density is not None:
            t.upload_density(host.density_info(density), density)
        t.set_camera(cam); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)

        def step(n):
            t.seed(SEED); t.clear_accum(); t.render(spp=n, max_depth=DEPTH, integrator=integ)
        # the launch order and split plan settle on per-SAMPLE costs, so short launches do most of it: 8 x 32 spp, then
        # three launches of the named length (the split plan's K keeps ramping for a few launches on config 3), then the timed ones
        for _ in range(SETTLE_LAUNCHES):
            step(min(32, spp))
        for _ in range(3):
            step(spp)
        t.synchronize(); t.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            step(spp)
        t.synchronize()
        wall = (time.perf_counter() - t0) / steps * 1e3
        st = t.stats()
        rays, kernel_ms = st.rays // steps, st.kernel_ms / max(1, st.launches)
        out = {"config": what, "value": round(rays / wall / 1e3, 1), "unit": "Mrays/s", "ms_per_step": round(wall, 3), "kernel_ms": round(kernel_ms, 3),
               "rays_per_step": int(rays), "steps": steps, "spp": spp, "kernel": kern,
               "mpaths_per_s": round(W * H * spp / wall / 1e3, 1), "triangles": int(scene.view.n_index // 3)}
        roof, pm = config_roofline(pmc, kern)
        out["roofline"] = roof
        if key == "hbm_point":
            # north_star's ">= 40 % of the HBM roofline" can only be asked here (the other scenes live in LDS / L2 / Infinity Cache)
            frac = None if pm is None else pm.get("l2_miss_frac_of_hbm_peak")
            share = None if pm is None or not pm.get("l2_miss_bytes_per_launch") else round(pm["l2_miss_write_bytes_per_launch"] / pm["l2_miss_bytes_per_launch"], 3)
            out["hbm"] = {"mrays": out["value"], "l2_miss_frac_of_hbm_peak": frac, "spill_share": share, "target_frac": 0.40,
                          "target_met": None if frac is None else bool(frac >= 0.40),
                          "what": "L2-miss traffic of the 4 M-triangle scene (all of it HBM: 708 MB against a 256 MiB Infinity Cache) over 8 TB/s, from the "
                                  "committed PMC summary; spill_share = its write half, which is register spills (the frame's own writes are 66 MB). "
                                  "north_star's >= 40 % is NOT met and is not what bounds this path: dependent 64-byte gathers at full occupancy are "
                                  "latency-bound (DESIGN 4.4)"}
        else:
            out["oracle_check"] = check_against(gold, key, crc=frame_crc(t.download_accum()), rays=rays)
        return out
    finally:
        t.close()


def fast_math_leg(scene, cam, steps):
    """The same K steps on libtracer_amd_fast.so (the sources under fast-math rules, like the reference's MTL_FAST_MATH
    shaders).  Reported beside the headline, never as it: only the exact build is comparable with the oracle."""
    from tracer_amd import abi, device
    if not os.path.exists(device.fast_lib_path()):
        return None
    t = device.Tracer(0, fast_math=True)
    try:
        t.upload_scene(scene.view); t.set_camera(cam); t.set_environment((0.0, 0.0, 0.0)); t.resize(W, H)
        for _ in range(2):
            t.seed(SEED); t.render(spp=SPP, max_depth=DEPTH, integrator=abi.INTEGRATOR_PATH)
        t.synchronize(); t.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps):
            t.seed(SEED); t.render(spp=SPP, max_depth=DEPTH, integrator=abi.INTEGRATOR_PATH)
        t.synchronize()
        dt = time.perf_counter() - t0
        st = t.stats()
        return {"library": "tracer_amd/lib/libtracer_amd_fast.so", "value": round(st.rays / dt / 1e6, 2), "unit": "Mrays/s",
                "ms_per_step": round(dt / steps * 1e3, 3), "kernel_ms": round(st.kernel_ms / max(1, st.launches), 3),
                "rays_per_step": int(st.rays // steps),
                "what": "approximate division / sqrt, FMA contraction, denormals flushed; agrees with the exact build "
                        "statistically (tests/test_gpu_fast_math.py), not bit for bit -- not the headline"}
    finally:
        t.close()


def _code_only(text):
    """C / C++ source without comments and with white space collapsed: what the compiler sees of it."""
    import re
    pat = re.compile(r'//[^\n]*|/\*.*?\*/|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])*\'', re.S)
    text = pat.sub(lambda m: " " if m.group(0).startswith("/") else m.group(0), text)
    return " ".join(text.split())


def lib_source_hash():
    """Identity of the device library: sha256 over the CODE of the sources it is built from (the .so itself is not
    byte-reproducible across builds).  Comments and white space are stripped first, so that a comment edit does not
    invalidate a committed PMC summary (round 3 regenerated its profile set five times for that); anything the compiler
    sees does.  tools/pmc_summary.py stores the same value next to the counters it collects."""
    h = hashlib.sha256()
    files = []
    for d in ("tracer_amd/csrc", "include"):
        files += [os.path.join(d, f) for f in sorted(os.listdir(os.path.join(ROOT, d)))]
    for f in files:
        h.update(f.encode())
        h.update(_code_only(open(os.path.join(ROOT, f), "r", errors="replace").read()).encode())
    # ... and the options it is compiled with (the Makefile's flag lines, per-translation-unit ones included)
    try:
        for ln in open(os.path.join(ROOT, "Makefile"), "r", errors="replace"):
            if ln.split(":=")[0].strip() in ("HIPFLAGS",) or ln.startswith("EXTRA_") or ln.lstrip().startswith("-Wall -Wno-unused-function"):
                h.update(" ".join(ln.split()).encode())
    except OSError:
        pass
    return h.hexdigest()[:16]


def pmc_summary():
    """Newest committed PMC summary of the bench kernel (profiles/r*/pmc_config2.json, written by
    tools/pmc_summary.py from rocprofv3 --pmc passes of THIS script); None if absent."""
    import glob
    import re

    def round_of(path):                  # profiles/r03 < profiles/r10: by number, not by name
        m = re.search(r"r(\d+)$", os.path.basename(os.path.dirname(path)))
        return int(m.group(1)) if m else -1
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_config2.json")), key=round_of)
    if not files:
        return None
    p = json.load(open(files[-1]))
    p["_file"] = os.path.relpath(files[-1], ROOT)
    return p


def valu_roofline(p, kernel_ms=None, source=None, lib_hash=None):
    """The binding bound of these kernels: VALU issue.  A wave64 VALU instruction occupies a SIMD-32 for 2 cycles
    (MI355X_MICROARCH.md), so a launch of `cyc` shader cycles can issue at most N_SIMD * cyc / 2 of them.
    achieved / peak in wave-instructions per launch, from a PMC summary `p` (tools/pmc_summary.py)."""
    cyc = p["GRBM_GUI_ACTIVE"] / 8.0                       # summed over the 8 XCDs
    # bytes the L2s moved to / from the fabric per launch (tools/pmc_summary.py: sized TCC_EA0 request counters; Infinity-Cache
    # hits INCLUDED, so an upper bound of the HBM bytes that equals them only for working sets far beyond 256 MiB)
    t = p.get("l2_miss_bytes_per_launch", p.get("hbm_bytes_per_launch"))
    traffic = int(t) if t is not None else None
    ms = kernel_ms if kernel_ms is not None else p.get("kernel_ms")
    return {
        "bound": "valu-issue", "achieved": int(p["SQ_INSTS_VALU"]), "peak": int(N_SIMD * cyc / 2.0),
        "unit": "VALU wave-instructions/launch", "frac": round(p["SQ_INSTS_VALU"] * 2.0 / (N_SIMD * cyc), 4),
        "traffic": traffic,
        "traffic_what": "L2-miss bytes per launch, reads (32 / 64 / 128 B x TCC_EA0_RDREQ_*) + writes (WRITE_SIZE); fabric side: Infinity-Cache hits included",
        "l2_miss_traffic_frac_of_hbm_peak": None if traffic is None or not ms else round(traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
        "lane_utilisation": round(p["SQ_THREAD_CYCLES_VALU"] / (64.0 * p["SQ_INSTS_VALU"]), 4),
        "useful_lane_frac": round(p["SQ_THREAD_CYCLES_VALU"] / 64.0 * 2.0 / (N_SIMD * cyc), 4),
        "kernel": p["kernel"], "shader_cycles_per_launch": int(cyc), "pmc_kernel_ms": p.get("kernel_ms"),
        "source": source, "lib_source_hash": lib_hash,
    }


def roofline_from_pmc(kernel_ms):
    """`roofline` of the bench line from the committed PMC summary -- only when it was collected on the library that is
    running (same source hash, same kernel name); otherwise the object says why it carries no number."""
    p = pmc_summary()
    mine = lib_source_hash()
    if p is None:
        return {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "VALU wave-instructions/launch", "frac": None,
                "traffic": None, "stale": "no profiles/r*/pmc_config2.json (tools/pmc_collect.sh)", "lib_source_hash": mine}
    if p.get("lib_source_hash") != mine or not str(p.get("kernel", "")).startswith(KERNEL):
        return {"bound": "valu-issue", "achieved": None, "peak": None, "unit": "VALU wave-instructions/launch", "frac": None,
                "traffic": None, "lib_source_hash": mine,
                "stale": f"{p['_file']} was collected on library {p.get('lib_source_hash')} / kernel {p.get('kernel')!r}; "
                         f"running {mine} / {KERNEL!r}"}
    return valu_roofline(p, kernel_ms, p["_file"], mine)


def _with_c_stdout_on_stderr(fn):
    """Run fn() with file descriptor 1 pointing at stderr, flushing C stdio before restoring it."""
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        fn()
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """GPUs of this node WITHOUT initialising HIP in this process (a process that touched the GPU must not start
    the ranks): KFD topology nodes with SIMDs; ROCR/HIP_VISIBLE_DEVICES narrow it."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for d in os.listdir(base):
            try:
                props = open(os.path.join(base, d, "properties")).read()
            except OSError:
                continue
            for line in props.splitlines():
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except OSError:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            n = min(n, len([x for x in v.split(",") if x.strip()])) if n else len([x for x in v.split(",") if x.strip()])
    return n


def pick_device(local_rank, n_visible):
    """HIP device index of a rank: its own GPU when the rank sees all of the node's, index 0 when the launcher isolated it
    with a HIP_VISIBLE_DEVICES of one id, round-robin when there are more ranks than GPUs"""
    return local_rank % max(1, n_visible)


def compose_transport(bus_ids):
    """How the ranks compose, decided from what they actually HOLD: every rank publishes the PCI bus id of the device its
    context sits on (trc_device_pci_bus_id).  All different -> each rank has a GPU of its own: RCCL over xGMI.  Any two equal
    -> ranks share a device, where RCCL refuses the second rank: the host-staged collectives table.  (Counting
    HIP_VISIBLE_DEVICES entries cannot tell: a launcher that hands every rank ONE id makes each rank see one device.)"""
    ids = [str(b).strip().lower() for b in bus_ids]
    if not ids or any(not b for b in ids):
        return "table"
    return "rccl" if len(set(ids)) == len(ids) else "table"


def launch_ranks(n, argv, timeout_s=1800.0):
    """`python bench.py --gpus N` without a launcher: N fresh children of this script, one per rank, started before
    this process has touched the GPU (never an exec from a process that has).  Rank 0's stdout is relayed.  A rank that
    dies takes the others with it (they would wait in a barrier for ever): the children are OUR processes, ended by
    PID, and the launcher returns the failure instead of hanging."""
    import tempfile
    import threading
    port = _free_port()
    env0 = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    out0 = tempfile.TemporaryFile(mode="w+")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, cwd=ROOT,
                                      stdout=out0 if r == 0 else sys.stderr))
    deadline = time.monotonic() + timeout_s
    failed = None
    while any(p.poll() is None for p in procs):
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad or time.monotonic() > deadline:
            failed = bad[0].returncode if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            t_end = time.monotonic() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, t_end - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    if failed is not None:
        sys.stderr.write(f"bench.py: a rank failed (exit code {failed}); the other ranks were stopped\n")
        return abs(failed) or 1
    return max(abs(p.returncode) for p in procs)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaling", choices=("strong", "weak", "samples"), default="strong",
                    help="which N > 1 workload fills the top-level fields (the others go to other_scaling)")
    ap.add_argument("--sample-groups", type=int, default=0,
                    help="--scaling samples: number of seeds S the 64 samples are split over (default N); N / S tile ranks share each seed's frame")
    ap.add_argument("--other-timeout", type=float, default=600.0,
                    help="seconds an optional leg (other_scaling) may take before the line goes out without it")
    ap.add_argument("--rccl-timeout", type=float, default=300.0,
                    help="seconds a rank waits inside the RCCL bring-up before it exits non-zero (a peer that died would leave it there for ever)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: measure only the primary workload")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="N = 1: skip the other_configs legs (configs 3 / 4 / 5, traceVolume, the HBM point; profiling runs)")
    ap.add_argument("--no-fast-math", action="store_true",
                    help="skip the fast_math_variant leg (profiling runs: both builds name their kernels alike)")
    ap.add_argument("--no-cold", action="store_true",
                    help="profiling runs: skip the cold legs and launch the first frame as ONE pass (knob no_cold_probe), so that "
                         "every launch of the render kernel in a rocprofv3 trace is a whole 64-spp launch")
    ap.add_argument("--force-group", action="store_true",
                    help="exercise the rendezvous + RCCL compose path even with one rank (plumbing check)")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)     # test hook: this rank exits with 3 at once
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="ranks only rendezvous (sockets), exchange their ranks and print the line skeleton: the launcher "
                         "/ relay / reduction plumbing without a GPU (tests/test_bench_helpers.py)")
    args = ap.parse_args(argv)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, argv))

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes); before anything loads HIP
    rank = int(os.environ.get("RANK", "0"))
    if rank == args.fail_rank:
        raise SystemExit(3)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    grouped = world > 1 or args.force_group
    group = None
    if grouped:
        from tracer_amd.socket_group import SocketGroup      # rendezvous / barrier / max-over-ranks over TCP: no PyTorch
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        group = SocketGroup.from_env()

    def reduce_scalar(x, op):
        return x if group is None else group.allreduce_scalar(x, op)

    def gather_list(x):
        return [x] if group is None else group.gather(x)

    if args.rendezvous_only:
        ranks = gather_list(rank)
        total = reduce_scalar(float(rank + 1), "SUM")
        if rank == 0:
            print(json.dumps({"rendezvous_only": True, "n_gpus": world, "ranks": ranks, "sum": total,
                              "scaling": args.scaling, "torch_imported": "torch" in sys.modules}), flush=True)
        if group is not None:
            group.barrier()
            group.close()
        return

    from tracer_amd import abi, host
    from tracer_amd.device import Tracer, group_unique_id

    # Which transport composes: decided from the devices the ranks actually hold (compose_transport), not from counting
    # environment entries.  Ranks sharing a GPU (plumbing run on a 1-GPU box): RCCL refuses two ranks on one device, the compose
    # goes through the collectives table of trc_group_set_collectives (host-staged, sockets) -- the same collective program, the
    # same pipelined two-accumulator compose, every other part of the N-rank path as usual.  The line says "plumbing": ranks
    # time-slicing one GPU measure nothing.
    n_dev = visible_gpus() or 1
    device = pick_device(local_rank, n_dev)
    scene = host.HostScene(abi.SCENE_CORNELL_SPHERES)
    cam = host.prepare_camera(W, H)
    trc = Tracer(device)
    bus_id = trc.pci_bus_id()
    devices = gather_list(bus_id)
    transport = compose_transport(devices) if grouped else "none"
    # TRC_BENCH_NO_RCCL=1: the table even on distinct devices; TRC_BENCH_FORCE_RCCL=1: try RCCL even on a shared device (it
    # refuses: the test of the fallback below)
    if os.environ.get("TRC_BENCH_NO_RCCL") == "1":
        transport = "table"
    elif os.environ.get("TRC_BENCH_FORCE_RCCL") == "1" and grouped:
        transport = "rccl"
    no_rccl = transport != "rccl"
    plumbing = grouped and no_rccl
    trc.upload_scene(scene.view)
    trc.set_camera(cam)
    trc.set_environment((0.0, 0.0, 0.0))
    use_rccl = grouped and not no_rccl
    compose_fallback = None
    if use_rccl:
        # RCCL over xGMI is the compose path.  It has never run with N > 1 on hardware (one GPU per box in the build pool), so a
        # run whose communicators do not come up is not lost: every rank hears of it and all of them compose through the
        # host-staged socket table instead (the line says so in "compose_fallback"; slower, still the named frame).
        # Order matters: ncclCommInitRank and the first collective BLOCK until every rank has entered them, so (1) the ranks
        # agree over the socket group that each of them can load librccl BEFORE any of them enters RCCL, and (2) the bring-up
        # runs under a watchdog that makes a rank EXIT non-zero when it is still inside after --rccl-timeout seconds -- a peer
        # that died in there (device fault, ...) would otherwise leave the others blocked until the launcher's own timeout.
        import threading
        uid, err = None, ""
        try:
            import ctypes
            for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"):
                try:
                    ctypes.CDLL(name)
                    break
                except OSError:
                    continue
            else:
                raise OSError("librccl not loadable")
            if rank == 0:
                uid = group_unique_id()
        except Exception as e:
            err = f"rank {rank}: {e}"
        ready = group.allreduce_scalar(0 if err else 1, "MIN") >= 1
        uid = group.broadcast(uid)
        trc.resize(W, H)

        def init_comm():
            trc.group_init(uid, world, rank)
            trc.clear_accum()
            trc.group_reduce_accum(0)        # first collective: RCCL finishes its lazy set-up here
            trc.synchronize()
        ok = 0
        if ready and uid is not None:
            def give_up():
                sys.stderr.write(f"bench.py: rank {rank} still inside the RCCL bring-up after {args.rccl_timeout:.0f} s "
                                 f"(a peer failed in there?): exiting so that the launcher ends the job\n")
                sys.stderr.flush()
                os._exit(86)
            dog = threading.Timer(args.rccl_timeout, give_up)
            dog.daemon = True
            dog.start()
            try:
                # RCCL prints a version banner through C stdio on fd 1; keep stdout clean for the single JSON line
                _with_c_stdout_on_stderr(init_comm)
                ok = 1
            except Exception as e:
                err = f"rank {rank}: {e}"
            finally:
                dog.cancel()
        if group.allreduce_scalar(ok, "MIN") < 1:
            errs = [e for e in group.gather(err) if e]
            compose_fallback = "RCCL communicator not available (" + "; ".join(errs[:2]) + "): composed through trc_group_set_collectives, host-staged over TCP sockets"
            from tracer_amd.socket_group import SocketCollectives
            coll = SocketCollectives(group)
            trc.set_collectives(coll, world, rank)     # drops whatever communicator this rank did bring up
            use_rccl, plumbing = False, True
    elif plumbing:
        from tracer_amd.socket_group import SocketCollectives
        trc.resize(W, H)
        coll = SocketCollectives(group)
        trc.set_collectives(coll, world, rank)
    composing = use_rccl or plumbing

    goldens = load_goldens()

    def barrier():
        trc.synchronize()                      # hipStreamSynchronize on the render and the compose stream of this rank's device
        if group is not None:
            group.barrier()

    def measure(mode, steps, warmup):
        """K timed steps of one workload; returns the rank-0 view of it (dict) -- every rank must call it."""
        stacked = mode == "weak" and world > 1
        sharded = mode == "samples" and world > 1
        FH = H * world if stacked else H
        trc.resize(W, FH)
        # samples: S seeds x T tile ranks; rank r is tile rank r % T of sample group r // T (tracer_abi.h)
        S = (args.sample_groups or world) if sharded else 1
        if sharded and (world % S or SPP % S):
            raise SystemExit(f"--sample-groups {S}: must divide both the {world} ranks and the {SPP} samples")
        T = world // S if sharded else world
        group_of, tile_rank = (rank // T, rank % T) if sharded else (0, rank)
        if sharded and os.environ.get("TRC_BENCH_REVERSE_GROUPS") == "1":     # test hook: the same shards folded in another rank order --
            group_of = S - 1 - group_of                                       # a composed frame the committed CRC must reject (S >= 3)
        spp_rank = SPP // S

        def step(seed=SEED, collect_stats=False):
            if grouped:
                trc.clear_accum()            # non-owned tiles must be zero for the sum-compose
            trc.seed(abi.shard_seed(seed, group_of))
            trc.render(spp=spp_rank, max_depth=DEPTH, integrator=abi.INTEGRATOR_PATH, frame0=0, tile_rank=tile_rank,
                       tile_nranks=T, collect_stats=collect_stats, view_height=H)
            if composing:                    # overlaps with the next step's render (second accumulator + stream)
                if sharded:
                    trc.group_compose_samples_async(0, S)
                else:
                    trc.group_reduce_accum_async(0)

        # COLD: the first production launch of this block list on this context -- no durations of a previous launch to order
        # or split by.  trc_render runs it as an 8-sample head + the rest planned from the head; the same
        # launch as ONE cold pass (knob no_cold_probe) beside it.  What a host that renders one frame, resizes or moves the
        # camera gets; the headline below is the settled state of a progressive renderer.
        def first_launch(no_probe):
            trc.debug_set("no_cold_probe", 1 if no_probe else 0)      # also forgets what the context knows about the blocks
            barrier()
            trc.reset_stats()
            t = time.perf_counter()
            step()
            barrier()
            wall = (time.perf_counter() - t) * 1e3
            s0 = trc.stats()
            return {"first_launch_ms": round(reduce_scalar(wall, "MAX"), 3), "first_launch_kernel_ms": round(s0.kernel_ms, 3),
                    "value": round(reduce_scalar(float(s0.rays), "SUM") / reduce_scalar(wall, "MAX") / 1e3, 2)}
        if args.no_cold:
            trc.debug_set("no_cold_probe", 1)
            cold = {"first_launch_ms": None, "skipped": "--no-cold"}
        else:
            cold_single = first_launch(True)
            cold = first_launch(False)
            cold.update({"unit": "Mrays/s", "settle_launches": 0,
                         "single_cold_pass_ms": cold_single["first_launch_ms"], "single_cold_pass_kernel_ms": cold_single["first_launch_kernel_ms"],
                         "what": "first launch of the block list (fresh costs): 8-sample head + rest planned from it, wall ms incl. seed + "
                                 "launch-list kernels; single_cold_pass = the same launch as one unordered pass (knob no_cold_probe)"})
            trc.debug_set("no_cold_probe", 0)

        # exact algorithmic work of ONE step on this rank (instrumented kernel, untimed)
        trc.reset_stats()
        step(collect_stats=True)
        trc.synchronize()
        st1 = trc.stats()
        own_pixels = st1.paths // spp_rank
        bytes_per_launch = algorithmic_bytes(st1, own_pixels)
        rays_per_launch = st1.rays

        # the launch order and the split plan are built from the durations of previous launches of the same block list and
        # settle within ~8 launches: a progressive renderer is in that state from its ninth frame on, whatever
        # W the caller picked.  Untimed, reported as config.settle_launches.
        for _ in range(SETTLE_LAUNCHES):
            step()
        for _ in range(warmup):
            step()
        barrier()
        trc.reset_stats()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        st = trc.stats()
        assert st.rays == rays_per_launch * steps, "steps are not identical work"
        kernel_ms = st.kernel_ms / max(1, st.launches)          # HIP events on the render stream
        schedule_ms = st.schedule_ms / max(1, st.launches)      # ... and around the launch-list kernels (order, sort, plan), averaged
        shape = trc.launch_shape()                               # the two bounds no schedule of this rank's share can beat

        # SELF-VALIDATION (outside the timed region): the frame the last timed step composed, against the oracle's CRC32 of the
        # same frame (tests/golden/bench_goldens.json).  Tiles compose the one-GPU frame bit for bit; sample shards and stacked
        # views have their own committed CRCs.  False = the numbers above belong to a wrong frame: the run exits non-zero.
        crc_check = None
        if rank == 0:
            key = f"config2_samples_S{S}" if sharded else (f"config2_weak_N{world}" if stacked else "config2")
            frame = trc.download_composed() if composing else trc.download_accum()
            crc_check = check_against(goldens, key, crc=frame_crc(frame),
                                      rays=None if world > 1 else rays_per_launch)
            if os.environ.get("TRC_BENCH_EXPECT_GOLDEN"):           # test hook: hold the frame to another entry (a wrong one must fail)
                crc_check = check_against(goldens, os.environ["TRC_BENCH_EXPECT_GOLDEN"], crc=frame_crc(frame))

        # the same K steps with a DIFFERENT seed each (a progressive renderer never replays a frame: the adaptive
        # launch order then works from the previous frame's costs, not from this frame's own)
        barrier()
        trc.reset_stats()
        barrier()
        t1 = time.perf_counter()
        for i in range(steps):
            step(seed=SEED + 0x1000 + i)
        barrier()
        dt_vary = time.perf_counter() - t1
        st_vary = trc.stats()
        kernel_ms_vary = st_vary.kernel_ms / max(1, st_vary.launches)

        # the compose alone (not overlapped): one synchronous reduce per iteration
        compose_ms = None
        if composing:
            barrier()
            t2 = time.perf_counter()
            for _ in range(steps):
                if sharded:
                    trc.group_compose_samples(0, S)
                else:
                    trc.group_reduce_accum(0)
            barrier()
            compose_ms = (time.perf_counter() - t2) / steps * 1e3

        dt_max = reduce_scalar(dt, "MAX")
        dt_vary_max = reduce_scalar(dt_vary, "MAX")
        rays_total = reduce_scalar(float(st.rays), "SUM")
        rays_vary_total = reduce_scalar(float(st_vary.rays), "SUM")
        per_rank = gather_list({"rank": rank, "kernel_ms": round(kernel_ms, 3), "kernel_ms_vary_seed": round(kernel_ms_vary, 3),
                                "compose_ms": None if compose_ms is None else round(compose_ms, 3), "schedule_ms": round(schedule_ms, 4),
                                "rays_per_step": int(st.rays // steps), "device": device, "pci_bus_id": bus_id,
                                "sample_group": group_of, "tile_rank": tile_rank, "spp": spp_rank,
                                # a pixel's samples are one chain: the launch cannot end before its slowest wavefront-sized item
                                # (block or part), nor before the items' summed durations over the GPU's wavefront slots
                                "longest_chain_ms": round(shape["longest_entry_ms"], 3),
                                "work_over_slots_ms": round(shape["work_over_slots_ms"], 3),
                                "launch_entries": shape["entries"], "wave_slots": shape["wave_slots"]})
        name = "Mrays/s at 1920x1080x64spp"
        if stacked:
            name = f"Mrays/s over {world} stacked 1920x1080x64spp views, one view's worth of tiles per GPU (weak scaling)"
        if sharded:
            name = (f"Mrays/s at 1920x1080x64spp, the 64 samples per pixel split over {S} seeds ({spp_rank} per seed"
                    + (f", each seed's frame tiled over {T} GPUs" if T > 1 else "") + "; sample sharding)")
        return {
            "mode": mode if world > 1 else "single", "metric": name, "value": round(rays_total / dt_max / 1e6, 2),
            "ms_per_step": round(dt_max / steps * 1e3, 3), "rays_per_step": int(rays_total / steps),
            "frame": [W, FH], "paths_per_step": W * FH * SPP,
            "mpaths_per_s": round(W * FH * SPP * steps / dt_max / 1e6, 2),
            "vary_seed": {"value": round(rays_vary_total / dt_vary_max / 1e6, 2),
                          "ms_per_step": round(dt_vary_max / steps * 1e3, 3), "kernel_ms": round(kernel_ms_vary, 3),
                          "what": "the same K steps with a different RNG seed per step (frame k's block costs order frame k+1)"},
            "per_rank": per_rank, "kernel_ms": kernel_ms, "bytes_per_launch": bytes_per_launch, "cold": cold,
            "rays_per_launch": rays_per_launch, "sample_groups": S, "tile_ranks": T,
            "oracle_check": crc_check, "composed_crc_ok": None if crc_check is None else crc_check.get("crc_ok"),
        }

    primary = measure(args.scaling, args.steps, args.warmup)
    info = trc.device_info()                   # before any optional leg: the watchdog thread must never touch the context

    def build_line(primary):
        """rank 0's line without the optional legs (they are added as they finish)"""
        kernel_ms = primary["kernel_ms"]
        achieved = primary["bytes_per_launch"] / (kernel_ms * 1e-3) / 1e9
        roof = roofline_from_pmc(kernel_ms) if world == 1 else None
        FH = primary["frame"][1]
        compose = "none"
        sharded = primary["mode"] == "samples"
        if grouped:
            rccl_what = (f"sample shards: all-to-all of the accumulators' {world} pixel slices (ncclSend / ncclRecv), rank-ordered fold, gather "
                         f"to rank 0" if sharded else f"ncclReduce(sum) of the {W}x{FH} RGBA32F frame to rank 0")
            table_what = (f"sample shards ({world} pixel slices: alltoall, rank-ordered fold, gather to rank 0)" if sharded
                          else f"reduce(sum) of the {W}x{FH} frame")
            compose = (f"{rccl_what} on a second stream, overlapped with the next step" if use_rccl else
                       f"{table_what} through trc_group_set_collectives (host-staged, TCP sockets): {compose_fallback}" if compose_fallback else
                       f"PLUMBING RUN, NOT A MEASUREMENT: {world} ranks share {n_dev} GPU(s); {table_what} "
                       f"through trc_group_set_collectives (host-staged, TCP sockets) because RCCL refuses two ranks on one device")
        line = {
            "metric": primary["metric"], "value": primary["value"], "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": primary["ms_per_step"], "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RT_Metal Cornell box + 12 spheres (BASELINE config 2), 1920x1080x64spp, "
                                   "tracePath depth 8, 21 leaves / 41 BVH nodes" +
                                   ("" if world == 1 else
                                    (f"; ONE frame, its 16x16 tiles sharded over {world} GPUs" if primary["mode"] == "strong" else
                                     f"; ONE frame, its 64 samples per pixel split over {primary['sample_groups']} seeds x {primary['tile_ranks']} tile ranks "
                                     f"(rank-ordered mean of the shards; another sample set than one GPU's)" if sharded else
                                     f"; {world} such views stacked into one 1920x{FH} frame, one view's worth of tiles per GPU")),
                       "integrator": "tracePath", "rays_per_step": primary["rays_per_step"],
                       "paths_per_step": primary["paths_per_step"], "mpaths_per_s": primary["mpaths_per_s"],
                       "tiles": f"16x16 px, owner (tx+ty)%{primary['tile_ranks'] if world > 1 else 1}", "device": info["name"], "compose": compose,
                       "settle_launches": SETTLE_LAUNCHES},
            # the composed frame of the timed steps against the oracle's committed CRC32 (false: the run exits non-zero)
            "composed_crc_ok": primary["composed_crc_ok"], "oracle_check": primary["oracle_check"],
            "vary_seed": primary["vary_seed"],
            "cold": primary["cold"], "first_launch_ms": primary["cold"]["first_launch_ms"],
            # parity-imposed bounds of the timed launches on rank 0 (every rank's under per_rank at N > 1)
            "launch_bounds": {k: primary["per_rank"][0][k] for k in ("longest_chain_ms", "work_over_slots_ms", "launch_entries", "wave_slots")},
            "schedule_ms": primary["per_rank"][0]["schedule_ms"],      # launch-list kernels per step (a settled list re-plans every 4th launch)
            # the BINDING bound of the dominant kernel: VALU issue (PMC counters of this library, profiles/rNN/pmc_config2.json)
            "roofline": roof,
            # the north_star's figure: bytes the REFERENCE's access pattern would move for this work over the
            # kernel time.  Not a bound here -- the scene is LDS-resident, so it can exceed the HBM peak; physical traffic
            # is roofline.traffic / roofline.l2_miss_traffic_frac_of_hbm_peak
            "roofline_hbm_algorithmic": {
                "bound": "hbm (algorithmic bytes, not a physical bound)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "kernel": "k_render", "kernel_ms": round(kernel_ms, 3),
                "algorithmic_bytes_per_launch": int(primary["bytes_per_launch"]),
                "bytes_per_ray": round(primary["bytes_per_launch"] / max(1, primary["rays_per_launch"]), 1)},
        }
        if compose_fallback:
            line["compose_fallback"] = compose_fallback
        elif plumbing:
            line["plumbing"] = True
        if grouped:
            line["devices"] = devices               # PCI bus id per rank: what decided the transport
            line["transport"] = "rccl" if use_rccl else "table"
        if world > 1:
            line["per_rank"] = primary["per_rank"]
            # how to read an N-GPU curve of this line: the tile split renders the ONE frame BASELINE names, bit for bit, and a
            # pixel's samples are one sequential chain -- a share ends on its slowest item however few it owns (one GPU emulating a rank:
            # 39 / 26 / 22 % efficiency at N = 8 on configs 2 / 3 / 4, profiles/r06/share_bounds.txt); the split that scales (84 / 92 / 93 %
            # there) is the sample-sharded one, measured in the same run under other_scaling.samples
            line["scaling_note"] = ("strong = tile split of the named frame (bit-identical to one GPU; chain-bound: per_rank[].longest_chain_ms vs "
                                    "work_over_slots_ms); the split that scales is other_scaling.samples (sample shards, bit-defined)")
        if world == 1 and not args.no_fast_math:
            line["fast_math_variant"] = fast_math_leg(scene, cam, args.steps)
        if not args.no_cpu_baseline and world == 1:      # CPU baselines: rank 0 at N = 1 only
            line["cpu_baseline"] = cpu_baseline(scene, cam)
            line["cpu_rt_weekend"] = cpu_rt_weekend()
        return line

    # The headline is complete HERE; everything below is an optional leg that must never take it down.  Each leg runs under a
    # watchdog and inside try / except; its result (or why it was skipped) is added to the line, which goes out exactly once.
    import threading
    line = build_line(primary) if rank == 0 else None
    out_lock = threading.Lock()
    state = {"printed": False, "group_ok": True}

    def print_line():
        with out_lock:
            if state["printed"] or rank != 0:
                return
            state["printed"] = True
            print(json.dumps(line), flush=True)

    def run_leg(container, name, fn, timeout_s):
        """fn() under a watchdog.  Rank 0 alone decides a bail: it prints the line without the leg and ends the process; the peers
        wait a grace period (rank 0's exit closes their sockets first) and leave quietly.  An exception inside the leg is a skip;
        the ranks then agree on it, and a group that can no longer agree runs no further leg."""
        if not state["group_ok"]:
            if rank == 0:
                line.setdefault(container, {})[name] = {"skipped": "an earlier leg left the ranks out of step"}
            return
        finished = threading.Event()

        def bail():
            if finished.is_set():
                return
            if rank == 0:
                with out_lock:
                    if finished.is_set():
                        return
                    line.setdefault(container, {})[name] = {"skipped": f"did not finish within {timeout_s:.0f} s; the line goes out without it"}
                print_line()
                sys.stdout.flush()
                os._exit(0)
            time.sleep(30.0)
            os._exit(0)
        dog = threading.Timer(timeout_s, bail)
        dog.daemon = True
        dog.start()
        try:
            res = fn()
        except BaseException as e:                       # SystemExit of a sanity check included: the headline still goes out
            res = {"skipped": f"{type(e).__name__}: {e}"}
        finally:
            with out_lock:
                finished.set()
            dog.cancel()
        if group is not None:
            try:
                everybody = group.allreduce_scalar(0 if (isinstance(res, dict) and "skipped" in res) else 1, "MIN") >= 1
            except Exception as e:                       # a peer is gone or out of step: no further collective is safe
                everybody, state["group_ok"] = False, False
                res = {"skipped": f"the ranks lost step ({type(e).__name__}: {e})"}
            if not everybody and not (isinstance(res, dict) and "skipped" in res):
                res = {"skipped": "skipped on another rank"}
        if rank == 0:
            line.setdefault(container, {})[name] = res

    if world > 1 and not args.no_other_scaling:
        keys = ("mode", "metric", "value", "ms_per_step", "rays_per_step", "frame", "vary_seed", "cold", "per_rank",
                "sample_groups", "tile_ranks", "skipped", "composed_crc_ok", "oracle_check")
        for m in ("strong", "weak", "samples"):
            if m != args.scaling:
                run_leg("other_scaling", m, lambda m=m: {k: v for k, v in measure(m, args.steps, args.warmup).items() if k in keys}, args.other_timeout)
    if world == 1 and not args.no_other_configs:
        # every other BASELINE configuration + the HBM operating point, witnessed by the same run (each on a context of its own)
        for key in OTHER_CONFIGS:
            run_leg("other_configs", key, lambda key=key: other_config_leg(key, device), args.other_timeout)
        if rank == 0 and "hbm_point" in line.get("other_configs", {}) and "hbm" in line["other_configs"]["hbm_point"]:
            line["hbm_point"] = line["other_configs"]["hbm_point"]["hbm"]
    print_line()
    sys.stdout.flush()
    crc_bad = rank == 0 and (line.get("composed_crc_ok") is False or
                             any(isinstance(o, dict) and o.get("composed_crc_ok") is False for o in line.get("other_scaling", {}).values()))

    def teardown():
        if grouped:
            try:
                if composing:
                    trc.group_finalize()
                if state["group_ok"]:
                    group.barrier()
            except Exception as e:
                sys.stderr.write(f"bench.py: rank {rank}: teardown: {e}\n")
            group.close()
        trc.close()
    _with_c_stdout_on_stderr(teardown)
    os.dup2(2, 1)       # anything native code still prints at exit goes to stderr, after the JSON line
    if crc_bad:
        sys.stderr.write("bench.py: the composed frame does NOT match the oracle's committed CRC32 (composed_crc_ok false): the numbers of "
                         "this line belong to a wrong frame\n")
        sys.exit(4)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""gpurun_out/<tag>/ (tools/pmc_collect.sh) -> one JSON summary: per-launch averages of every counter for the kernel
whose name contains <kernel-substring> (the LONGEST-running match if several), the rocprofv3 --stats row of that
kernel, the derived figures, the workload's own JSON line and the source hash of the library that ran.

    python3 tools/pmc_summary.py <tag> <kernel-substring> <out.json>

Derived (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are KB; GRBM_GUI_ACTIVE is summed over the 8 XCDs; a wave64 VALU
instruction issues over 2 cycles of a SIMD-32.
L2-miss traffic (`l2_miss_*`): bytes the L2s request from / write to the fabric -- Infinity-Cache hits INCLUDED (the counters sit
on the L2's memory side; TCC_EA0_RDREQ_DRAM counts the same requests whether the table fits the Infinity Cache or not,
profiles/r05/fetch_size_calibration.txt), so it bounds HBM traffic from above and equals it only for working sets far beyond
256 MiB.  Reads = 32 / 64 / 128 bytes x the sized request counters when they were collected (on gfx950 every read request of
this path's gathers is 128 bytes, and FETCH_SIZE tallies them at 64: TCC_BUBBLE reads 0), else 2 x FETCH_SIZE.
"""
import collections, csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, sub, out_path = sys.argv[1], sys.argv[2], sys.argv[3]
# only the LAST n launches of the kernel count (the program's timed, settled steps: a first launch runs as an 8-sample head + the
# rest, an instrumented or cold launch is another workload); 0 = all of them
last_n = int(sys.argv[4]) if len(sys.argv) > 4 else 0


def lib_source_hash():
    """the library's identity as bench.py computes it (code only: comments and white space do not count)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench.lib_source_hash()


stats_row, kernel = None, None
for f in glob.glob(f"gpurun_out/{tag}/trace/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Name"]]
    if rows:
        stats_row = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        kernel = stats_row["Name"]
if kernel is None:
    raise SystemExit(f"no kernel matching {sub!r} in gpurun_out/{tag}/trace")
vals = collections.defaultdict(list)
meta = {}
for f in glob.glob(f"gpurun_out/{tag}/pmc_*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == kernel]
    if last_n:
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-last_n:]
        rows = [r for r in rows if int(r["Dispatch_Id"]) in ids]
    for r in rows:
        if True:
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {"VGPR_Count": int(r["VGPR_Count"]), "LDS_Block_Size": int(r["LDS_Block_Size"]),
                    "Scratch_Size": int(r.get("Scratch_Size", 0) or 0), "Workgroup_Size": int(r["Workgroup_Size"]),
                    "Grid_Size": int(r["Grid_Size"])}
res = {"kernel": kernel.replace("void ", "").replace("(KRender)", ""), "tag": tag, "lib_source_hash": lib_source_hash()}
res.update(meta)
for c, v in sorted(vals.items()):
    res[c] = sum(v) / len(v)
res["launches_sampled"] = {c: len(v) for c, v in vals.items()}
res["kernel_ms"] = float(stats_row["AverageNs"]) / 1e6
if last_n:                                  # the same launches' durations from the kernel trace
    for f in glob.glob(f"gpurun_out/{tag}/trace/*/*_kernel_trace.csv"):
        tr = sorted([r for r in csv.DictReader(open(f)) if r["Kernel_Name"] == kernel], key=lambda r: int(r["Start_Timestamp"]))[-last_n:]
        if tr:
            res["kernel_ms"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr) / len(tr) / 1e6
            res["kernel_ms_launches"] = len(tr)
res["kernel_stats_row"] = stats_row
rd = None
if all(k in res for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
    rd = 32 * res["TCC_EA0_RDREQ_32B_sum"] + 64 * res["TCC_EA0_RDREQ_64B_sum"] + 128 * res["TCC_EA0_RDREQ_128B_sum"]
    res["l2_miss_read_source"] = "TCC_EA0_RDREQ_{32B,64B,128B}_sum"
elif "FETCH_SIZE" in res:
    rd = 2 * res["FETCH_SIZE"] * 1024
    res["l2_miss_read_source"] = "2 x FETCH_SIZE"
if rd is not None and "WRITE_SIZE" in res:
    res["l2_miss_read_bytes_per_launch"] = int(rd)
    res["l2_miss_write_bytes_per_launch"] = int(res["WRITE_SIZE"] * 1024)
    res["l2_miss_bytes_per_launch"] = int(rd + res["WRITE_SIZE"] * 1024)
    res["l2_miss_gbs"] = res["l2_miss_bytes_per_launch"] / res["kernel_ms"] / 1e6
    res["l2_miss_frac_of_hbm_peak"] = res["l2_miss_gbs"] / 8000.0
    res["hbm_bytes_per_launch"] = res["l2_miss_bytes_per_launch"]      # the name rounds 1-4 used (an upper bound of HBM traffic: Infinity-Cache hits included)
if "GRBM_GUI_ACTIVE" in res and "SQ_INSTS_VALU" in res:
    cyc = res["GRBM_GUI_ACTIVE"] / 8.0
    res["shader_cycles_per_launch"] = cyc
    res["valu_issue_frac"] = res["SQ_INSTS_VALU"] * 2.0 / (1024 * cyc)
    if "SQ_THREAD_CYCLES_VALU" in res:
        res["valu_lane_utilisation"] = res["SQ_THREAD_CYCLES_VALU"] / (64.0 * res["SQ_INSTS_VALU"])
        res["valu_useful_lane_frac"] = res["valu_issue_frac"] * res["valu_lane_utilisation"]
if "TCC_HIT_sum" in res and "TCC_MISS_sum" in res and res["TCC_HIT_sum"] + res["TCC_MISS_sum"] > 0:
    res["l2_hit_rate"] = res["TCC_HIT_sum"] / (res["TCC_HIT_sum"] + res["TCC_MISS_sum"])
if "SQ_WAVE_CYCLES" in res:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if k in res:
            res[k.lower() + "_frac_of_wave_cycles"] = res[k] / res["SQ_WAVE_CYCLES"]
try:
    last = [l for l in open(f"gpurun_out/{tag}/trace.log").read().splitlines() if l.startswith("{")][-1]
    res["workload_line"] = json.loads(last)
    # the program's own `roofline` objects quote whichever PMC summary was committed when it ran -- an older library's, by
    # construction, while THIS summary is being collected: not part of what was measured here
    for k in ("roofline", "roofline_per_pass"):
        res["workload_line"].pop(k, None)
except Exception:
    res["workload_line"] = None
json.dump(res, open(out_path, "w"), indent=1, sort_keys=True)
print(json.dumps({k: res[k] for k in res if k not in ("kernel_stats_row", "launches_sampled", "workload_line")}, indent=1, sort_keys=True))

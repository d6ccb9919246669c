#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (made by tools_pmc.sh) into per-launch averages for k_render."""
import collections, csv, glob, json, sys
tag = sys.argv[1]
out = {}
for f in glob.glob(f"gpurun_out/{tag}/pmc_*/*/*_counter_collection.csv"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_render" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
            out.setdefault(r["Kernel_Name"], {})["VGPR_Count"] = int(r["VGPR_Count"])
            out[r["Kernel_Name"]]["LDS_Block_Size"] = int(r["LDS_Block_Size"])
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
for k, d in out.items():
    if "SQ_THREAD_CYCLES_VALU" in d and "SQ_INSTS_VALU" in d:
        d["derived_valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (d["SQ_INSTS_VALU"] * 64)
for f in glob.glob(f"gpurun_out/{tag}/trace/*/*_kernel_stats.csv"):
    print(open(f).read())
print(json.dumps(out, indent=1, sort_keys=True))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)

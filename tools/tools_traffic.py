#!/usr/bin/env python3
"""gpurun_out/<tag>/pmc_fetch + pmc_write (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)
-> profiles/<round>/traffic.json.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the
bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so it is doubled."""
import collections, csv, glob, json, sys
tag, out = sys.argv[1], sys.argv[2]
vals = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_render<true, false, 0>" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
fetch_kb = sum(vals["FETCH_SIZE"]) / len(vals["FETCH_SIZE"])
write_kb = sum(vals["WRITE_SIZE"]) / len(vals["WRITE_SIZE"])
res = {"kernel": "k_render<true, false, 0>", "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
       "correction": "FETCH_SIZE x2 (gfx950 128-B requests tallied at 64 B)",
       "hbm_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024)}
json.dump(res, open(out, "w"), indent=1)
print(res)

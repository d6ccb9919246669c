#!/bin/bash
# What the fabric-side counters count for this path's access pattern (MI355X_MICROARCH.md: "other access widths are uncalibrated:
# calibrate on a known byte count in your own access pattern").  tools/probe/gather_probe reads a known number of bytes the way the
# render kernels read nodes (64-byte records, four 16-byte loads per lane) and triangles (48 bytes, three loads), from tables beyond
# and inside the 256 MiB Infinity Cache, beside a plain streaming read; each mode once under the kernel trace (ms) and once per
# counter group.       usage: bash tools/fetch_size_calibration.sh <out-dir>      (prints the table)
export TMPDIR=/tmp
O=${1:-gpurun_out/calib}; W=$O/calib_raw; mkdir -p $W
[ tools/probe/gather_probe -nt tools/probe/gather_probe.hip ] || /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o tools/probe/gather_probe tools/probe/gather_probe.hip
for mode in "stream 2048" "chain1 2048 64" "chain2 2048 64" "64 2048 64" "coop64 2048 64" "48 2048 64" "chain1 128 64" "64 128 64" "coop64 128 64" "chain1 8 64" "64 8 64"; do
  tag=$(echo $mode | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/${tag}_trace -- tools/probe/gather_probe $mode > $W/${tag}_trace.log 2>&1
  n=0
  for pass in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum"; do
    n=$((n+1)); rocprofv3 --pmc $pass --output-format csv -d $W/${tag}_pmc$n -- tools/probe/gather_probe $mode > $W/${tag}_pmc$n.log 2>&1
  done
done
python3 - "$W" <<'PY'
import csv, glob, collections, json, sys
W = sys.argv[1]
print("# \"chain\": the next record's address depends on the record just read, the four loads issued back to back as in the render kernels' ISA;")
print("# \"each after the last\": the probe's loop waits for a load before it issues the next (what the compiler made of a loop with a run-time count)")
print("# mode: what every lane reads; table: size of the table the records are drawn from (uniformly at random); 3 launches each, means")
print("# requested = bytes the lanes asked for; EA reads = TCC_EA0_RDREQ_sum (all of them 128-byte requests on gfx950: _32B and _64B read 0);")
print("# FETCH_SIZE as rocprofv3 reports it (KB -> bytes): RDREQ x 64, i.e. HALF of the bytes moved (its TCC_BUBBLE term reads 0)")
print(f"{'mode':42s} {'table':>7s} {'ms':>7s} {'requested GB':>12s} {'useful GB/s':>11s} {'TCC_REQ M':>10s} {'TCC_MISS M':>10s} {'EA reads M':>10s} {'of them DRAM':>12s} {'x 128 B = GB':>12s} {'per record':>10s} {'FETCH_SIZE GB':>13s}")
names = {"stream": "16 B per lane, contiguous", "64": "64-B record, 4 loads, each after the last", "coop64": "64-B record, 4 lanes x 1 load", "48": "48-B record, 3 loads, each after the last",
         "chain1": "64-B record, 4 loads together, chain", "chain2": "64-B record, 1 + 3 loads, chain"}
for mode in ["stream_2048", "chain1_2048_64", "chain2_2048_64", "64_2048_64", "coop64_2048_64", "48_2048_64", "chain1_128_64", "64_128_64", "coop64_128_64", "chain1_8_64", "64_8_64"]:
    v = {}
    for f in glob.glob(f"{W}/{mode}_trace/*/*_kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "k_" in r["Name"]: v["ms"] = float(r["AverageNs"]) / 1e6
    for f in glob.glob(f"{W}/{mode}_pmc*/*/*_counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, x in acc.items(): v[k] = sum(x) / len(x)
    line = json.loads([l for l in open(f"{W}/{mode}_trace.log") if l.startswith("{")][-1])
    req = line["bytes_requested_per_launch"]
    kind = mode.split("_")[0]; mb = mode.split("_")[1]
    recs = req / line["record_bytes"]
    print(f"{names[kind]:42s} {mb + ' MB':>7s} {v['ms']:7.3f} {req / 1e9:12.3f} {req / 1e9 / (v['ms'] / 1e3):11.0f} {v.get('TCC_REQ_sum', 0) / 1e6:10.1f} {v.get('TCC_MISS_sum', 0) / 1e6:10.1f} "
          f"{v.get('TCC_EA0_RDREQ_sum', 0) / 1e6:10.1f} {v.get('TCC_EA0_RDREQ_DRAM_sum', 0) / 1e6:12.1f} {v.get('TCC_EA0_RDREQ_sum', 0) * 128 / 1e9:12.2f} "
          f"{(v.get('TCC_EA0_RDREQ_sum', 0) / recs if kind != 'stream' else float('nan')):10.2f} {v.get('FETCH_SIZE', 0) * 1024 / 1e9:13.2f}")
PY
rm -rf $W

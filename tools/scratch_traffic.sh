# HBM traffic of the config-4 kernel WITHOUT register spills (a 4-wave build: `make variant NAME=p4w DEFS="-DTRC_PATH_WAVES_GLOBAL=4 -DTRC_PWG_WAVES_PATH=16 -DTRC_PWG_PER_CU_PATH=1"`),
# to split the shipped 6-wave build's traffic (profiles/r03/pmc_config4.json) into scratch and node / triangle fetches.  gpurun -- bash tools/scratch_traffic.sh
export TMPDIR=/tmp
export TRC_AMD_LIB=$PWD/build/libp4w.so
R=$PWD; T=r03_c4_nospill; mkdir -p gpurun_out/$T
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/$T/pmc_$n -- python3 tools/config_bench.py --config 4 --spp 32 --steps 3 > gpurun_out/$T/$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for n in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VMEM_RD_SQ_INSTS_VMEM_WR"):
    for f in glob.glob(f"gpurun_out/r03_c4_nospill/pmc_{n}/*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void k_render_pwg<0, false>"):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        # per dispatch the counter rows come per XCD/instance: sum per dispatch id
        by = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("void k_render_pwg<0, false>"):
                by[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for k, d in by.items():
            v = list(d.values())
            print(n, k, "launches", len(v), "mean per launch", sum(v) / len(v), "VGPR", r.get("VGPR_Count"), "scratch", r.get("Scratch_Size"))
PY

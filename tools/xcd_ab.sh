#!/bin/bash
# A/B of the XCD windows of the persistent-workgroup launches (k_xcd_windows): one global queue / windows with macro-tiles of
# 2^k x 2^k blocks / the control (one macro-tile = the whole frame: eight queues, no affinity).   gpurun -- 'bash tools/xcd_ab.sh'
cd ${GRAFT_REPO_ROOT:-.}
pj() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['kernel_ms'], 'ms', d['mrays_per_s'], 'Mrays/s')"; }
for c in ${CONFIGS:-4 3 volume}; do
  spp=32; [ $c = volume ] && spp=16
  TRC_NO_XCD_WINDOWS=1 python3 tools/config_bench.py --config $c --spp $spp --steps 4 --warmup 8 2>/dev/null | pj "config $c global queue            "
  for k in ${SHIFTS:-1 2 3 4 5 9}; do
    TRC_XCD_MACRO_SHIFT=$k python3 tools/config_bench.py --config $c --spp $spp --steps 4 --warmup 8 2>/dev/null | pj "config $c xcd windows, macro 2^$((k-1))"
  done
  TRC_NO_XCD_WINDOWS=1 python3 tools/config_bench.py --config $c --spp $spp --steps 4 --warmup 8 2>/dev/null | pj "config $c global queue (again)    "
done

#!/bin/bash
# usage: tools/pmc_collect.sh <tag> <program> [args...]   e.g.  tools/pmc_collect.sh r02_c3 python3 tools/config_bench.py --config 3 --spp 32
# rocprofv3 kernel trace + PMC passes (one counter group per pass; --pmc is never combined with a trace domain other
# than the kernel trace) of the program into gpurun_out/<tag>/.  The program itself follows `--` (no shell, no env).
export TMPDIR=/tmp
R=$PWD; T=$1; shift; mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/trace -- "$@" > gpurun_out/$T/trace.log 2>&1
pass() { n=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$T/pmc_$n -- "${CMD[@]}" > gpurun_out/$T/pmc_$n.log 2>&1; }
CMD=("$@")
pass sq  SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass sq2 GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT
pass sq3 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
# the fabric-side requests by size (FETCH_SIZE's expression counts 128-byte requests through TCC_BUBBLE, which reads 0 on gfx950:
# tools/probe/gather_probe.hip, profiles/r05/fetch_size_calibration.txt) and the writes the same way
pass ea TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
pass eaw TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
tail -n 1 gpurun_out/$T/trace.log | cut -c1-400

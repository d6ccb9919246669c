#!/bin/bash
# usage: tools_pmc.sh <tag>  -- kernel trace + PMC passes of bench.py into gpurun_out/<tag>/
export TMPDIR=/tmp
R=$PWD; T=$1; mkdir -p gpurun_out/$T
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/trace -- $B > gpurun_out/$T/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$T/pmc_sq -- $B > gpurun_out/$T/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/$T/pmc_sq2 -- $B > gpurun_out/$T/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INSTS_VALU_TRANS SQ_LEVEL_WAVES SQ_CYCLES --output-format csv -d $R/gpurun_out/$T/pmc_sq3 -- $B > gpurun_out/$T/pmc_sq3.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_fetch -- $B > gpurun_out/$T/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_write -- $B > gpurun_out/$T/pmc_write.log 2>&1
tail -n 1 gpurun_out/$T/trace.log | cut -c1-200

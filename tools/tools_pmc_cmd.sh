#!/bin/bash
# usage: tools_pmc_cmd.sh <tag> <python script> -- kernel trace + PMC passes into gpurun_out/<tag>/
export TMPDIR=/tmp
R=$PWD; T=$1; S=$2; mkdir -p gpurun_out/$T
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$T/trace -- python3 $S > gpurun_out/$T/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$T/pmc_sq -- python3 $S > gpurun_out/$T/pmc_sq.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/$T/pmc_sq2 -- python3 $S > gpurun_out/$T/pmc_sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_fetch -- python3 $S > gpurun_out/$T/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$T/pmc_write -- python3 $S > gpurun_out/$T/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCC_REQ_sum --output-format csv -d $R/gpurun_out/$T/pmc_tcc -- python3 $S > gpurun_out/$T/pmc_tcc.log 2>&1
tail -n 2 gpurun_out/$T/trace.log | cut -c1-250

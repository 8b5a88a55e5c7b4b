#!/bin/bash
# PMC passes over the pair-dictionary SpMV (27-point 192^3), one rocprofv3 run per counter set.  Run on the GPU box from the repo root.
set -o pipefail
R=$PWD
export TMPDIR=/tmp
cd /tmp
CMD="python3 $R/scripts/bench_kernels.py --stencil ${STENCIL:-27} --grid ${GRID:-192} --vec 0 --reps 3"
rm -rf $R/gpurun_out/pmc_pair_*
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY --output-format csv -d $R/gpurun_out/pmc_pair_A -- $CMD > $R/gpurun_out/pmc_pair_A.log 2>&1 || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM --output-format csv -d $R/gpurun_out/pmc_pair_B -- $CMD > $R/gpurun_out/pmc_pair_B.log 2>&1 || exit 2
timeout -s KILL 150 rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum --output-format csv -d $R/gpurun_out/pmc_pair_C -- $CMD > $R/gpurun_out/pmc_pair_C.log 2>&1 || exit 3
timeout -s KILL 150 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc_pair_D -- $CMD > $R/gpurun_out/pmc_pair_D.log 2>&1 || exit 4
echo pmc collected

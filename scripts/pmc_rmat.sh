#!/bin/bash
# Counter passes for the two R-MAT SpMV kernels (spmv_rowblock_kernel, spmv_wave_lds_kernel) on a gpurun box.
# Usage: bash scripts/pmc_rmat.sh <scale>      results: gpurun_out/pmc_rmat<scale>_<pass>/, summary with scripts/pmc_summary.py rmat<scale>
set -o pipefail
R=$PWD
S=${1:-20}
TAG=rmat$S
export TMPDIR=/tmp
export PHIP_LDS_HOT=1   # build the LDS layout too (off by default)
cd /tmp
pass() {
  local name=$1; shift
  local ctr=$1; shift
  rm -rf $R/gpurun_out/pmc_${TAG}_${name}
  timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name} -- python3 $R/scripts/exp_rmat_lds.py $S > $R/gpurun_out/pmc_${TAG}_${name}.log 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/pmc_${TAG}_${name}.log; return 1; }
  echo "pass $name ok"
}
pass inst "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" || exit 1
pass act "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH" || exit 2
pass tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" || exit 3
pass ta "TA_BUSY_avr TA_TA_BUSY_sum" || exit 4
pass grbm "GRBM_GUI_ACTIVE GRBM_COUNT" || exit 5
pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" || exit 6
pass lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" || exit 7
echo pmc done

#!/bin/bash
# Counter passes for the SpMV kernel of a bench workload (gpurun box).  Usage: bash scripts/pmc_spmv.sh <tag> [bench args]
# Each pass is a run of its own (rocprofv3 does not split counters); results: gpurun_out/pmc_<tag>_<pass>/
set -o pipefail
R=$PWD
TAG=$1; shift
export TMPDIR=/tmp
cd /tmp
pass() {
  local name=$1; shift
  local ctr=$1; shift
  rm -rf $R/gpurun_out/pmc_${TAG}_${name}
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/pmc_${TAG}_${name} -- python3 $R/bench.py --no-cpu-baseline --legs none --steps 5 --warmup 1 "$@" > $R/gpurun_out/pmc_${TAG}_${name}.log 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/pmc_${TAG}_${name}.log; return 1; }
}
pass inst "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "$@" || exit 1
pass act "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH" "$@" || exit 2
pass tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "$@" || exit 3
pass ta "TA_BUSY_avr TA_TA_BUSY_sum" "$@" || exit 4
pass grbm "GRBM_GUI_ACTIVE GRBM_COUNT" "$@" || exit 5
pass tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "$@" || exit 6
pass tcp2 "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_DATA_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "$@" || exit 7
pass lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "$@" || exit 8
echo pmc done

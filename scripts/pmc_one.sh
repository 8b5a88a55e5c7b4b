#!/bin/bash
# rocprofv3 counter passes over scripts/exp_one_spmv.py (one pass per counter group; gpurun box).
# usage: bash scripts/pmc_one.sh <tag> <exp_one_spmv arguments...>  -> gpurun_out/r03/pmc_<tag>_<group>/
set -o pipefail
R=$PWD
export TMPDIR=/tmp
tag=$1; shift
mkdir -p $R/gpurun_out/r03
cd /tmp
i=0
PMCSETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
if [ -n "$PMC_MORE" ]; then
  PMCSETS+=("TA_BUSY_avr TA_TA_BUSY_sum" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum")
fi
for grp in "${PMCSETS[@]}"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/r03/pmc_${tag}_$i
  timeout -s KILL 240 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/r03/pmc_${tag}_$i -- python3 $R/scripts/exp_one_spmv.py "$@" > $R/gpurun_out/r03/pmc_${tag}_$i.log 2>&1 || { echo "pass $i failed"; tail -5 $R/gpurun_out/r03/pmc_${tag}_$i.log; exit 1; }
done
python3 $R/scripts/pmc_one_summary.py $R/gpurun_out/r03 $tag

#!/bin/bash
# L2 / fabric counters of the R-MAT product at a given scale (gpurun box).  Usage: bash scripts/pmc_rmat_l2.sh <scale>
set -o pipefail
R=$PWD
S=${1:-22}
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/pmc_rmatl2_$S
timeout -s KILL 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $R/gpurun_out/pmc_rmatl2_$S -- python3 $R/scripts/exp_rmat_lds.py $S > $R/gpurun_out/pmc_rmatl2_$S.log 2>&1 || { echo failed; tail -3 $R/gpurun_out/pmc_rmatl2_$S.log; exit 1; }
rm -rf $R/gpurun_out/pmc_rmatl2b_$S
timeout -s KILL 200 rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $R/gpurun_out/pmc_rmatl2b_$S -- python3 $R/scripts/exp_rmat_lds.py $S > $R/gpurun_out/pmc_rmatl2b_$S.log 2>&1 || { echo failed; exit 2; }
echo done

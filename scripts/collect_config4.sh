#!/bin/bash
# rocprofv3 kernel trace and FETCH_SIZE / WRITE_SIZE passes of BASELINE config 4 (convection-diffusion 192^3, GMRES(30)+Jacobi) on the GPU box.
set -o pipefail
R=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_c4_kt $R/gpurun_out/prof_c4_fetch $R/gpurun_out/prof_c4_write
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4_kt -- python3 $R/scripts/bench_configs.py --configs 4 > $R/gpurun_out/prof_c4_kt.log 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_c4_fetch -- python3 $R/scripts/bench_configs.py --configs 4 --its 31 > $R/gpurun_out/prof_c4_fetch.log 2>&1 || exit 2
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_c4_write -- python3 $R/scripts/bench_configs.py --configs 4 --its 31 > $R/gpurun_out/prof_c4_write.log 2>&1 || exit 3
find $R/gpurun_out/prof_c4_kt -name '*_kernel_trace.csv' -size +20M -delete
echo config 4 profiled

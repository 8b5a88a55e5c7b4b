#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace + two separate PMC passes of the default bench, then an R-MAT PMC pass.
# Usage: bash scripts/collect_profiles.sh   (from the repo root; results under gpurun_out/prof_*)
set -o pipefail
R=$PWD
export TMPDIR=/tmp
cd /tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write $R/gpurun_out/prof_rmat_fetch $R/gpurun_out/prof_rmat_kt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_kt.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/gpurun_out/prof_fetch.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 > $R/gpurun_out/prof_write.log 2>&1 || exit 3
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_rmat_kt -- python3 $R/scripts/bench_kernels.py --rmat 21 --reps 20 --vec 0 > $R/gpurun_out/prof_rmat_kt.log 2>&1 || exit 4
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_rmat_fetch -- python3 $R/scripts/bench_kernels.py --rmat 21 --reps 5 --vec 0 > $R/gpurun_out/prof_rmat_fetch.log 2>&1 || exit 5
echo profiles collected

#!/bin/bash
# rocprofv3 kernel trace + stats of one python script run (gpurun box): bash scripts/kt_run.sh <tag> <script> [args...] -> gpurun_out/r03/kt_<tag>/
set -o pipefail
R=$PWD
export TMPDIR=/tmp
tag=$1; shift
mkdir -p $R/gpurun_out/r03
cd /tmp
rm -rf $R/gpurun_out/r03/kt_$tag
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03/kt_$tag -- python3 $R/"$@" > $R/gpurun_out/r03/kt_$tag.log 2>&1 || { tail -5 $R/gpurun_out/r03/kt_$tag.log; exit 1; }
find $R/gpurun_out/r03/kt_$tag -name '*_kernel_trace.csv' -size +20M -delete
f=$(find $R/gpurun_out/r03/kt_$tag -name '*kernel_stats.csv' | head -1)
cut -d, -f1-4 "$f" | head -12

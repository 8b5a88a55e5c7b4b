#!/bin/bash
# Runs on the GPU box (gpurun): for each bench workload a rocprofv3 kernel trace and two separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# Usage: bash scripts/collect_profiles.sh   (from the repo root; results under gpurun_out/prof_<workload>_{kt,fetch,write});
# then, in the container, python scripts/summarize_profiles.py r02 copies the summaries into profiles/.
set -o pipefail
R=$PWD
export TMPDIR=/tmp
cd /tmp
run() { # name, bench arguments...
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_${name}_kt $R/gpurun_out/prof_${name}_fetch $R/gpurun_out/prof_${name}_write
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${name}_kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_${name}_kt.log 2>&1 || return 1
  timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${name}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 "$@" > $R/gpurun_out/prof_${name}_fetch.log 2>&1 || return 2
  timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${name}_write -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 "$@" > $R/gpurun_out/prof_${name}_write.log 2>&1 || return 3
  # the per-dispatch traces are large: keep the stats and the counter tables only
  find $R/gpurun_out/prof_${name}_kt -name '*_kernel_trace.csv' -size +20M -delete
  echo "$name done"
}
run box27_256 --workload box27_256 --legs none || exit 1
run lap7_128 --workload lap7_128 --legs none || exit 2
run lap7_256_var --workload lap7_128 --legs lap7_256_var --steps 20 || exit 3
echo profiles collected

#!/bin/bash
# Runs on the GPU box (gpurun): for every workload the bench line prints, a rocprofv3 kernel trace (--stats) and two separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of the same bench command.  Usage: bash scripts/collect_profiles.sh [workload ...]   (from the repo root;
# results under gpurun_out/prof_<workload>_{kt,fetch,write}); then, in the container, python scripts/summarize_profiles.py r03 copies the
# summaries into profiles/.  The main workload runs as `bench.py --workload W --legs none`, a leg as `bench.py --leg-only L`, so that a
# profile holds the kernels of one matrix.
set -o pipefail
R=$PWD
export TMPDIR=/tmp
cd /tmp
run() { # name, bench arguments...
  local name=$1; shift
  rm -rf $R/gpurun_out/prof_${name}_kt $R/gpurun_out/prof_${name}_fetch $R/gpurun_out/prof_${name}_write
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${name}_kt -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/prof_${name}_kt.log 2>&1 || { echo "$name: kernel trace failed"; tail -3 $R/gpurun_out/prof_${name}_kt.log; return 1; }
  timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_${name}_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 "$@" > $R/gpurun_out/prof_${name}_fetch.log 2>&1 || { echo "$name: FETCH pass failed"; return 2; }
  timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_${name}_write -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 2 "$@" > $R/gpurun_out/prof_${name}_write.log 2>&1 || { echo "$name: WRITE pass failed"; return 3; }
  # the per-dispatch traces are large: keep the stats and the counter tables only
  find $R/gpurun_out/prof_${name}_kt -name '*_kernel_trace.csv' -size +20M -delete
  echo "$name done"
}
mkdir -p $R/gpurun_out
WL="${@:-box27_256 lap7_128 lap7_256_var box27_256_var rmat_22}"
for w in $WL; do
  case $w in
    box27_256) run box27_256 --workload box27_256 --legs none || exit 1 ;;
    *) run $w --leg-only $w --steps 100 || exit 1 ;;
  esac
done
echo profiles collected

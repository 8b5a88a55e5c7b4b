#!/bin/bash
# sweep of the diagonal-mask kernels: two rows a lane vs four rows a lane with 1..2 groups in flight (run on the GPU box)
set -o pipefail
for w in ${WORKLOADS:-box27_256 lap7_256 lap7_128}; do
  for cfg in ${CFGS:-PHIP_DMASK_ROWS=2 PHIP_DMASK_GROUPS=1 PHIP_DMASK_GROUPS=2}; do
    echo "== $w $cfg"
    env $cfg PHIP_DMASK_DOT=1 timeout -k 10 200 python -u scripts/exp_dot_cost.py $w | grep workload || exit 1
  done
done

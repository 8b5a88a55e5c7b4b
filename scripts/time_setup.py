#!/usr/bin/env python3
"""Where phip_mat_create spends its time (PHIP_SETUP_TIMING=1 prints the phases).  python scripts/time_setup.py [workload]"""
import os
import sys
import time

os.environ["PHIP_SETUP_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from petsc_amd import capi  # noqa: E402

kind, n, _ = bench.parse_workload(sys.argv[1] if len(sys.argv) > 1 else "box27_256")
t = time.time()
N, m, rs, ai, aj, aa = bench.local_rows(kind, n, 1, 0, "strong")
print("generate %.2f s" % (time.time() - t), flush=True)
ctx = capi.Context(0)
t = time.time()
A = ctx.mat_create(m, m, ai, aj, aa)
print("phip_mat_create %.2f s (%d rows, %d nonzeros)" % (time.time() - t, m, int(ai[-1])), flush=True)
ctx.check(ctx.lib.phip_mat_destroy(A))
ctx.close()

#!/usr/bin/env python3
"""A few launches of ONE SpMV configuration, for rocprofv3 --pmc passes:
    python3 scripts/exp_one_spmv.py <lap7_N|box27_N|rmat_S> <fmt 0|1> <xcd chunk> [reps] [var 0|1]
fmt 1 = CSR row blocks, 0 = the library's choice.  Prints one JSON line with the average launch time."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from petsc_amd import capi, generators as G, petsc as P  # noqa: E402

name, fmt, chunk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
var = int(sys.argv[5]) if len(sys.argv) > 5 else 1
comm = P.Comm(device=0, backend="self")
L = capi.load()
tm = B.Timer(L, comm.ctx)
if name.startswith("rmat_"):
    ai, aj, aa = G.rmat(int(name.split("_")[1]))
    m = len(ai) - 1
else:
    kind, n, _ = B.parse_workload(name)
    N, m, rs, ai, aj, aa = B.local_rows(kind, n, 1, 0, "strong", bool(var))
nz = int(ai[-1])
A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
hA = A.device_mat(0)
x, y = A.create_vecs()
x.set_array(np.linspace(-1.0, 1.0, m))
dx, dy = x.device_ptr(), y.device_ptr()
assert L.phip_mat_set_format(hA, fmt) == 0
assert L.phip_set_xcd_chunk(chunk) == 0
kernel, layout, real, alg = B.layout_of(L, hA, m, m, nz)
for _ in range(3):
    L.phip_mat_mult(hA, dx, dy)
L.phip_sync(comm.ctx)
us = tm.region(lambda: L.phip_mat_mult(hA, dx, dy), reps) * 1e3
print(json.dumps({"matrix": name, "var": var, "kernel": kernel, "chunk": chunk, "us": round(us, 1), "layout_bytes": real, "algorithmic_bytes": alg}), flush=True)

#!/usr/bin/env python3
"""Where does the fused dot of the SpMV cost its time?  plain vs fused-dot product back to back, and KSPCG with / without by-products.
   python scripts/exp_dot_cost.py [box27_256|lap7_128|...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Timer, local_rows, parse_workload  # noqa: E402
from petsc_amd import capi, petsc as P  # noqa: E402
from petsc_amd.capi import Ticket  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "box27_256"
kind, n, _ = parse_workload(name)
comm = P.Comm(device=0, backend="self")
L = capi.load()
tm = Timer(L, comm.ctx)
N, m, rs, ai, aj, aa = local_rows(kind, n, 1, 0, "strong")
A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
hA = A.device_mat(0)
x, y = A.create_vecs()
x.set_array(np.linspace(-1, 1, m))
dx, dy = x.device_ptr(), y.device_ptr()
out = {"workload": name}
out["plain_us"] = tm.kernel(lambda: L.phip_mat_mult(hA, dx, dy))[0]
tk = Ticket()
out["dot_us"] = tm.kernel(lambda: L.phip_mat_mult_dot(hA, dx, dy, dx, 0, C.byref(tk)))[0]
out["dot_publish_us"] = tm.kernel(lambda: L.phip_mat_mult_dot(hA, dx, dy, dx, 1, C.byref(tk)))[0]
# flush caches between launches with a big elementwise op
ctxw = capi.Context.__new__(capi.Context); ctxw.lib, ctxw.h = L, comm.ctx
big = [ctxw.to_device(np.zeros(m)) for _ in range(3)]
def flush():
    L.phip_vec_waxpy(comm.ctx, m, big[0], 0.5, big[1], big[2]); L.phip_stream(comm.ctx)
out["flush_us"] = tm.kernel(flush)[0]
out["flush_plus_plain_us"] = tm.kernel(lambda: (flush(), L.phip_mat_mult(hA, dx, dy)))[0]
out["flush_plus_dot_us"] = tm.kernel(lambda: (flush(), L.phip_mat_mult_dot(hA, dx, dy, dx, 0, C.byref(tk))))[0]
b = y.duplicate(); x.set(1.0); A.mult(x, b)
for bp in (True, False):
    comm.set_byproducts(bp)
    ksp = P.KSP(comm, "cg", "jacobi"); ksp.set_operators(A)
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=30); ksp.solve(b, x)
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=200)
    L.phip_device_sync(comm.ctx); t = time.perf_counter(); r = ksp.solve(b, x); L.phip_device_sync(comm.ctx)
    out["cg_its_per_s_byproducts_%s" % bp] = 200 / (time.perf_counter() - t)
    ksp.destroy()
print(json.dumps(out))

#!/usr/bin/env python3
"""Where do the 13 us between the plain product (87 us back to back) and the product inside KSPCG (100 us) go on 27-point 256^3?
Back to back: the plain product, the product with its fused dot (weights = x, as KSPCG asks), and both with x alternating between
two vectors that a streaming kernel has just rewritten (what KSPCG's VecAYPX does to P before every product).
    python3 scripts/exp_spmv_dot.py [box27_256]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from petsc_amd import capi, petsc as P  # noqa: E402
from petsc_amd.capi import Ticket  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "box27_256"
comm = P.Comm(device=0, backend="self")
L = capi.load()
tm = B.Timer(L, comm.ctx)
kind, n, _ = B.parse_workload(name)
N, m, rs, ai, aj, aa = B.local_rows(kind, n, 1, 0, "strong", False)
A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
hA = A.device_mat(0)
x, y = A.create_vecs()
z = x.duplicate()
x.set_array(np.linspace(-1.0, 1.0, m))
z.set_array(np.linspace(1.0, -1.0, m))
dx, dy, dz = x.device_ptr(), y.device_ptr(), z.device_ptr()
tk = Ticket()
ctx = comm.ctx
out = {"matrix": name}


def plain():
    L.phip_mat_mult(hA, dx, dy)


def dot():
    L.phip_mat_mult_dot(hA, dx, dy, dx, 0, C.byref(tk))


def rewrite_then(f):
    def g():
        L.phip_vec_aypx(ctx, m, dx, 1.0000001, dz)  # x = z + b x: x leaves the kernel as KSPCG's P does
        L.phip_stream(ctx)
        f()
    return g


def only_rewrite():
    L.phip_vec_aypx(ctx, m, dx, 1.0000001, dz)
    L.phip_stream(ctx)


for label, f in (("plain", plain), ("fused_dot", dot), ("rewrite_x", only_rewrite), ("rewrite_x+plain", rewrite_then(plain)), ("rewrite_x+fused_dot", rewrite_then(dot))):
    for _ in range(5):
        f()
    L.phip_sync(ctx)
    out[label + "_us"] = round(tm.region(f, 100) * 1e3, 2)
print(json.dumps(out), flush=True)

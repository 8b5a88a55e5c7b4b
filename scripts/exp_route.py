#!/usr/bin/env python3
"""Cold-column routing on R-MAT: product time with routing (several H), with the hot-column order, and with neither.
usage: python scripts/exp_route.py <scale> [H ...]  -> JSON lines"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from petsc_amd import capi, generators as G, petsc as P  # noqa: E402

scale = int(sys.argv[1])
Hs = [int(v) for v in sys.argv[2:]] or [262144]
comm = P.Comm(device=0, backend="self")
L = capi.load()
tm = B.Timer(L, comm.ctx)
t0 = time.time()
ai, aj, aa = G.rmat(scale)
m, nz = len(ai) - 1, int(ai[-1])
gen_s = time.time() - t0
x = np.linspace(-1.0, 1.0, m)
ref = None
L.phip_set_default_hot_columns(2)
for H in Hs:
    L.phip_set_default_route(2)
    L.phip_set_route_hot_columns(H)
    t0 = time.time()
    A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
    setup_s = time.time() - t0
    hA = A.device_mat(0)
    xv, yv = A.create_vecs()
    xv.set_array(x)
    dx, dy = xv.device_ptr(), yv.device_ptr()
    b, u, h, nc, nb = C.c_int(), C.c_int(), C.c_int32(), C.c_int64(), C.c_int32()
    L.phip_mat_route_info(hA, C.byref(b), C.byref(u), C.byref(h), C.byref(nc), C.byref(nb))
    for name, route, hot in (("route", 1, 0), ("hot order", 0, 1), ("plain", 0, 0)):
        if name != "route" and H != Hs[0]:
            continue
        L.phip_mat_set_route(hA, route)
        L.phip_mat_set_hot_columns(hA, hot)
        avg, pcts = tm.kernel(lambda: L.phip_mat_mult(hA, dx, dy), reps=50, warm=5)
        y = yv.array()
        if ref is None:
            ref = y
        print(json.dumps({"matrix": "rmat_%d" % scale, "rows": m, "nnz": nz, "mode": name, "H": h.value if route else None, "cold_nonzeros": nc.value if route else None,
                          "cold_share": round(nc.value / nz, 4) if route else None, "bins": nb.value if route else None, "us": round(avg, 1), "p50": pcts[1],
                          "G_gathers_per_s": round(nz / avg / 1e3, 1), "same_bits": bool(np.array_equal(y, ref)), "setup_s": round(setup_s, 2), "generate_s": round(gen_s, 1)}), flush=True)
    for o in (xv, yv, A):
        o.destroy()
comm.finalize()

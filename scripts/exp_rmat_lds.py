#!/usr/bin/env python3
"""R-MAT SpMV: the row-block kernel against the wave kernel with the hot columns in LDS (spmv_wave_lds_kernel).
   python scripts/exp_rmat_lds.py <scale> [<scale> ...]      one JSON line per scale"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi, generators as G  # noqa: E402

ctx = capi.Context(0)
L = ctx.lib
for scale in [int(s) for s in sys.argv[1:]] or [20]:
    t0 = time.time()
    ai, aj, aa = G.rmat(scale)
    tg = time.time() - t0
    trunc = int(os.environ.get("RMAT_TRUNC", "0"))
    if trunc:  # experiment: no row longer than trunc (what would the kernels do without the few very long rows?)
        lens = np.diff(ai)
        keep = (np.arange(len(aj)) - np.repeat(ai[:-1], lens)) < trunc
        aj, aa = aj[keep], aa[keep]
        ai = np.concatenate([[0], np.cumsum(np.minimum(lens, trunc))]).astype(np.int32)
    m, nz = len(ai) - 1, len(aj)
    t0 = time.time()
    A = ctx.mat_create(m, m, ai, aj, aa)
    tc = time.time() - t0
    b, u, K, sh = C.c_int(), C.c_int(), C.c_int32(), C.c_double()
    ctx.check(L.phip_mat_lds_hot_info(A, C.byref(b), C.byref(u), C.byref(K), C.byref(sh)))
    hb, hu, s0, s1 = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    ctx.check(L.phip_mat_hot_columns_info(A, C.byref(hb), C.byref(hu), C.byref(s0), C.byref(s1)))
    x = G.random_vector(m)
    dx, dy = ctx.to_device(x), ctx.to_device(np.zeros(m))
    out = {"scale": scale, "rows": m, "nnz": nz, "gen_s": round(tg, 1), "create_s": round(tc, 1), "lds_built": b.value, "lds_K": K.value, "lds_share": round(sh.value, 4),
           "hot_order_built": hb.value, "alg_bytes": 12 * nz + 20 * m + 4}
    ys = {}
    for name, on in (("lds", 1), ("rowblock", 0)):
        if on and not b.value:
            continue
        ctx.check(L.phip_mat_set_lds_hot(A, on))
        for _ in range(5):
            L.phip_mat_mult(A, dx, dy)
        ctx.sync()
        ms = min(ctx.time_region(lambda: L.phip_mat_mult(A, dx, dy), 20) for _ in range(3))
        ys[name] = ctx.to_host(dy, m)
        out[name + "_us"] = round(ms * 1e3, 1)
        out[name + "_GBps"] = round(out["alg_bytes"] / ms / 1e6, 1)
        out[name + "_Ggathers_per_s"] = round(nz / ms / 1e6, 1)
    if len(ys) == 2:
        out["max_rel_diff"] = float(np.abs(ys["lds"] - ys["rowblock"]).max() / np.abs(ys["rowblock"]).max())
    rows = np.arange(0, m, max(m // 2000, 1))
    ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], x[aj[ai[r]:ai[r + 1]]]) for r in rows])
    out["max_rel_err_vs_host_rows"] = float(np.abs(list(ys.values())[0][rows] - ref).max() / np.abs(ref).max())
    print(json.dumps(out), flush=True)
    ctx.free(dx); ctx.free(dy)
    ctx.check(L.phip_mat_destroy(A))
    del ai, aj, aa

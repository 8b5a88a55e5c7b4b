#!/usr/bin/env python3
"""Experiment: do the 17 / 31 equally aligned streams of VecMDot / VecMAXPY collide on HBM channels?  The vectors are carved out of
one allocation at a pitch of n*8 + skew bytes.  python scripts/exp_mdot_skew.py --n 7077888"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=192 ** 3)
    ap.add_argument("--nv", type=int, default=30)
    ap.add_argument("--skews", default="0,256,1024,4352,65792,1048832")
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    n, nv = a.n, a.nv
    for skew in [int(s) for s in a.skews.split(",")]:
        pitch = (n * 8 + 255) // 256 * 256 + skew
        base = ctx.malloc(pitch * (nv + 1) + 4096)
        addr = [base.value + k * pitch for k in range(nv + 1)]
        one = np.ones(n)
        for ad in addr:
            ctx.upload(C.c_void_p(ad), one)
        ys = (C.c_void_p * nv)(*[C.c_void_p(ad) for ad in addr[1:]])
        z = (C.c_double * nv)()
        alpha = (C.c_double * nv)(*([1e-9] * nv))
        x = C.c_void_p(addr[0])
        ctx.check(L.phip_vec_mdot(ctx.h, n, x, nv, ys, z))
        assert z[0] == float(n)
        dout = ctx.malloc(8 * 128)
        md = min(ctx.time_region(lambda: ctx.check(L.phip_vec_reduce_local(ctx.h, 0, n, x, nv, ys, dout)), 20) for _ in range(3))
        mx = min(ctx.time_region(lambda: ctx.check(L.phip_vec_maxpy(ctx.h, n, x, nv, alpha, ys)), 20) for _ in range(3))
        print(json.dumps({"n": n, "nv": nv, "skew_bytes": skew, "mdot_us": round(md * 1e3, 1), "mdot_GBps": round(8 * n * (nv + 1) / md / 1e6, 1),
                          "maxpy_us": round(mx * 1e3, 1), "maxpy_GBps": round(8 * n * (nv + 2) / mx / 1e6, 1)}), flush=True)
        ctx.free(dout); ctx.free(base)
    ctx.close()


if __name__ == "__main__":
    main()

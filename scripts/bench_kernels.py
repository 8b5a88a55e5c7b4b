#!/usr/bin/env python3
"""Kernel-level micro-benchmarks through the C ABI (developer tool; bench.py is the judged entry point).

    python scripts/bench_kernels.py [--grid 128] [--stencil 7|27] [--reps 200]
Prints effective (algorithmic) GB/s per kernel, timed with hipEvents on the library's stream.
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi, generators as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", type=int, default=128)
    ap.add_argument("--stencil", type=int, default=7)
    ap.add_argument("--reps", type=int, default=200)
    ap.add_argument("--tpr", type=int, default=0)
    ap.add_argument("--vec", type=int, default=1)
    ap.add_argument("--fmt", type=int, default=0)
    ap.add_argument("--rmat", type=int, default=0, help="R-MAT scale instead of a stencil")
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    print(json.dumps(ctx.device_info()))
    if a.rmat:
        cache = "/tmp/rmat_%d.npz" % a.rmat
        if os.path.exists(cache):
            z = np.load(cache)
            ai, aj, aa = z["i"], z["j"], z["a"]
        else:
            ai, aj, aa = G.rmat(a.rmat)
            np.savez(cache, i=ai, j=aj, a=aa)
    else:
        ai, aj, aa = (G.laplace3d_7pt if a.stencil == 7 else G.box3d_27pt)(a.grid)
    m = len(ai) - 1
    nz = len(aj)
    A = ctx.mat_create(m, m, ai, aj, aa)
    if a.tpr:
        ctx.check(L.phip_mat_set_threads_per_row(A, a.tpr))
    if a.fmt:
        ctx.check(L.phip_mat_set_format(A, a.fmt))
    print(json.dumps({k: (v if not isinstance(v, np.ndarray) else None) for k, v in ctx.mat_plan(A).items()}))
    x = G.random_vector(m)
    dx, dy, dz = ctx.to_device(x), ctx.to_device(np.zeros(m)), ctx.to_device(x[::-1].copy())
    res = {}

    def run(name, fn, nbytes):
        for _ in range(20):
            fn()
        ctx.sync()
        ms = min(ctx.time_region(fn, a.reps) for _ in range(3))
        res[name] = dict(us=round(ms * 1e3, 2), GBps=round(nbytes / ms / 1e6, 1), pct_of_8TBps=round(nbytes / ms / 1e6 / 80.0, 1))
        print(name, res[name], flush=True)

    run("spmv", lambda: L.phip_mat_mult(A, dx, dy), 12 * nz + 20 * m + 4)
    run("spmv_add", lambda: L.phip_mat_multadd(A, dx, dy, dy), 12 * nz + 28 * m + 4)
    import ctypes as C
    from petsc_amd.capi import Ticket
    tk = Ticket()
    run("spmv+dot(async)", lambda: L.phip_mat_mult_dot(A, dx, dy, dx, 0, C.byref(tk)), 12 * nz + 28 * m + 4)
    hv = (C.c_double * 8)()

    def spmv_dot_sync():
        L.phip_mat_mult_dot(A, dx, dy, dx, 1, C.byref(tk))
        L.phip_ticket_wait(ctx.h, C.byref(tk), hv)
    run("spmv+dot(sync)", spmv_dot_sync, 12 * nz + 28 * m + 4)
    tw = Ticket()
    run("pmult+3red(async)", lambda: L.phip_vec_pointwise_mult_reduce(ctx.h, m, dy, dx, dz, 0, C.byref(tw)), 24 * m)
    if a.vec:
        run("axpy", lambda: L.phip_vec_axpy(ctx.h, m, dy, 0.5, dx), 24 * m)
        run("aypx", lambda: L.phip_vec_aypx(ctx.h, m, dy, 0.5, dx), 24 * m)
        run("pointwise_mult", lambda: L.phip_vec_pointwise_mult(ctx.h, m, dy, dx, dz), 24 * m)
        run("copy", lambda: L.phip_vec_copy(ctx.h, m, dx, dy), 16 * m)
        import ctypes as C
        zz = (C.c_double * 64)()
        run("dot(sync)", lambda: L.phip_vec_dot(ctx.h, m, dx, dz, zz), 16 * m)
        run("norm(sync)", lambda: L.phip_vec_norm(ctx.h, m, dx, 1, zz), 8 * m)
        dout = ctx.malloc(8 * 128)
        ys1 = (C.c_void_p * 1)(dz)
        run("dot(async)", lambda: L.phip_vec_reduce_local(ctx.h, 0, m, dx, 1, ys1, dout), 16 * m)
        run("norm(async)", lambda: L.phip_vec_reduce_local(ctx.h, 1, m, dx, 1, None, dout), 8 * m)
        vs = [ctx.to_device(G.random_vector(m, 100 + k)) for k in range(30)]
        for nv in (2, 4, 8, 16, 30):
            arr = (C.c_void_p * nv)(*vs[:nv])
            al = np.linspace(0.1, 1.0, nv)
            run("mdot%d(async)" % nv, lambda: L.phip_vec_reduce_local(ctx.h, 0, m, dx, nv, arr, dout), 8 * m * (nv + 1))
            run("maxpy%d" % nv, lambda: L.phip_vec_maxpy(ctx.h, m, dy, nv, al.ctypes.data_as(C.POINTER(C.c_double)), arr), 8 * m * (nv + 2))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_kernels_%d_%d.json" % (a.stencil, a.grid), "w"), indent=1)
    ctx.close()


if __name__ == "__main__":
    main()

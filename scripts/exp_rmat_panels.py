#!/usr/bin/env python3
"""Experiment: R-MAT SpMV as a sum of column-panel products y = sum_p A_p x (each A_p a compressed-row CSR handled by the existing
kernels), so that the panel's slice of x stays in L2.  python scripts/exp_rmat_panels.py --scale 22 --panels 1,4,16,64"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi, generators as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=21)
    ap.add_argument("--panels", default="1,4,16,64")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    ai, aj, aa = G.rmat(a.scale)
    m = len(ai) - 1
    nz = len(aj)
    rows = np.repeat(np.arange(m, dtype=np.int32), np.diff(ai))
    x = G.random_vector(m)
    dx = ctx.to_device(x)
    ref = None
    for P in [int(p) for p in a.panels.split(",")]:
        W = (m + P - 1) // P
        mats = []
        pairs = 0
        pan = aj // W
        for p in range(P):
            sel = np.nonzero(pan == p)[0]          # row-major order is kept
            r, c, v = rows[sel], aj[sel], aa[sel]
            cnt = np.bincount(r, minlength=m).astype(np.int64)
            pi = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
            nzr = np.nonzero(cnt)[0].astype(np.int32)
            pairs += len(nzr)
            use = (m - len(nzr)) >= 0.6 * m
            cp = (np.concatenate([[0], np.cumsum(cnt[nzr])]).astype(np.int32), nzr) if use else None
            mats.append(ctx.mat_create(m, m, pi, c, v, cp))
        dy = ctx.to_device(np.zeros(m))

        def run():
            ctx.check(L.phip_mat_mult(mats[0], dx, dy))
            for A in mats[1:]:
                ctx.check(L.phip_mat_multadd(A, dx, dy, dy))
        run()
        y = ctx.to_host(dy, m)
        if ref is None:
            ref = y
        err = float(np.abs(y - ref).max() / np.abs(ref).max())
        ms = min(ctx.time_region(run, a.reps) for _ in range(3))
        print(json.dumps({"scale": a.scale, "panels": P, "panel_cols": W, "row_panel_pairs": pairs, "pairs_per_nnz": round(pairs / nz, 3),
                          "us": round(ms * 1e3, 1), "alg_GBps": round((12 * nz + 20 * m) / ms / 1e6, 1), "relerr_vs_1panel": err}), flush=True)
        for A in mats:
            ctx.check(L.phip_mat_destroy(A))
        ctx.free(dy)
    ctx.close()


if __name__ == "__main__":
    main()

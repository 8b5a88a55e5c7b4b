#!/usr/bin/env python3
"""Experiment: R-MAT SpMV with the columns relabelled by descending frequency (hot columns share cache lines), the nonzeros of a row kept
in their original order (so the row sums do not change).  Legs: the matrix as it is; relabelled by hand with x permuted on the host (the
kernel alone); the library's hot-column order (phip_set_default_hot_columns(2)), which pays for the pass over x inside the product.  python scripts/exp_rmat_relabel.py --scale 22"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi, generators as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=21)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    ai, aj, aa = G.rmat(a.scale)
    m = len(ai) - 1
    nz = len(aj)
    x = G.random_vector(m)
    cnt = np.bincount(aj, minlength=m)
    order = np.argsort(-cnt, kind="stable")           # order[k] = old column that becomes column k
    newcol = np.empty(m, dtype=np.int32)
    newcol[order] = np.arange(m, dtype=np.int32)
    lines = {"orig": len(np.unique(aj >> 4)), "relabel": len(np.unique(newcol[aj] >> 4))}
    ref = None
    ctx.check(L.phip_set_default_hot_columns(0))
    for name, cols, xv in (("orig", aj, x), ("relabel", newcol[aj], x[order]), ("library", aj, x)):
        if name == "library":  # the library's own hot-column order: includes the pass that copies x into the order
            ctx.check(L.phip_set_default_hot_columns(2))
        # share of the gathers that fall into the hottest 32768 lines (one XCD's L2 = 4 MiB)
        lc = np.bincount(cols >> 4)
        top = np.sort(lc)[::-1]
        share = float(top[:32768].sum() / nz)
        A = ctx.mat_create(m, m, ai, cols.astype(np.int32), aa, None)
        dx = ctx.to_device(np.ascontiguousarray(xv))
        dy = ctx.to_device(np.zeros(m))

        def run():
            ctx.check(L.phip_mat_mult(A, dx, dy))
        run()
        y = ctx.to_host(dy, m)
        if ref is None:
            ref = y
        ms = min(ctx.time_region(run, a.reps) for _ in range(3))
        print(json.dumps({"scale": a.scale, "columns": name, "us": round(ms * 1e3, 1), "alg_GBps": round((12 * nz + 20 * m) / ms / 1e6, 1),
                          "Ggathers_per_s": round(nz / ms / 1e6, 1), "share_top32k_lines": round(share, 3), "lines_touched": lines.get(name),
                          "bit_identical_to_orig": bool(np.array_equal(y, ref))}), flush=True)
        ctx.check(L.phip_mat_destroy(A))
        ctx.free(dx); ctx.free(dy)
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Experiment (access pattern only, results are NOT a product): what would the row-block kernel gain if the lanes of a wave gathered
x in ascending column order?  The column indices of every row block of the plan are sorted within the block (row membership is
ignored, so y is meaningless); matrix stream, LDS traffic and row sums stay as they are, only the gather addresses of neighbouring lanes
move together.  An upper bound for a layout that stores each block column-sorted with a 2-byte slot index per nonzero.
python scripts/exp_rmat_blocksort.py --scale 22"""
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
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    ctx.check(L.phip_set_default_hot_columns(0))
    ai, aj, aa = G.rmat(a.scale)
    m = len(ai) - 1
    nz = len(aj)
    x = G.random_vector(m)
    cnt = np.bincount(aj, minlength=m)
    order = np.argsort(-cnt, kind="stable")
    newcol = np.empty(m, dtype=np.int32)
    newcol[order] = np.arange(m, dtype=np.int32)
    dx = ctx.to_device(x)
    dy = ctx.to_device(np.zeros(m))
    for name, cols in (("orig", aj), ("hot", newcol[aj])):
        for sort in (0, 1):
            cj = cols.astype(np.int32)
            if sort:
                A0 = ctx.mat_create(m, m, ai, cj, aa, None)
                nb = C.c_int32()
                ctx.check(L.phip_mat_plan_info(A0, C.byref(nb), None, None, None))
                br = np.zeros(nb.value + 1, dtype=np.int32)
                ctx.check(L.phip_mat_plan_get(A0, br.ctypes.data_as(C.POINTER(C.c_int32))))
                ctx.check(L.phip_mat_destroy(A0))
                bid = np.repeat(np.arange(nb.value, dtype=np.int64), np.diff(ai[br].astype(np.int64)))
                cj = cj[np.lexsort((cj, bid))]
            A = ctx.mat_create(m, m, ai, cj, aa, None)

            def run():
                ctx.check(L.phip_mat_mult(A, dx, dy))
            run()
            ms = min(ctx.time_region(run, a.reps) for _ in range(3))
            print(json.dumps({"scale": a.scale, "columns": name, "sorted_within_block": bool(sort), "us": round(ms * 1e3, 1),
                              "Ggathers_per_s": round(nz / ms / 1e6, 1)}), flush=True)
            ctx.check(L.phip_mat_destroy(A))
    ctx.close()


if __name__ == "__main__":
    main()

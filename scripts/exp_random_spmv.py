#!/usr/bin/env python3
"""Experiment: where does the irregular-matrix SpMV time go?  Same rows x nnz as R-MAT scale S, but (a) uniformly random columns,
(b) columns confined to a window around the diagonal (gathers hit L1/L2), (c) R-MAT itself.  python scripts/exp_random_spmv.py --scale 21"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi, generators as G  # noqa: E402


def csr_from_pairs(m, r, c, rng):
    key = np.unique(r.astype(np.int64) * m + c)
    r, c = (key // m).astype(np.int32), (key % m).astype(np.int32)
    ai = np.concatenate([[0], np.cumsum(np.bincount(r, minlength=m))]).astype(np.int32)
    return ai, c, rng.uniform(0.0, 1.0, len(c))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=21)
    ap.add_argument("--deg", type=int, default=16)
    a = ap.parse_args()
    ctx = capi.Context(0)
    L = ctx.lib
    m = 1 << a.scale
    rng = np.random.default_rng(1)
    cases = []
    r = np.repeat(np.arange(m, dtype=np.int64), a.deg)
    cases.append(("uniform random columns", csr_from_pairs(m, r, rng.integers(0, m, len(r)), rng)))
    for w in (1 << 17, 1 << 12):
        c = np.clip(r + rng.integers(-w, w, len(r)), 0, m - 1)
        cases.append(("columns within +-%d of the diagonal" % w, csr_from_pairs(m, r, c, rng)))
    cases.append(("R-MAT", G.rmat(a.scale)))
    x = G.random_vector(m)
    dx = ctx.to_device(x)
    for name, (ai, aj, aa) in cases:
        nz = len(aj)
        A = ctx.mat_create(m, m, ai, aj, aa)
        dy = ctx.to_device(np.zeros(m))
        ctx.check(L.phip_mat_mult(A, dx, dy))
        ms = min(ctx.time_region(lambda: ctx.check(L.phip_mat_mult(A, dx, dy)), 20) for _ in range(3))
        print(json.dumps({"case": name, "rows": m, "nnz": nz, "max_row": int(np.diff(ai).max()), "us": round(ms * 1e3, 1),
                          "alg_GBps": round((12 * nz + 20 * m) / ms / 1e6, 1), "ns_per_nnz": round(ms * 1e6 / nz, 4)}), flush=True)
        ctx.check(L.phip_mat_destroy(A))
        ctx.free(dy)
    ctx.close()


if __name__ == "__main__":
    main()

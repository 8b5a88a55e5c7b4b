"""numpy derivation of MPIAIJ's split products, independent of the C oracle and of the library: ownership ranges (psplit.c:88),
garray (mmaij.c:69-101), compact off-diagonal column ids, and the VecScatter PtoS plan lists (vpscat.c:2327-2447).
Used by make_golden.py (fixtures of the reference's data files) and, at run time, by the tests of matrices too large to store plans for."""
import numpy as np


def ranges(N, size):
    n = [N // size + (1 if (N % size) > r else 0) for r in range(size)]
    return np.concatenate([[0], np.cumsum(n)]).astype(np.int32)


def split_products(ai, aj, N, size):
    out = {}
    rg = ranges(N, size)
    out["ranges"] = rg
    garrays = []
    for r in range(size):
        rs, re = int(rg[r]), int(rg[r + 1])
        cols = aj[ai[rs]:ai[re]]
        offm = (cols < rs) | (cols >= re)
        off = cols[offm]
        g = np.unique(off).astype(np.int32)
        garrays.append(g)
        out["garray_%d" % r] = g
        # compact ids of the off-diagonal entries, in storage order
        out["Bj_%d" % r] = np.searchsorted(g, off).astype(np.int32)
        rowlen_b = np.add.reduceat(np.concatenate([offm, [False]]).astype(np.int64), np.minimum(ai[rs:re] - ai[rs], len(offm)))[:re - rs] if re > rs else np.zeros(0, np.int64)
        rowlen_b = np.where(ai[rs + 1:re + 1] > ai[rs:re], rowlen_b, 0)
        out["Bi_%d" % r] = np.concatenate([[0], np.cumsum(rowlen_b)]).astype(np.int32)
    for r in range(size):
        rs, re = int(rg[r]), int(rg[r + 1])
        g = garrays[r]
        owner = np.searchsorted(rg, g, side="right") - 1
        procs, counts = np.unique(owner, return_counts=True)
        out["from_procs_%d" % r] = procs.astype(np.int32)
        out["from_starts_%d" % r] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        out["from_indices_%d" % r] = np.arange(len(g), dtype=np.int32)
        tp, ts, ti = [], [0], []
        for p in range(size):
            if p == r:
                continue
            gp = garrays[p]
            mine = gp[(gp >= rs) & (gp < re)] - rs
            if len(mine):
                tp.append(p)
                ti.append(mine)
                ts.append(ts[-1] + len(mine))
        out["to_procs_%d" % r] = np.array(tp, dtype=np.int32)
        out["to_starts_%d" % r] = np.array(ts, dtype=np.int32)
        out["to_indices_%d" % r] = (np.concatenate(ti) if ti else np.zeros(0)).astype(np.int32)
    return out

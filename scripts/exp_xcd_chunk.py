#!/usr/bin/env python3
"""XCD chunk sweep (phip_set_xcd_chunk) of the CSR row-block and SELL-dict kernels on out-of-cache matrices with perturbed values.
usage: python scripts/exp_xcd_chunk.py [lap7_256 box27_192 ...]  -> one JSON line per (matrix, layout, chunk)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B  # noqa: E402
from petsc_amd import capi, petsc as P  # noqa: E402


def main():
    names = sys.argv[1:] or ["lap7_256"]
    comm = P.Comm(device=0, backend="self")
    L = capi.load()
    tm = B.Timer(L, comm.ctx)
    for name in names:
        kind, n, _ = B.parse_workload(name)
        N, m, rs, ai, aj, aa = B.local_rows(kind, n, 1, 0, "strong", True)
        nz = int(ai[-1])
        A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
        hA = A.device_mat(0)
        x, y = A.create_vecs()
        x.set_array(np.linspace(-1.0, 1.0, m))
        dx, dy = x.device_ptr(), y.device_ptr()
        ref = None
        for fmt in [int(v) for v in os.environ.get('FMTS', '1,0').split(',')]:
            assert L.phip_mat_set_format(hA, fmt) == 0
            kernel, layout, real, alg = B.layout_of(L, hA, m, m, nz)
            for chunk in [int(v) for v in os.environ.get('CHUNKS', '1,2,4,8,16,32,64,128').split(',')]:
                assert L.phip_set_xcd_chunk(chunk) == 0
                avg, pcts = tm.kernel(lambda: L.phip_mat_mult(hA, dx, dy), reps=100, warm=10)
                yv = y.array()
                if ref is None:
                    ref = yv
                print(json.dumps({"matrix": name + "_var", "kernel": kernel, "chunk": chunk, "us": round(avg, 1), "p50": pcts[1],
                                  "frac_of_8TBps_on_algorithmic_bytes": round(alg / (avg * 1e-6) / 8e12, 4),
                                  "frac_on_layout_bytes": round(real / (avg * 1e-6) / 8e12, 4), "same_bits": bool(np.array_equal(yv, ref))}), flush=True)
        L.phip_set_xcd_chunk(0)
        for o in (x, y, A):
            o.destroy()
    comm.finalize()


if __name__ == "__main__":
    main()

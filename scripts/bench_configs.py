#!/usr/bin/env python3
"""One-GPU measurements of the BASELINE.json configs other than the judged bench line (developer tool; bench.py is the judged one).

    python scripts/bench_configs.py [--configs 1,3,4,5] [--grid3 256] [--grid4 192] [--rmat-scale 22]
Prints one JSON line per config.  All GPU work goes through the PETSc-API mirror (MATAIJHIP / VECHIP).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from petsc_amd import capi, generators as G, petsc as P  # noqa: E402


def timed_solve(comm, L, A, b, x, ksp_type, its, **kw):
    ksp = P.KSP(comm, ksp_type, "jacobi")
    ksp.set_operators(A)
    if ksp_type == "gmres":
        ksp.gmres_set_restart(kw.get("restart", 30))
        ksp.gmres_set_cgs_refinement(kw.get("refine", P.REFINE_NEVER))
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=max(its // 10, 2))
    ksp.solve(b, x)
    ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=its)
    L.phip_device_sync(comm.ctx)
    t = time.perf_counter()
    res = ksp.solve(b, x)
    L.phip_device_sync(comm.ctx)
    dt = time.perf_counter() - t
    ksp.destroy()
    return res, dt


def spmv_b2b(comm, L, A, x, y, reps=100):
    ctxw = capi.Context.__new__(capi.Context)
    ctxw.lib, ctxw.h = L, comm.ctx
    for _ in range(10):
        A.mult(x, y)
    L.phip_device_sync(comm.ctx)
    return min(ctxw.time_region(lambda: A.mult(x, y), reps) for _ in range(3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,3,4,5")
    ap.add_argument("--grid1", type=int, default=1000)
    ap.add_argument("--grid3", type=int, default=256)
    ap.add_argument("--grid4", type=int, default=192)
    ap.add_argument("--rmat-scale", type=int, default=22)
    ap.add_argument("--its", type=int, default=60)
    a = ap.parse_args()
    todo = [int(c) for c in a.configs.split(",")]
    comm = P.Comm(device=0, backend="self")
    L = capi.load()
    for cfg in todo:
        t0 = time.time()
        if cfg == 1:
            ai, aj, aa = G.laplace2d_5pt(a.grid1, a.grid1)
            name = "ex2 2-D 5-point %d^2, KSPCG+PCJACOBI rtol 1e-2/((m+1)(n+1))" % a.grid1
        elif cfg == 3:
            ai, aj, aa = G.box3d_27pt(a.grid3)
            name = "3-D 27-point %d^3, KSPCG+PCJACOBI" % a.grid3
        elif cfg == 4:
            ai, aj, aa = G.convdiff3d_7pt(a.grid4)
            name = "3-D convection-diffusion %d^3, KSPGMRES(30)+PCJACOBI" % a.grid4
        else:
            ai, aj, aa = G.rmat(a.rmat_scale)
            name = "R-MAT scale %d, SpMV" % a.rmat_scale
        gen_s = time.time() - t0
        m, nz = len(ai) - 1, len(aj)
        t0 = time.time()
        A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
        up_s = time.time() - t0
        x, b = A.create_vecs()
        y = x.duplicate()
        x.set_array(G.random_vector(m))
        spmv_ms = spmv_b2b(comm, L, A, x, y)
        sp_bytes = 12 * nz + 20 * m + 4
        # as bench.py: GB/s only on bytes the kernel really moves in its device layout (never above the peak); the SURVEY 8d CSR bytes as a ratio
        kernel, _layout, real_bytes, _alg = B.layout_of(L, A.device_mat(0), m, m, nz)
        out = {"config": cfg, "name": name, "rows": m, "nnz": nz, "generate_s": round(gen_s, 1), "assemble_upload_s": round(up_s, 1),
               "spmv_kernel": kernel, "spmv_us": round(spmv_ms * 1e3, 1), "spmv_real_GBps": round(real_bytes / spmv_ms / 1e6, 1),
               "spmv_pct_of_8TBps": round(real_bytes / spmv_ms / 1e6 / 80, 1), "algorithmic_over_layout_bytes": round(sp_bytes / float(real_bytes), 2)}
        if cfg in (1, 3):
            x.set(1.0)
            A.mult(x, b)
            if cfg == 1:
                ksp = P.KSP(comm, "cg", "jacobi")
                ksp.set_operators(A)
                ksp.set_tolerances(rtol=1e-2 / ((a.grid1 + 1) * (a.grid1 + 1)))
                ksp.solve(b, x)
                L.phip_device_sync(comm.ctx)
                t = time.perf_counter()
                res = ksp.solve(b, x)
                L.phip_device_sync(comm.ctx)
                dt = time.perf_counter() - t
                ksp.destroy()
                out.update(cg_its=res["its"], cg_reason=res["reason"], cg_rnorm=res["rnorm"], cg_solve_s=round(dt, 4), cg_iters_per_s=round(res["its"] / dt, 1))
                # the reference path of this config: the oracle on one host core
                from oracle.oracle import Oracle
                op = Oracle(fast=True).operator(ai, aj, aa, 1)
                bh = op.mult(np.ones(m))
                t = time.perf_counter()
                ref = op.solve(bh, "cg", "jacobi", rtol=1e-2 / ((a.grid1 + 1) * (a.grid1 + 1)))
                dtc = time.perf_counter() - t
                out.update(cpu_its=ref["its"], cpu_rnorm=ref["rnorm"], cpu_solve_s=round(dtc, 2), cpu_iters_per_s=round(ref["its"] / dtc, 1),
                           its_match=abs(ref["its"] - res["its"]) <= 1, rnorm_rel_diff=abs(ref["rnorm"] - res["rnorm"]) / ref["history"][0])
            else:
                res, dt = timed_solve(comm, L, A, b, x, "cg", a.its)
                it_bytes = sp_bytes + 136 * m
                out.update(cg_its=res["its"], cg_iters_per_s=round(res["its"] / dt, 1), cg_csr_equivalent_GBps=round(it_bytes * res["its"] / dt / 1e9, 1))  # SURVEY 8d bytes of a CSR implementation / time: a rate of work, not of memory traffic
        elif cfg == 4:
            x.set(1.0)
            A.mult(x, b)
            res, dt = timed_solve(comm, L, A, b, x, "gmres", a.its)
            # per inner iteration it: SpMV + PCApply 24m + mdot(it+1) 8m(it+2) + maxpy(it+1) 8m(it+3) + norm 8m + scale 16m
            tot = 0
            for k in range(res["its"]):
                it = k % 30
                tot += sp_bytes + 24 * m + 8 * m * (it + 2) + 8 * m * (it + 3) + 24 * m
            out.update(gmres_its=res["its"], gmres_iters_per_s=round(res["its"] / dt, 1), gmres_csr_equivalent_GBps=round(tot / dt / 1e9, 1))
        else:
            lens = np.diff(ai)
            per_rank = [int(ai[(r + 1) * m // 8] - ai[r * m // 8]) for r in range(8)]
            out.update(max_row=int(lens.max()), empty_rows=int((lens == 0).sum()), nnz_imbalance_8ranks=round(max(per_rank) / (nz / 8), 3))
        print(json.dumps(out), flush=True)
        for o in (x, y, b, A):
            o.destroy()
        del ai, aj, aa
    comm.finalize()


if __name__ == "__main__":
    main()

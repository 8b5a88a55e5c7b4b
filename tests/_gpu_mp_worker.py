"""Worker of tests/test_host_gpu.py::test_multiprocess_shared_gpu and ::test_multiprocess_rccl: one PROCESS per rank.  Backend "shm":
all ranks on GPU 0 (shared-memory ring for the host collectives, ghost exchange staged through the host).  Backend "rccl": one GPU
per rank, the data path of bench.py --gpus N (grouped ncclSend/ncclRecv ghost exchange, host ring or ncclAllReduce reductions).
argv: key(hex) nranks rank [backend [device]]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    key, n, r = bytes.fromhex(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    backend = sys.argv[4] if len(sys.argv) > 4 else "shm"
    device = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    from petsc_amd import generators as G, petsc as P
    from oracle.oracle import Oracle
    orc = Oracle()
    comm = P.Comm(device=device, backend=backend, id128=key, nranks=n, rank=r)
    ai, aj, aa = G.convdiff3d_7pt(12, 10, 9)
    N = len(ai) - 1
    op = orc.operator(ai, aj, aa, n)
    rg = orc.ownership_ranges(N, n)
    rs, re = int(rg[r]), int(rg[r + 1])
    xg = np.cos(np.arange(N) * 0.37)
    A = P.Mat.from_global_csr(comm, ai, aj, aa)
    x = P.Vec.from_array(comm, xg[rs:re], N)
    y = x.duplicate()
    A.mult(x, y)
    yref = op.mult(xg)
    assert np.abs(y.array() - yref[rs:re]).max() <= 1e-12 * np.abs(yref).max()
    for k in range(25):   # consecutive exchanges through one plan: buffers and signals are reused, nothing may be overtaken
        x.scale(-1.5 if k % 2 else 0.5)
        xg = xg * (-1.5 if k % 2 else 0.5)
        A.mult(x, y)
        yk = op.mult(xg)
        assert np.abs(y.array() - yk[rs:re]).max() <= 1e-12 * np.abs(yk).max(), k
    A.mult_transpose(x, y)
    tref = op.mult_transpose(xg)
    assert np.abs(y.array() - tref[rs:re]).max() <= 1e-12 * np.abs(tref).max()
    assert abs(x.dot(x) - xg @ xg) <= 1e-12 * (xg @ xg) and x.norm(P.NORM_INFINITY) == np.abs(xg).max()
    b_g = op.mult(np.ones(N))
    b = P.Vec.from_array(comm, b_g[rs:re], N)
    for ksp_type in ("gmres", "cg"):
        if ksp_type == "cg":
            sym = G.laplace3d_7pt(11, dirichlet_rows=False)
            for o in (A, x, y, b):
                o.destroy()
            N = len(sym[0]) - 1
            op = orc.operator(*sym, n)
            rg = orc.ownership_ranges(N, n)
            rs, re = int(rg[r]), int(rg[r + 1])
            A = P.Mat.from_global_csr(comm, *sym)
            b_g = op.mult(np.ones(N))
            b = P.Vec.from_array(comm, b_g[rs:re], N)
        sol = b.duplicate()
        ksp = P.KSP(comm, ksp_type, "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-9)
        res = ksp.solve(b, sol)
        ref = op.solve(b_g, ksp_type, "jacobi", rtol=1e-9)
        assert abs(res["its"] - ref["its"]) <= 1 and res["reason"] == ref["reason"], (res, ref["its"])
        assert np.abs(sol.array() - ref["x"][rs:re]).max() <= 1e-8
        ksp.destroy(); sol.destroy()
    import ctypes as C
    from petsc_amd import capi
    cnt = C.c_ulonglong()
    capi.load().phip_halo_ipc_count(C.byref(cnt))
    mode = os.environ.get("PHIP_HALO_IPC")        # "1": peer writes or an error; "0": never; unset: tried with rccl, kept if the trial passes
    if mode == "1":
        assert cnt.value > 30, "peer-write exchanges: %d with PHIP_HALO_IPC=1" % cnt.value
    elif mode == "0" or backend != "rccl":
        assert cnt.value == 0, "peer-write exchanges: %d although not asked for" % cnt.value
    print("ghost exchange of rank %d: %s (%d peer-write exchanges)" % (r, "peer writes" if cnt.value > 30 else "backend path", cnt.value))
    comm.finalize()
    print("mp ok rank %d" % r)


if __name__ == "__main__":
    main()

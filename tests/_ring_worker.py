"""Worker of tests/test_host_ring.py: one process per rank of a "shm" communicator (include/petschip.h): the host reduction ring that
the rccl backend uses for dot products and norms, without any device.  argv: key(hex) nranks rank rounds."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    key, n, r, rounds = bytes.fromhex(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from petsc_amd import capi
    L = capi.load()
    ctx = C.c_void_p()
    assert L.phip_init(-1, C.byref(ctx)) == 0
    comm = C.c_void_p()
    rc = L.phip_comm_create(ctx, b"shm", key, n, r, C.byref(comm))
    assert rc == 0, capi.last_error()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(99)          # same stream on every rank: rank p's contribution to round k is table[k, p]
    table = rng.standard_normal((rounds, n, 5))
    for k in range(rounds):
        nv = 1 + k % 5
        v = table[k, r, :nv].copy()
        assert L.phip_comm_allreduce_host(comm, dp(v), nv, 0) == 0, capi.last_error()
        ref = table[k, 0, :nv].copy()
        for p in range(1, n):
            ref += table[k, p, :nv]          # rank order, as the library combines
        assert np.array_equal(v, ref), (k, v, ref)
        if k % 7 == 0:
            m = np.array([table[k, r, 0], float("nan") if (k % 21 == 0 and r == n - 1) else 0.0])
            assert L.phip_comm_allreduce_host(comm, dp(m), 2, 1) == 0
            assert m[0] == table[k, :, 0].max() and (np.isnan(m[1]) if k % 21 == 0 else m[1] == 0.0)
    # all-gather of a payload longer than one ring slot (chunked), then a barrier
    mine = np.full(700, float(r)) + np.arange(700)
    out = np.zeros(700 * n)
    assert L.phip_comm_allgather_host(comm, mine.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), mine.nbytes) == 0
    for p in range(n):
        assert np.array_equal(out[700 * p:700 * (p + 1)], np.full(700, float(p)) + np.arange(700))
    assert L.phip_comm_barrier(comm) == 0
    d = C.c_void_p()
    assert L.phip_comm_allreduce(comm, d, 1, 0) != 0   # no device traffic on this backend
    assert L.phip_comm_destroy(comm) == 0
    assert L.phip_finalize(ctx) == 0
    print("ring ok rank %d" % r)


if __name__ == "__main__":
    main()

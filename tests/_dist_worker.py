"""Worker of the multi-process CPU tests (launched by tests/test_host_cpu.py, one process per rank, gloo).

Planning-only contexts (no GPU): every rank assembles its part of an MPIAIJHIP matrix from a golden fixture and checks the
index work -- row ownership, diagonal / off-diagonal split, garray, compact column ids, compressed rows, the VecScatter PtoS
plan -- bit for bit against the fixture and against the oracle.  mode "stash": rank 0 inserts EVERY entry, so all other
ranks' rows travel through the stash exchange (mpiaij.c:619-629, 680-737).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    name, mode = sys.argv[1], sys.argv[2]
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from petsc_amd import petsc as P, torchcomm
    from oracle.oracle import Oracle
    orc = Oracle()
    if os.environ.get("PHIP_TEST_COMM") == "shm":
        # the library's own multi-process communicator (shared-memory ring): all-gather and all-to-all of the plan handshake through it
        comm = P.Comm(device=-1, backend="shm", id128=bytes.fromhex(os.environ["PHIP_TEST_KEY"]), nranks=world, rank=rank)
    else:
        ag, av = torchcomm.make_callbacks(dist)
        comm = P.Comm.from_callbacks(-1, world, rank, ag, av)
    g = np.load(os.path.join(ROOT, "tests", "golden", "mat_%s.npz" % name), allow_pickle=False)
    ai, aj, aa, N = g["i"], g["j"], g["a"], int(g["N"])
    rg = orc.ownership_ranges(N, world)
    rs, re = int(rg[rank]), int(rg[rank + 1])
    if mode == "arrays":
        A = P.Mat.from_global_csr(comm, ai, aj, aa)
    else:
        A = P.Mat(comm, P.PETSC_DECIDE, P.PETSC_DECIDE, N, N, "aijhip")
        assert A.type() == "mpiaijhip"
        A.mpiaij_set_preallocation(3, None, 1, None)  # too small on purpose: rows must grow (aij.h:148-179)
        if mode == "local":
            perm = np.random.default_rng(rank).permutation(np.arange(ai[rs], ai[re]))
            rows = np.repeat(np.arange(N, dtype=np.int32), np.diff(ai))
            for k in perm:
                A.set_values([rows[k]], [aj[k]], [aa[k]], P.INSERT_VALUES)
        elif mode == "stash":
            if rank == 0:
                for r in range(N):
                    A.set_values([r], aj[ai[r]:ai[r + 1]], aa[ai[r]:ai[r + 1]], P.ADD_VALUES)
        A.assemble()
    assert A.ownership_range() == (rs, re)
    sp = orc.mpiaij_split(ai[rs:re + 1] - ai[rs], aj[ai[rs]:ai[re]], aa[ai[rs]:ai[re]], N, rs, re)
    a_blk, b_blk, gh = A.block(0), A.block(1), A.ghosts()
    for key, arr in (("i", sp["Ai"]), ("j", sp["Aj"]), ("a", sp["Aa"])):
        np.testing.assert_array_equal(a_blk[key], arr, err_msg="A." + key)
    for key, arr in (("i", sp["Bi"]), ("j", sp["Bj"]), ("a", sp["Ba"])):
        np.testing.assert_array_equal(b_blk[key], arr, err_msg="B." + key)
    np.testing.assert_array_equal(gh["garray"], sp["garray"])
    pre = "s%d_" % world
    if pre + "ranges" in g.files:
        np.testing.assert_array_equal(gh["garray"], g[pre + "garray_%d" % rank])
        np.testing.assert_array_equal(b_blk["j"], g[pre + "Bj_%d" % rank])
        np.testing.assert_array_equal(b_blk["i"], g[pre + "Bi_%d" % rank])
        for key in ("from_procs", "from_starts", "to_procs", "to_starts", "to_indices"):
            np.testing.assert_array_equal(gh[key], g[pre + "%s_%d" % (key, rank)], err_msg=key)
    use, cpi, ridx = orc.check_compressed_row(sp["Bi"])
    assert b_blk["cprow_use"] == use
    if use:
        np.testing.assert_array_equal(b_blk["cprow_i"], cpi)
        np.testing.assert_array_equal(b_blk["cprow_rindex"], ridx)
    np.testing.assert_array_equal(a_blk["diag"], orc.mark_diagonal(sp["Ai"], sp["Aj"]))
    # numerical entry points must fail loudly without a GPU (no CPU fallback)
    x, y = A.create_vecs()
    try:
        A.mult(x, y)
        raise AssertionError("MatMult ran without a GPU")
    except P.PetscError as e:
        assert e.code == 56, e
    A.destroy()
    comm.finalize()
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()

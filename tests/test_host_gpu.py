"""GPU parity tests of the PETSc-API mirror (libpetschip_host.so -> libpetschip.so) against the oracle and the reference's
golden outputs.  Bars (BASELINE.json): index work bit-exact; SpMV <= 1e-12 relative; KSP +-1 iteration and <= 1e-10 relative
final residual (relative to the initial residual norm); 6-digit -ksp_monitor_short histories equal to the reference's files.

Multi-rank cases use the "threads" communicator: the ranks are host threads of this process sharing the one GPU, so the
whole distributed code path (row split, stash, garray, scatter plan, device pack kernel, lvec offsets, MultAdd on the
compressed-row off-diagonal block, all-reduce ordering) runs on a single-GPU box.
"""
import os
import struct
import threading

import numpy as np
import pytest

from oracle.oracle import REFINE_ALWAYS, REFINE_IFNEEDED, REFINE_NEVER

pytestmark = pytest.mark.gpu

_key = [1000]


def run_ranks(n, fn):
    """Run fn(comm) on n ranks (threads backend; n == 1 -> self). Returns the list of results in rank order."""
    from petsc_amd import petsc as P
    if n == 1:
        comm = P.Comm(device=0, backend="self")
        try:
            return [fn(comm)]
        finally:
            comm.finalize()
    _key[0] += 1
    key = struct.pack("<Q", _key[0]) + bytes(120)
    res, err = [None] * n, [None] * n

    def work(r):
        comm = None
        try:
            comm = P.Comm(device=0, backend="threads", id128=key, nranks=n, rank=r)
            res[r] = fn(comm)
        except BaseException as e:  # noqa: BLE001
            err[r] = e
        finally:
            if comm is not None and err[r] is None:
                comm.finalize()

    ts = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    for r in range(n):
        if err[r] is not None:
            raise err[r]
    return res


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, "mat_%s.npz" % name), allow_pickle=False)


def gather(parts):
    return np.concatenate(parts)


def local_slice(comm, N):
    m = N // comm.size + (1 if (N % comm.size) > comm.rank else 0)
    rs = sum(N // comm.size + (1 if (N % comm.size) > r else 0) for r in range(comm.rank))
    return rs, rs + m


def monitor_short(v):
    if v > 1e-9:
        return "%g" % v
    if v > 1e-11:
        return "%5.3e" % v
    return "< 1.e-11"


def ex18_vectors(n=15):
    """the two vectors of src/vec/vec/examples/tests/ex18.c:27-32"""
    i = np.arange(n, dtype=np.float64)
    x = i + 1.0 / (i + .35)
    return x, x + 1.375547826473644376


def test_vecdot_known_answer_ex18():
    """src/vec/vec/examples/tests/ex18.c ("Compares BLAS dots on different machines"): VecDot of two 15-entry vectors must print as
    tests/output/ex18_1.out, 13 significant digits.  On one rank (as the reference runs it) and split over 2 and 3."""
    from petsc_amd import petsc as P
    xg, yg = ex18_vectors()

    def body(comm):
        rs, re = local_slice(comm, len(xg))
        x = P.Vec.from_array(comm, xg[rs:re], len(xg))
        y = P.Vec.from_array(comm, yg[rs:re], len(xg))
        v = x.dot(y)
        x.destroy(); y.destroy()
        return v

    for size in (1, 2, 3):
        for v in run_ranks(size, body):
            assert "Vector inner product %16.12e" % v == "Vector inner product 1.202435522115e+03"


# ---------------------------------------------------------------------------------------------------------------------
def test_vec_closed_forms_ex1():
    """src/vec/vec/examples/tutorials/ex1.c:71-153 (golden output ex1_1.out: every error prints 0)."""
    from petsc_amd import petsc as P

    def body(comm):
        n = 20
        x = P.Vec(comm, P.PETSC_DECIDE, n, "hip")
        y, w = x.duplicate(), x.duplicate()
        z = [x.duplicate() for _ in range(3)]
        one, two, three = 1.0, 2.0, 3.0
        x.set(one); y.set(two); z[0].set(one); z[1].set(two); z[2].set(three)
        assert x.dot(x) == n
        np.testing.assert_array_equal(x.mdot(z), [n, 2 * n, 3 * n])
        x.scale(two)
        assert abs(x.norm() - 2 * np.sqrt(n)) < 1e-14 * n
        y.copy_to(w)
        assert abs(w.norm() - 2 * np.sqrt(n)) < 1e-14 * n
        y.axpy(three, x)
        assert abs(y.norm() - 8 * np.sqrt(n)) < 1e-13 * n
        y.aypx(two, x)
        assert abs(y.norm() - 18 * np.sqrt(n)) < 1e-13 * n
        x.swap(y)
        assert abs(y.norm() - 2 * np.sqrt(n)) < 1e-13 * n and abs(x.norm() - 18 * np.sqrt(n)) < 1e-13 * n
        w.waxpy(two, x, y)
        assert abs(w.norm() - 38 * np.sqrt(n)) < 1e-12 * n
        w.pointwise_mult(y, x)
        assert abs(w.norm() - 36 * np.sqrt(n)) < 1e-12 * n
        w.pointwise_divide(x, y)
        assert abs(w.norm() - 9 * np.sqrt(n)) < 1e-12 * n
        x.set(one)
        x.maxpy([one, two, three], z)  # 1 + 1 + 4 + 9
        assert abs(x.norm() - 15 * np.sqrt(n)) < 1e-12 * n
        assert x.norm(P.NORM_1) == 15.0 * n and x.norm(P.NORM_INFINITY) == 15.0
        n1, n2 = x.norm(P.NORM_1_AND_2)
        assert n1 == 15.0 * n and abs(n2 - 15 * np.sqrt(n)) < 1e-12 * n
        assert abs(x.normalize() - 15 * np.sqrt(n)) < 1e-12 * n and abs(x.norm() - 1.0) < 1e-14
        # VecDotNorm2 (vinv.c:1369-1413): (s.t, t.t); VecMTDot = VecMDot for real scalars
        x.set(two); y.set(three)
        assert x.dot_norm2(y) == (6.0 * n, 9.0 * n)
        np.testing.assert_array_equal(x.mtdot(z), x.mdot(z))
        with pytest.raises(P.PetscError) as e:
            x.axpy(1.0, x)
        assert e.value.code == 61
        for v in [x, y, w] + z:
            v.destroy()
        return True

    for size in (1, 2, 3):
        assert all(run_ranks(size, body))


def test_setrandom_placearray():
    """VecSetRandom with the default rand48 PetscRandom (rand48.c:6-42; seed 0x12345678 + 76543*rank, randomc.c:347) reproduces glibc's
    srand48/drand48 sequence; VecPlaceArray / VecResetArray (rvector.c:1955-2030) swap the values in and out."""
    import ctypes as C
    from petsc_amd import petsc as P
    libc = C.CDLL(None)
    libc.drand48.restype = C.c_double

    def rand48(seed, n):
        libc.srand48(C.c_long(seed))
        return np.array([libc.drand48() for _ in range(n)])

    N = 1000

    def body(comm):
        x = P.Vec(comm, P.PETSC_DECIDE, N, "hip")
        x.set_random()
        a = x.array().copy()
        x.set_random(seed=42, interval=(-2.0, 3.0))
        b = x.array().copy()
        mine = np.arange(x.local_size, dtype=np.float64) + 100.0 * comm.rank
        x.set(7.0)
        x.place_array(mine)
        assert np.array_equal(x.array(), mine)
        x.scale(2.0)                       # computes on the placed values ...
        nrm = x.norm(P.NORM_INFINITY)
        x.reset_array()
        assert np.array_equal(mine, 2.0 * (np.arange(len(mine)) + 100.0 * comm.rank))   # ... which the caller's array holds afterwards
        assert np.array_equal(x.array(), np.full(len(mine), 7.0))                       # the vector's own values are back
        with pytest.raises(P.PetscError):
            x.place_array(mine); x.place_array(mine)
        x.reset_array()
        x.destroy()
        return a, b, nrm

    for size in (1, 3):
        res = run_ranks(size, body)
        for r, (a, b, nrm) in enumerate(res):
            np.testing.assert_array_equal(a, rand48(0x12345678 + 76543 * r, len(a)))
            np.testing.assert_array_equal(b, 5.0 * rand48(42, len(b)) + -2.0)
        assert res[0][2] == max(2.0 * (len(a) - 1 + 100.0 * r) for r, (a, _, _) in enumerate(res))


@pytest.mark.parametrize("name", ["tiny", "small", "medium", "arco1", "poisson1", "spd-real-int32-float64", "ns-real-int32-float64", "poisson2", "fem1", "afiro", "golan", "A4"])
@pytest.mark.parametrize("size", [1, 2, 3, 4])
def test_matmult_golden(orc, golden_dir, name, size):
    from petsc_amd import petsc as P
    g = load(golden_dir, name)
    N = int(g["N"])
    if int(g["M"]) != N:
        pytest.skip("rectangular")
    if size > 1 and "s%d_ranges" % size not in g.files:
        pytest.skip("no fixture for this split")

    def body(comm):
        rs, re = local_slice(comm, N)
        A = P.Mat.from_global_csr(comm, g["i"], g["j"], g["a"])
        x = P.Vec.from_array(comm, g["x"][rs:re], N)
        y = x.duplicate()
        A.mult(x, y)
        out = [y.array()]
        y0 = P.Vec.from_array(comm, g["y0"][rs:re], N)
        z = x.duplicate()
        A.mult_add(x, y0, z)
        out.append(z.array())
        A.mult_add(x, y0, y0)  # in place, as MatMult_MPIAIJ calls it (mpiaij.c:996)
        out.append(y0.array())
        d = x.duplicate()
        A.get_diagonal(d)
        out.append(d.array())
        if comm.size > 1:
            gh = A.ghosts()
            pre = "s%d_" % comm.size
            np.testing.assert_array_equal(gh["garray"], g[pre + "garray_%d" % comm.rank])
            for key in ("from_procs", "from_starts", "to_procs", "to_starts", "to_indices"):
                np.testing.assert_array_equal(gh[key], g[pre + "%s_%d" % (key, comm.rank)], err_msg=key)
        for o in (x, y, y0, z, d, A):
            o.destroy()
        return out

    res = run_ranks(size, body)
    y = gather([r[0] for r in res]); z = gather([r[1] for r in res]); z2 = gather([r[2] for r in res]); d = gather([r[3] for r in res])
    scale = np.abs(g["y"]).max() + 1e-300
    assert np.abs(y - g["y"]).max() <= 1e-12 * scale
    zo = orc.matmultadd(g["i"], g["j"], g["a"], g["x"], g["y0"])
    assert np.abs(z - zo).max() <= 1e-12 * (scale + 1) and np.abs(z2 - zo).max() <= 1e-12 * (scale + 1)
    np.testing.assert_array_equal(d, g["diag"])


@pytest.mark.parametrize("name", ["small", "medium", "arco1", "poisson1", "ns-real-int32-float64", "poisson2", "fem1", "golan", "A4"])
@pytest.mark.parametrize("size", [1, 2, 3, 4])
def test_matmulttranspose_golden(orc, golden_dir, name, size):
    """MatMultTranspose / MatMultTransposeAdd on Seq and MPI matrices (aij.c:1207-1266, mpiaij.c:1029-1056, 1102-1117), the MPI form
    with its reverse ADD scatter, on the reference's data files, 1-4 ranks; then the same after new values with the old pattern."""
    import scipy.sparse as sp
    from petsc_amd import petsc as P
    g = load(golden_dir, name)
    N = int(g["N"])
    rng = np.random.default_rng(17)
    xg, y0g = rng.standard_normal(N), rng.standard_normal(N)
    a2 = g["a"] * rng.uniform(0.5, 2.0, len(g["a"]))

    def body(comm):
        rs, re = local_slice(comm, N)
        A = P.Mat.from_global_csr(comm, g["i"], g["j"], g["a"])
        x = P.Vec.from_array(comm, xg[rs:re], N)
        y = x.duplicate()
        A.mult_transpose(x, y)
        out = [y.array()]
        y0 = P.Vec.from_array(comm, y0g[rs:re], N)
        z = x.duplicate()
        A.mult_transpose_add(x, y0, z)
        out.append(z.array())
        A.mult_transpose_add(x, y0, y0)
        out.append(y0.array())
        with pytest.raises(P.PetscError) as e:
            A.mult_transpose(x, x)
        assert e.value.code == 73
        # same nonzero pattern, new values: MatZeroEntries + MatSetValues + MatAssemblyEnd re-uploads values only
        A.set_global_values(g["i"], g["j"], a2)
        A.mult_transpose(x, y)
        out.append(y.array())
        for o in (x, y, y0, z, A):
            o.destroy()
        return out

    res = run_ranks(size, body)
    parts = [gather([r[k] for r in res]) for k in range(4)]
    ref = sp.csr_matrix((g["a"], g["j"], g["i"]), shape=(N, N)).T @ xg
    scale = np.abs(ref).max() + 1e-300
    op = orc.operator(g["i"], g["j"], g["a"], size)
    assert np.abs(parts[0] - ref).max() <= 1e-12 * scale
    assert np.abs(parts[0] - op.mult_transpose(xg)).max() <= 1e-12 * scale
    zo = op.mult_transpose(xg, y0g)
    assert np.abs(parts[1] - zo).max() <= 1e-12 * (scale + 1) and np.abs(parts[2] - zo).max() <= 1e-12 * (scale + 1)
    ref2 = sp.csr_matrix((a2, g["j"], g["i"]), shape=(N, N)).T @ xg
    assert np.abs(parts[3] - ref2).max() <= 1e-12 * (np.abs(ref2).max() + 1e-300)


@pytest.mark.parametrize("size", [1, 2, 3])
def test_scale_diagonalscale_shift(golden_dir, size):
    """MatScale (aij.c:2475, mpiaij.c:1139), MatDiagonalScale (aij.c:2200, mpiaij.c:2040: the right vector's ghosts travel through
    MatMult's scatter), MatShift (gcreate.c:6-24) keep the nonzero pattern, so only values are re-sent to the device -- into the CSR,
    SELL-dict and transposed copies alike.  Checked through MatMult and MatMultTranspose against scipy."""
    import scipy.sparse as sp
    from petsc_amd import generators as G, petsc as P
    rng = np.random.default_rng(23)
    for ai, aj, aa in ((lambda g: (g["i"], g["j"], g["a"]))(load(golden_dir, "medium")), G.convdiff3d_7pt(9, 8, 7)):
        N = len(ai) - 1
        xg, lg, rg_ = rng.standard_normal(N), rng.uniform(0.5, 2.0, N), rng.uniform(0.5, 2.0, N)
        S = sp.csr_matrix((aa, aj, ai), shape=(N, N))
        R1 = S * 2.5
        R2 = sp.diags(lg) @ R1 @ sp.diags(rg_)
        R3 = R2 + 0.75 * sp.identity(N)

        def body(comm):
            rs, re = local_slice(comm, N)
            A = P.Mat.from_global_csr(comm, ai, aj, aa)
            x = P.Vec.from_array(comm, xg[rs:re], N)
            l = P.Vec.from_array(comm, lg[rs:re], N)
            r = P.Vec.from_array(comm, rg_[rs:re], N)
            y = x.duplicate()
            out = []
            A.mult_transpose(x, y)  # builds the transposed copies before the values change
            A.scale(2.5); A.mult(x, y); out.append(y.array())
            A.diagonal_scale(l, r); A.mult(x, y); out.append(y.array())
            A.mult_transpose(x, y); out.append(y.array())
            A.shift(0.75); A.mult(x, y); out.append(y.array())
            A.mult_transpose(x, y); out.append(y.array())
            A.scale(1.0); A.diagonal_scale(None, None)
            A.diagonal_scale(l, None); A.diagonal_scale(None, r); A.mult(x, y); out.append(y.array())
            for o in (x, y, l, r, A):
                o.destroy()
            return out

        res = run_ranks(size, body)
        got = [gather([r[k] for r in res]) for k in range(6)]
        R4 = sp.diags(lg) @ R3 @ sp.diags(rg_)
        for g_, ref in zip(got, (R1 @ xg, R2 @ xg, R2.T @ xg, R3 @ xg, R3.T @ xg, R4 @ xg)):
            assert np.abs(g_ - ref).max() <= 1e-12 * (np.abs(ref).max() + 1e-300)


@pytest.mark.parametrize("size", [1, 2])
def test_value_updates_stay_on_device(size):
    """MatScale / MatShift / MatDiagonalScale / MatZeroEntries of the HIP types run on the device copy (include/petschip.h "value updates
    that keep the nonzero pattern"); the host CSR catches up when somebody asks for values: MatHipGetBlockArrays (= what MatGetRow /
    MatView would read) returns the host loops' bits, a later MatSetValues + MatAssemblyEnd lands on the updated values, and the
    constant-coefficient layouts (pair dictionary, diagonal masks) survive scale and shift."""
    import ctypes as C
    from petsc_amd import capi, generators as G, petsc as P
    ai, aj, aa = G.laplace3d_7pt(10, 9, 8, dirichlet_rows=False)
    N = len(ai) - 1
    L = capi.load()

    def body(comm):
        rs, re = local_slice(comm, N)
        A = P.Mat.from_global_csr(comm, ai, aj, aa)
        before = [A.block(w)["a"].copy() for w in range(2 if comm.size > 1 else 1)]
        du = C.c_int()
        L.phip_mat_diag_mask_info(A.device_mat(0), None, C.byref(du), None)
        assert du.value == 1
        A.scale(3.0)
        A.shift(0.5)
        L.phip_mat_diag_mask_info(A.device_mat(0), None, C.byref(du), None)
        assert du.value == 1                                    # still the diagonal masks: nothing was rebuilt, nothing re-sent
        blk = A.block(0)
        want = before[0] * 3.0
        dpos = blk["diag"]
        want[dpos] = want[dpos] + 0.5
        np.testing.assert_array_equal(blk["a"], want)           # the host copy caught up (phip_mat_get_values)
        if comm.size > 1:
            np.testing.assert_array_equal(A.block(1)["a"], before[1] * 3.0)
        x = P.Vec.from_array(comm, np.ones(re - rs), N)
        y = x.duplicate()
        A.mult(x, y)
        y1 = y.array()
        A.set_values([rs], [rs], [2.0], P.ADD_VALUES)           # host-side change on top of the device-side ones
        A.assemble()
        A.mult(x, y)
        y2 = y.array()
        assert y2[0] == y1[0] + 2.0 and np.array_equal(y2[1:], y1[1:])
        A.zero_entries()
        A.mult(x, y)
        assert not y.array().any()
        for o in (x, y, A):
            o.destroy()
        return y1

    import scipy.sparse as sp
    S = sp.csr_matrix((aa, aj, ai), shape=(N, N)) * 3.0 + 0.5 * sp.identity(N)
    got = gather(run_ranks(size, body))
    assert np.abs(got - S @ np.ones(N)).max() <= 1e-12 * np.abs(S @ np.ones(N)).max()


@pytest.mark.parametrize("size", [2, 3])
def test_stash_after_device_side_scale(orc, size):
    """Off-process MatSetValues after a device-side MatScale (mpiaij.c:619-629 stash, :719-737 drain): the rank that owns the rows only
    RECEIVES entries; its host values, behind since the scale ran on the device, must catch up before the entries are added -- and the
    scale must survive the upload that follows."""
    import scipy.sparse as sp
    from petsc_amd import generators as G, petsc as P
    ai, aj, aa = G.convdiff3d_7pt(7, 6, 5)
    N = len(ai) - 1
    S = sp.csr_matrix((aa, aj, ai), shape=(N, N)) * 1.75
    S = S.tolil()
    rows = [0, 1, 3]
    for r in rows:
        S[r, r] += 0.5          # from the last rank
        S[r, aj[ai[r] + 1]] += 0.25
    S = S.tocsr()
    xg = np.linspace(-1.0, 2.0, N)
    ref = orc.matmult(S.indptr.astype(np.int32), S.indices.astype(np.int32), S.data, xg)

    def body(comm):
        rs, re = local_slice(comm, N)
        A = P.Mat.from_global_csr(comm, ai, aj, aa)
        x = P.Vec.from_array(comm, xg[rs:re], N)
        y = x.duplicate()
        A.mult(x, y)
        A.scale(1.75)                               # on the device: every rank's host CSR is now behind
        if comm.rank == comm.size - 1:              # rank 0 sets nothing itself
            for r in rows:
                A.set_values([r], [r, int(aj[ai[r] + 1])], [0.5, 0.25], P.ADD_VALUES)
        A.assemble()
        A.mult(x, y)
        out = y.array()
        for o in (x, y, A):
            o.destroy()
        return out

    got = gather(run_ranks(size, body))
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_ranks_without_rows(orc):
    """More ranks than rows (PetscSplitOwnership gives some ranks nothing, psplit.c:88), a diagonal matrix (no ghost entries anywhere)
    and a matrix whose ranks all talk to all others: products, reductions and a solve still agree with the oracle."""
    from petsc_amd import petsc as P
    rng = np.random.default_rng(31)
    cases = []
    n = 3
    cases.append((np.array([0, 2, 5, 7], np.int32), np.array([0, 1, 0, 1, 2, 1, 2], np.int32), np.array([4.0, -1, -1, 4, -1, -1, 4])))
    n2 = 37
    cases.append((np.arange(n2 + 1, dtype=np.int32), np.arange(n2, dtype=np.int32), rng.uniform(1.0, 2.0, n2)))
    import scipy.sparse as sp
    D = sp.random(29, 29, density=0.6, random_state=5, format="csr") + sp.identity(29) * 30.0
    D = (D + D.T).tocsr(); D.sort_indices()
    cases.append((D.indptr.astype(np.int32), D.indices.astype(np.int32), D.data.copy()))
    for ai, aj, aa in cases:
        N = len(ai) - 1
        xg = rng.standard_normal(N)
        bg = orc.operator(ai, aj, aa, 1).mult(np.ones(N))

        def body(comm):
            rs, re = local_slice(comm, N)
            A = P.Mat.from_global_csr(comm, ai, aj, aa)
            x = P.Vec.from_array(comm, xg[rs:re], N)
            y = x.duplicate()
            A.mult(x, y)
            out = [y.array(), x.dot(y), x.norm(), x.norm(P.NORM_INFINITY)]
            A.mult_transpose(x, y)
            out.append(y.array())
            out.append(ksp_case(comm, ai, aj, aa, bg, "cg", rtol=1e-10))
            for o in (x, y, A):
                o.destroy()
            return out

        for size in (2, 4, 5):
            res = run_ranks(size, body)
            op = orc.operator(ai, aj, aa, size)
            yref = op.mult(xg)
            assert np.abs(gather([r[0] for r in res]) - yref).max() <= 1e-12 * np.abs(yref).max()
            assert abs(res[0][1] - xg @ yref) <= 1e-12 * np.abs(xg * yref).sum()
            assert abs(res[0][2] - np.linalg.norm(xg)) <= 1e-13 * np.linalg.norm(xg) and res[0][3] == np.abs(xg).max()
            assert np.abs(gather([r[4] for r in res]) - op.mult_transpose(xg)).max() <= 1e-12 * np.abs(yref).max()
            check_ksp([r[5] for r in res], op.solve(bg, "cg", "jacobi", rtol=1e-10))


def ksp_case(comm, ai, aj, aa, b_global, ksp_type, pc="jacobi", x0=None, **opts):
    from petsc_amd import petsc as P
    N = len(ai) - 1
    rs, re = local_slice(comm, N)
    A = P.Mat.from_global_csr(comm, ai, aj, aa)
    b = P.Vec.from_array(comm, b_global[rs:re], N)
    x = b.duplicate()
    ksp = P.KSP(comm, ksp_type, pc)
    ksp.set_operators(A)
    ksp.set_tolerances(rtol=opts.get("rtol", P.PETSC_DEFAULT), max_it=opts.get("max_it", P.PETSC_DEFAULT))
    if "normtype" in opts:
        ksp.set_norm_type(opts["normtype"])
    if "restart" in opts:
        ksp.gmres_set_restart(opts["restart"])
    if "cgs_refine" in opts:
        ksp.gmres_set_cgs_refinement(opts["cgs_refine"])
    if opts.get("single_reduction"):
        ksp.cg_set_single_reduction(True)
    if opts.get("useabs"):
        ksp.pc_jacobi_set_use_abs(True)
    if x0 is not None:
        x.set_array(x0[rs:re])
        ksp.set_initial_guess_nonzero(True)
    ksp.set_residual_history()
    res = ksp.solve(b, x)
    res["x"] = x.array()
    for o in (ksp, x, b, A):
        o.destroy()
    return res


def check_ksp(res_ranks, ref):
    r0 = res_ranks[0]
    x = gather([r["x"] for r in res_ranks])
    assert r0["reason"] == ref["reason"], (r0["reason"], ref["reason"])
    assert abs(r0["its"] - ref["its"]) <= 1, (r0["its"], ref["its"])
    rn0 = ref["history"][0] if len(ref["history"]) else 1.0
    if r0["its"] == ref["its"]:
        assert abs(r0["rnorm"] - ref["rnorm"]) <= 1e-10 * rn0, (r0["rnorm"], ref["rnorm"])
        k = min(len(r0.get("history", [])), len(ref["history"]))
        assert k == 0 or np.abs(r0["history"][:k] - ref["history"][:k]).max() <= 1e-9 * rn0
    assert np.abs(x - ref["x"]).max() <= 1e-6 * (np.abs(ref["x"]).max() + 1e-300)


@pytest.mark.parametrize("size", [1, 2, 4])
def test_cg_jacobi_parity(orc, golden_dir, size):
    from petsc_amd import generators as G
    g = load(golden_dir, "spd-real-int32-float64")
    cases = [((g["i"], g["j"], g["a"]), g["vec0"], dict(rtol=1e-12))]
    for gen in (G.laplace3d_7pt(14, dirichlet_rows=False), G.box3d_27pt(11), G.laplace2d_5pt(40, 37)):
        n = len(gen[0]) - 1
        op = orc.operator(*gen, 1)
        cases.append((gen, op.mult(np.ones(n)), dict(rtol=1e-8)))
    for (ai, aj, aa), b, opts in cases:
        ref = orc.operator(ai, aj, aa, size).solve(b, "cg", "jacobi", **opts)
        check_ksp(run_ranks(size, lambda comm: ksp_case(comm, ai, aj, aa, b, "cg", **opts)), ref)


@pytest.mark.parametrize("size", [2, 3])
def test_reductions_device_route(orc, size):
    """The ranks' partial results can meet on the host (default: shared-memory ring, phip_comm_allreduce_host) or on the device
    (ncclAllReduce; PetscHipSetHostReductions(comm, PETSC_FALSE)).  Both routes: same CG / GMRES histories, same reductions."""
    from petsc_amd import generators as G, petsc as P
    ai, aj, aa = G.convdiff3d_7pt(11, 9, 8)
    n = len(ai) - 1
    b = orc.operator(ai, aj, aa, 1).mult(np.ones(n))
    sym = G.laplace3d_7pt(12, dirichlet_rows=False)
    bs = orc.operator(*sym, 1).mult(np.ones(len(sym[0]) - 1))

    def body(comm, host):
        comm.set_host_reductions(host)
        out = [ksp_case(comm, ai, aj, aa, b, "gmres", rtol=1e-9), ksp_case(comm, *sym, bs, "cg", rtol=1e-9)]
        rs, re = local_slice(comm, n)
        x = P.Vec.from_array(comm, np.linspace(-1.0, 2.0, n)[rs:re], n)
        y = P.Vec.from_array(comm, np.cos(np.arange(n))[rs:re], n)
        red = [x.dot(y), x.norm(), x.norm(P.NORM_1), x.norm(P.NORM_INFINITY), *x.norm(P.NORM_1_AND_2), *x.mdot([y, x, y])]
        x.destroy(); y.destroy()
        return out, red

    host = run_ranks(size, lambda comm: body(comm, True))
    dev = run_ranks(size, lambda comm: body(comm, False))
    ref_g = orc.operator(ai, aj, aa, size).solve(b, "gmres", "jacobi", rtol=1e-9)
    ref_c = orc.operator(*sym, size).solve(bs, "cg", "jacobi", rtol=1e-9)
    for res in (host, dev):
        check_ksp([r[0][0] for r in res], ref_g)
        check_ksp([r[0][1] for r in res], ref_c)
        for r in res[1:]:
            assert r[1] == res[0][1]  # every rank holds the same bits
    np.testing.assert_allclose(host[0][1], dev[0][1], rtol=1e-13)


@pytest.mark.parametrize("size", [1, 2, 3])
def test_split_reductions_and_pipecg(orc, size):
    """VecDotBegin/End, VecNormBegin/End (src/vec/vec/utils/comb.c) return what the blocking calls return, in Begin order, with other
    work queued in between; KSPPIPECG (pipecg.c:31-169) follows the oracle's restatement: +-1 iteration, 1e-10 relative residual."""
    from petsc_amd import generators as G, petsc as P
    sym = G.laplace3d_7pt(13, dirichlet_rows=False)
    n = len(sym[0]) - 1
    bs = orc.operator(*sym, 1).mult(np.ones(n))
    xg, yg = np.linspace(-1.0, 2.0, n), np.cos(np.arange(n))

    def body(comm):
        rs, re = local_slice(comm, n)
        x = P.Vec.from_array(comm, xg[rs:re], n)
        y = P.Vec.from_array(comm, yg[rs:re], n)
        w = x.duplicate()
        blocking = [x.dot(y), x.norm(), y.norm(P.NORM_1), y.norm(P.NORM_INFINITY), x.norm(P.NORM_1_AND_2), y.dot(y)]
        x.dot_begin(y); x.norm_begin(); y.norm_begin(P.NORM_1); y.norm_begin(P.NORM_INFINITY); x.norm_begin(P.NORM_1_AND_2); y.dot_begin(y)
        w.waxpy(2.0, x, y)   # work queued behind the reduction kernels
        split = [x.dot_end(y), x.norm_end(), y.norm_end(P.NORM_1), y.norm_end(P.NORM_INFINITY), x.norm_end(P.NORM_1_AND_2), y.dot_end(y)]
        # a second round reuses the queue; ending in another order is an error (comb.c)
        x.dot_begin(y); y.norm_begin()
        with pytest.raises(P.PetscError):
            y.norm_end()
        assert x.dot_end(y) == blocking[0] and y.norm_end() == np.sqrt(blocking[5])
        out = [blocking, split, w.array()]
        for nt in (P.KSP_NORM_PRECONDITIONED, P.KSP_NORM_UNPRECONDITIONED, P.KSP_NORM_NATURAL):
            out.append(ksp_case(comm, *sym, bs, "pipecg", rtol=1e-9, normtype=nt))
        for v in (x, y, w):
            v.destroy()
        return out

    res = run_ranks(size, body)
    assert res[0][0] == res[0][1]                      # the same kernels, the same order of sums: the same bits
    np.testing.assert_array_equal(gather([r[2] for r in res]), orc.waxpy(2.0, xg, yg))
    for k, nt in enumerate((1, 2, 3)):
        ref = orc.operator(*sym, size).solve(bs, "pipecg", "jacobi", rtol=1e-9, normtype=nt)
        check_ksp([r[3 + k] for r in res], ref)


def test_cg_variants(orc):
    """norm types (cg.c:136-160), -ksp_cg_single_reduction (cg.c:226-243), non-zero initial guess (cg.c:131-132)."""
    from petsc_amd import generators as G, petsc as P
    ai, aj, aa = G.laplace3d_7pt(12, dirichlet_rows=False)
    n = len(ai) - 1
    op = orc.operator(ai, aj, aa, 1)
    b = op.mult(np.linspace(-1, 1, n))
    for nt in (P.KSP_NORM_PRECONDITIONED, P.KSP_NORM_UNPRECONDITIONED, P.KSP_NORM_NATURAL):
        ref = op.solve(b, "cg", "jacobi", rtol=1e-9, normtype=nt)
        check_ksp(run_ranks(1, lambda comm: ksp_case(comm, ai, aj, aa, b, "cg", rtol=1e-9, normtype=nt)), ref)
    x0 = np.cos(np.arange(n))
    ref = op.solve(b, "cg", "jacobi", rtol=1e-9, x0=x0)
    check_ksp(run_ranks(2, lambda comm: ksp_case(comm, ai, aj, aa, b, "cg", rtol=1e-9, x0=x0)), ref)
    ref = op.solve(b, "cg", "jacobi", rtol=1e-9)
    res = run_ranks(1, lambda comm: ksp_case(comm, ai, aj, aa, b, "cg", rtol=1e-9, single_reduction=True))
    assert abs(res[0]["its"] - ref["its"]) <= 1 and res[0]["reason"] == 2
    assert np.abs(res[0]["x"] - ref["x"]).max() <= 1e-6 * np.abs(ref["x"]).max()
    ref = op.solve(b, "cg", "none", rtol=1e-9)
    check_ksp(run_ranks(1, lambda comm: ksp_case(comm, ai, aj, aa, b, "cg", pc="none", rtol=1e-9)), ref)


@pytest.mark.parametrize("size", [1, 3])
def test_gmres_jacobi_parity(orc, golden_dir, size):
    from petsc_amd import generators as G
    ai, aj, aa = G.convdiff3d_7pt(13, 11, 9)
    n = len(ai) - 1
    b = orc.operator(ai, aj, aa, 1).mult(np.ones(n))
    for refine in (REFINE_NEVER, REFINE_IFNEEDED, REFINE_ALWAYS):
        for restart in (30, 7):
            ref = orc.operator(ai, aj, aa, size).solve(b, "gmres", "jacobi", rtol=1e-9, restart=restart, cgs_refine=refine)
            check_ksp(run_ranks(size, lambda comm: ksp_case(comm, ai, aj, aa, b, "gmres", rtol=1e-9, restart=restart, cgs_refine=refine)), ref)
    g = load(golden_dir, "ns-real-int32-float64")
    ref = orc.operator(g["i"], g["j"], g["a"], size).solve(g["vec0"], "gmres", "jacobi", rtol=1e-10)
    check_ksp(run_ranks(size, lambda comm: ksp_case(comm, g["i"], g["j"], g["a"], g["vec0"], "gmres", rtol=1e-10)), ref)


EX3_1_OUT = ["0.541987", "0.327572", "0.201868", "0.146074", "0.102129", "0.0549947", "0.0216871", "< 1.e-11"]
EX5_1_OUT = [["4.16083", "1.32287", "0.625076", "0.211511", "0.0201553", "< 1.e-11"],
             ["5.16667", "1.01143", "0.241047", "0.0452703", "0.00290094", "< 1.e-11"]]


@pytest.mark.parametrize("size", [1, 2])
def test_ex3_golden_history(size):
    """src/ksp/ksp/examples/tests/ex3.c end to end through the mirrored API: element-by-element ADD_VALUES assembly (rank r
    inserts elements r, r+size, ... so off-process rows go through the stash), MatZeroRows on the boundary, GMRES + Jacobi
    with -ksp_gmres_cgs_refinement_type refine_always; the residual history must print as tests/output/ex3_1.out."""
    from petsc_amd import petsc as P
    m = 5
    N = (m + 1) ** 2
    h = 1.0 / m
    H = h * h
    Ke = np.array([H / 6, -.125 * H, H / 12, -.125 * H, -.125 * H, H / 6, -.125 * H, H / 12,
                   H / 12, -.125 * H, H / 6, -.125 * H, -.125 * H, H / 12, -.125 * H, H / 6])

    def body(comm):
        A = P.Mat(comm, P.PETSC_DECIDE, P.PETSC_DECIDE, N, N, "aijhip")
        A.set_up()
        for e in range(comm.rank, m * m, comm.size):
            i0 = (m + 1) * (e // m) + (e % m)
            idx = [i0, i0 + 1, i0 + m + 2, i0 + m + 1]
            A.set_values(idx, idx, Ke, P.ADD_VALUES)
        A.assemble()
        rs, re = A.ownership_range()
        u, b = A.create_vecs()
        brows = [r for r in range(rs, re) if r // (m + 1) in (0, m) or r % (m + 1) in (0, m)]
        vals = [h * (r // (m + 1)) for r in brows]
        u.set_values(brows, vals); u.assemble()
        b.set_values(brows, vals); b.assemble()
        A.zero_rows(brows, 1.0)
        ksp = P.KSP(comm, "gmres", "jacobi")
        ksp.set_operators(A)
        ksp.set_initial_guess_nonzero(True)
        ksp.gmres_set_cgs_refinement(P.REFINE_ALWAYS)
        ksp.set_residual_history()
        res = ksp.solve(b, u)
        res["x"] = u.array()
        for o in (ksp, u, b, A):
            o.destroy()
        return res

    res = run_ranks(size, body)
    assert [monitor_short(v) for v in res[0]["history"]] == EX3_1_OUT
    assert res[0]["its"] == 7 and res[0]["reason"] == 2
    ustar = np.array([h * (i // (m + 1)) for i in range(N)])
    assert np.linalg.norm(gather([r["x"] for r in res]) - ustar) < 1e-13


def test_ex5_golden_history():
    """src/ksp/ksp/examples/tutorials/ex5.c (two solves, the second after re-assembling new values into the same pattern)."""
    from petsc_amd import petsc as P
    m, n = 3, 2

    def body(comm):
        A = P.Mat(comm, m * n, m * n, m * n, m * n, "aijhip")
        A.seqaij_set_preallocation(5)
        ksp = P.KSP(comm, "gmres", "jacobi")
        ksp.gmres_set_cgs_refinement(P.REFINE_ALWAYS)
        ksp.set_residual_history()
        hists = []
        for dv in (4.0, 6.0):
            if hists:
                A.zero_entries()
            for Ii in range(m * n):
                i, j = Ii // n, Ii % n
                if i > 0: A.set_values([Ii], [Ii - n], [-1.0], P.ADD_VALUES)
                if i < m - 1: A.set_values([Ii], [Ii + n], [-1.0], P.ADD_VALUES)
                if j > 0: A.set_values([Ii], [Ii - 1], [-1.0], P.ADD_VALUES)
                if j < n - 1: A.set_values([Ii], [Ii + 1], [-1.0], P.ADD_VALUES)
                A.set_values([Ii], [Ii], [dv], P.ADD_VALUES)
            A.assemble()
            u, b = A.create_vecs()
            u.set_array(np.arange(m * n, dtype=float))
            A.mult(u, b)
            x = u.duplicate()
            ksp.set_operators(A)
            res = ksp.solve(b, x)
            assert res["its"] == 5
            hists.append([monitor_short(v) for v in res["history"]])
            for o in (u, b, x):
                o.destroy()
        ksp.destroy(); A.destroy()
        return hists

    assert run_ranks(1, body)[0] == EX5_1_OUT


@pytest.mark.parametrize("size", [1, 2])
def test_ex10_file_flow(orc, golden_dir, tmp_path, size):
    """src/ksp/ksp/examples/tutorials/ex10.c:103-147 shape: MatLoad, VecLoad of the right-hand side from the same file, KSPSolve;
    then VecView/VecLoad round trip."""
    from petsc_amd import petsc as P
    g = load(golden_dir, "spd-real-int32-float64")
    src = os.path.join(golden_dir, "petscbin_spd-real-int32-float64.dat")
    out = str(tmp_path / ("sol%d.dat" % size))

    def body(comm):
        A = P.Mat(comm)
        off = A.load(src)
        N = A.sizes()[2]
        b = P.Vec(comm, P.PETSC_DECIDE, N, "hip")
        b.load(src, off)
        rs, re = b.ownership_range()
        np.testing.assert_array_equal(b.array(), g["vec0"][rs:re])
        x = b.duplicate()
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-12)
        res = ksp.solve(b, x)
        x.view(out)
        x2 = b.duplicate()
        x2.load(out)
        np.testing.assert_array_equal(x2.array(), x.array())
        res["x"] = x.array()
        for o in (ksp, x, x2, b, A):
            o.destroy()
        return res

    ref = orc.operator(g["i"], g["j"], g["a"], size).solve(g["vec0"], "cg", "jacobi", rtol=1e-12)
    check_ksp(run_ranks(size, body), ref)


def test_jacobi_variants(orc):
    """PCJacobi rowmax / rowsum / abs / zero diagonal -> 1.0 (jacobi.c:174-195)."""
    from petsc_amd import petsc as P
    ai = np.array([0, 2, 4, 5, 7], np.int32)
    aj = np.array([0, 1, 0, 2, 2, 1, 3], np.int32)
    aa = np.array([2.0, -1.0, 3.0, 4.0, -5.0, 1.0, 0.0])

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, 4, 4, ai, aj, aa)
        out = {}
        for nm, jt, ua in (("diag", P.PC_JACOBI_DIAGONAL, False), ("abs", P.PC_JACOBI_DIAGONAL, True), ("rowmax", P.PC_JACOBI_ROWMAX, False), ("rowsum", P.PC_JACOBI_ROWSUM, False)):
            ksp = P.KSP(comm, "cg", "jacobi")
            ksp.set_operators(A)
            ksp.pc_jacobi_set_type(jt)
            ksp.pc_jacobi_set_use_abs(ua)
            x = P.Vec.from_array(comm, np.ones(4))
            y = x.duplicate()
            P.chk(P.lib().PCApply(ksp.pc, x.h, y.h))
            out[nm] = y.array()
            for o in (ksp, x, y):
                o.destroy()
        A.destroy()
        return out

    r = run_ranks(1, body)[0]
    np.testing.assert_array_equal(r["diag"], [0.5, 1.0, -0.2, 1.0])  # row 1 has no diagonal entry, row 3 an explicit zero
    np.testing.assert_array_equal(r["abs"], [0.5, 1.0, 0.2, 1.0])
    np.testing.assert_array_equal(r["rowmax"], [0.5, 0.25, 0.2, 1.0])
    np.testing.assert_array_equal(r["rowsum"], [1.0, 1.0 / 7.0, -0.2, 1.0])


def test_full_size_properties(orc):
    """BASELINE config 2 at full size (128^3, 7-point): linearity and symmetry of the SpMV, agreement with the oracle on
    a strided sample of rows (the oracle computes only those rows), and 40 KSPCG + PCJACOBI iterations against the oracle."""
    from petsc_amd import generators as G, petsc as P
    ai, aj, aa = G.laplace3d_7pt(128, dirichlet_rows=False)
    n = len(ai) - 1

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        rng = np.random.default_rng(1)
        xa, xb = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        x, y = A.create_vecs()
        outs = []
        for v in (xa, xb, 2.0 * xa - 3.0 * xb):
            x.set_array(v)
            A.mult(x, y)
            outs.append(y.array())
        ya, yb, yc = outs
        assert np.abs(yc - (2.0 * ya - 3.0 * yb)).max() <= 1e-12 * np.abs(yc).max()
        assert abs(xb @ ya - xa @ yb) <= 1e-12 * np.abs(xb * ya).sum()  # symmetric matrix
        rows = np.arange(0, n, 4099)
        ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], xa[aj[ai[r]:ai[r + 1]]]) for r in rows])
        assert np.abs(ya[rows] - ref).max() <= 1e-12 * np.abs(ref).max()
        d = x.duplicate()
        A.get_diagonal(d)
        np.testing.assert_array_equal(d.array(), np.full(n, 6.0))
        bv = P.Vec.from_array(comm, xb, n)
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-30, max_it=40)
        ksp.set_residual_history()
        x.set(0.0)
        res = ksp.solve(bv, x)
        res["x"] = x.array()
        for o in (ksp, bv, x, y, d, A):
            o.destroy()
        return res, xb

    res, b = run_ranks(1, body)[0]
    # 40 KSPCG + PCJACOBI iterations of the same system by the oracle
    ref = orc.operator(ai, aj, aa, 1).solve(b, "cg", "jacobi", rtol=1e-30, max_it=40)
    assert (res["its"], res["reason"]) == (ref["its"], ref["reason"]) and res["its"] == 40
    rn0 = ref["history"][0]
    k = min(len(res["history"]), len(ref["history"]))
    assert k >= 40 and np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-9 * rn0
    assert abs(res["rnorm"] - ref["rnorm"]) <= 1e-10 * rn0
    assert np.abs(res["x"] - ref["x"]).max() <= 1e-9 * np.abs(ref["x"]).max()


def test_full_size_config3(orc):
    """BASELINE config 3 at full size -- the bench's north-star workload: 27-point box stencil on 256^3 (16.8 M rows, 449 M nonzeros)
    through the default layout (diagonal masks, group descriptors).  Size-independent properties of the product (linearity, symmetry,
    a strided sample of rows against numpy), MatGetDiagonal, and 12 iterations of KSPCG + PCJACOBI against the oracle on the same
    system: same iteration count and reason, residual history within 1e-9, final residual within 1e-10 (the north star's bar)."""
    from petsc_amd import generators as G, petsc as P
    nn = 256
    n = nn ** 3
    ai, aj, aa = G.stencil_rows_native(27, nn, nn, nn, 0, n)
    assert len(aj) == 449455096
    rng = np.random.default_rng(3)
    xa, xb = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    b = rng.uniform(-1, 1, n)
    its = 12

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        x, y = A.create_vecs()
        outs = []
        for v in (xa, xb, 2.0 * xa - 3.0 * xb):
            x.set_array(v)
            A.mult(x, y)
            outs.append(y.array())
        ya, yb, yc = outs
        assert np.abs(yc - (2.0 * ya - 3.0 * yb)).max() <= 1e-12 * np.abs(yc).max()
        assert abs(xb @ ya - xa @ yb) <= 1e-12 * np.abs(xb * ya).sum()  # symmetric matrix
        rows = np.concatenate([np.arange(0, n, 40009), [n - 1, n - nn, nn * nn - 1, nn * nn]])
        ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], xa[aj[ai[r]:ai[r + 1]]]) for r in rows])
        assert np.abs(ya[rows] - ref).max() <= 1e-12 * np.abs(ref).max()
        d = x.duplicate()
        A.get_diagonal(d)
        dref = np.array([aa[ai[r]:ai[r + 1]][aj[ai[r]:ai[r + 1]] == r][0] for r in rows])
        np.testing.assert_array_equal(d.array()[rows], dref)
        bv = P.Vec.from_array(comm, b, n)
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-30, max_it=its)
        ksp.set_residual_history()
        x.set(0.0)
        res = ksp.solve(bv, x)
        res["x"] = x.array()
        for o in (ksp, x, y, d, bv, A):
            o.destroy()
        return res

    res = run_ranks(1, body)[0]
    ref = orc.operator(ai, aj, aa, 1).solve(b, "cg", "jacobi", rtol=1e-30, max_it=its)
    assert (res["its"], res["reason"]) == (ref["its"], ref["reason"]) and res["its"] == its
    rn0 = ref["history"][0]
    k = min(len(res["history"]), len(ref["history"]))
    assert k >= its and np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-9 * rn0
    assert abs(res["rnorm"] - ref["rnorm"]) <= 1e-10 * rn0
    assert np.abs(res["x"] - ref["x"]).max() <= 1e-9 * np.abs(ref["x"]).max()


def test_full_size_config3_to_tolerance():
    """BASELINE config 3 at full size, solved TO TOLERANCE ("a to-tolerance run (rtol 1e-8) for iteration-count parity"):
    27-point 256^3, b = A 1, KSPCG + PCJACOBI, KSPConvergedDefault (iterativ.c:744-817) with rtol 1e-8 and the defaults of
    itcreate.c:706-714 otherwise.  The oracle runs the same system as MPIAIJ would on 16 row blocks, one thread each (its -O3 build; the
    result does not depend on the number of threads, tests/test_oracle.py).  The north star's bar: same reason, iteration count within
    1, residual norms within 1e-10 (relative to the initial one) at every iteration both solves reach, and the same solution."""
    from oracle.oracle import Oracle
    from petsc_amd import generators as G, petsc as P
    nn = 256
    n = nn ** 3
    ai, aj, aa = G.stencil_rows_native(27, nn, nn, nn, 0, n)
    op = Oracle(fast=True).operator(ai, aj, aa, 16, 16)
    b = op.mult(np.ones(n))
    ref = op.solve(b, "cg", "jacobi", rtol=1e-8)

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        x, y = A.create_vecs()
        x.set(1.0)
        A.mult(x, y)
        np.testing.assert_array_equal(y.array(), b)   # integer-valued entries, sums of at most 27 of them: exact in every order
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-8)
        ksp.set_residual_history()
        res = ksp.solve(y, x)
        res["x"] = x.array()
        for o in (ksp, x, y, A):
            o.destroy()
        return res

    res = run_ranks(1, body)[0]
    assert res["reason"] == ref["reason"] == 2, (res["reason"], ref["reason"])        # KSP_CONVERGED_RTOL
    assert abs(res["its"] - ref["its"]) <= 1 and res["its"] > 100, (res["its"], ref["its"])
    rn0 = ref["history"][0]
    k = min(len(res["history"]), len(ref["history"]))
    assert k >= min(res["its"], ref["its"]) and np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-10 * rn0
    assert res["history"][k - 1] <= 1.0001e-8 * rn0 or res["history"][-1] <= 1e-8 * rn0
    assert np.abs(res["x"] - 1.0).max() <= 1e-5 and np.abs(res["x"] - ref["x"]).max() <= 1e-9


def test_full_size_config1(orc):
    """BASELINE config 1 at full size: src/ksp/ksp/examples/tutorials/ex2.c with -m 1000 -n 1000 (2-D 5-point Laplacian, b = A 1,
    KSPCG + PCJACOBI to rtol 1e-2 / ((m+1)(n+1))) against the oracle: iteration count within 1, same reason, and -- when the counts agree --
    final residual within 1e-10 of the initial one; the error norm ex2 prints (|x - 1|) agrees to 1e-6."""
    from petsc_amd import generators as G, petsc as P
    mm = 1000
    ai, aj, aa = G.laplace2d_5pt(mm, mm)
    n = len(ai) - 1
    rtol = 1e-2 / ((mm + 1) * (mm + 1))
    bref = orc.operator(ai, aj, aa, 1).mult(np.ones(n))

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        x, b = A.create_vecs()
        x.set(1.0)
        A.mult(x, b)
        np.testing.assert_array_equal(b.array(), bref)
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=rtol)
        ksp.set_residual_history()
        x.set(0.0)
        res = ksp.solve(b, x)
        res["x"] = x.array()
        for o in (ksp, x, b, A):
            o.destroy()
        return res

    res = run_ranks(1, body)[0]
    ref = orc.operator(ai, aj, aa, 1).solve(bref, "cg", "jacobi", rtol=rtol)
    assert res["reason"] == ref["reason"] and abs(res["its"] - ref["its"]) <= 1 and ref["its"] > 1000
    rn0 = ref["history"][0]
    if res["its"] == ref["its"]:
        assert abs(res["rnorm"] - ref["rnorm"]) <= 1e-10 * rn0
    k = min(len(res["history"]), len(ref["history"]), 200)
    assert np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-9 * rn0
    assert abs(np.linalg.norm(res["x"] - 1.0) - np.linalg.norm(ref["x"] - 1.0)) <= 1e-6 * np.linalg.norm(ref["x"] - 1.0)


def test_full_size_config4(orc):
    """BASELINE config 4 at full size: 3-D convection-diffusion 192^3 (7.08 M rows, 49.3 M nonzeros, nonsymmetric), KSPGMRES(30) +
    PCJACOBI for 36 iterations (one restart) against the oracle on the same system: iteration count, reason, residual history within
    1e-9 and final residual within 1e-10; the product on a strided sample of rows against numpy, and the transposed product through
    (A^T x) . y = x . (A y)."""
    from petsc_amd import generators as G, petsc as P
    nn = 192
    ai, aj, aa = G.convdiff3d_7pt(nn)
    n = len(ai) - 1
    assert (n, len(aj)) == (7077888, 49324032)
    rng = np.random.default_rng(4)
    xa, xb, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    its = 36

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        x, y = A.create_vecs()
        x.set_array(xa)
        A.mult(x, y)
        ya = y.array()
        rows = np.concatenate([np.arange(0, n, 20011), [n - 1, nn * nn - 1, nn * nn]])
        ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], xa[aj[ai[r]:ai[r + 1]]]) for r in rows])
        assert np.abs(ya[rows] - ref).max() <= 1e-12 * np.abs(ref).max()
        x.set_array(xb)
        A.mult_transpose(x, y)
        assert abs(y.array() @ xa - xb @ ya) <= 1e-12 * np.abs(xb * ya).sum()
        bv = P.Vec.from_array(comm, b, n)
        ksp = P.KSP(comm, "gmres", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-30, max_it=its)
        ksp.gmres_set_restart(30)
        ksp.set_residual_history()
        x.set(0.0)
        res = ksp.solve(bv, x)
        res["x"] = x.array()
        for o in (ksp, x, y, bv, A):
            o.destroy()
        return res

    res = run_ranks(1, body)[0]
    ref = orc.operator(ai, aj, aa, 1).solve(b, "gmres", "jacobi", rtol=1e-30, max_it=its, restart=30)
    assert (res["its"], res["reason"]) == (ref["its"], ref["reason"]) and res["its"] == its
    rn0 = ref["history"][0]
    k = min(len(res["history"]), len(ref["history"]))
    assert k >= its and np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-9 * rn0
    assert abs(res["rnorm"] - ref["rnorm"]) <= 1e-10 * rn0
    assert np.abs(res["x"] - ref["x"]).max() <= 1e-9 * np.abs(ref["x"]).max()


EX5_2_OUT = [["142.26", "56.7664", "28.4652", "16.0205", "6.37762", "2.18849", "0.698897", "0.1896", "0.0311023", "0.00173053", "< 1.e-11"],
             ["171.949", "51.7756", "16.2674", "5.93715", "0.787126", "0.156071", "0.0363183", "0.005189", "0.000484311", "1.74022e-05"]]
EX2F_1_OUT = ["1.11803", "0.591608", "0.316228", "< 1.e-11"]


def test_ex5_2_golden_history():
    """src/ksp/ksp/examples/tutorials/ex5.c on 2 ranks with -ksp_rtol .000001 (makefile runex5_2; n = 2 * size, u = local index + 100 *
    rank): the second system is assembled by COLUMN blocks (ex5.c:232-241), so half of every rank's values travel through the stash;
    both residual histories must print as tutorials/output/ex5_2.out."""
    from petsc_amd import petsc as P
    m, size = 3, 2
    n = 2 * size

    def body(comm):
        rank = comm.rank
        A = P.Mat(comm, P.PETSC_DECIDE, P.PETSC_DECIDE, m * n, m * n, "aijhip")
        A.set_up()
        rs, re = A.ownership_range()

        def stencil(Ii, dv):
            i, j = Ii // n, Ii % n
            if i > 0: A.set_values([Ii], [Ii - n], [-1.0], P.ADD_VALUES)
            if i < m - 1: A.set_values([Ii], [Ii + n], [-1.0], P.ADD_VALUES)
            if j > 0: A.set_values([Ii], [Ii - 1], [-1.0], P.ADD_VALUES)
            if j < n - 1: A.set_values([Ii], [Ii + 1], [-1.0], P.ADD_VALUES)
            A.set_values([Ii], [Ii], [dv], P.ADD_VALUES)

        for Ii in range(rs, re):
            stencil(Ii, 4.0)
        A.assemble()
        u, b = A.create_vecs()
        u.set_array(np.arange(re - rs, dtype=float) + 100.0 * rank)
        x = u.duplicate()
        ksp = P.KSP(comm, "gmres", "jacobi")
        ksp.gmres_set_cgs_refinement(P.REFINE_ALWAYS)
        ksp.set_tolerances(rtol=1e-6)
        ksp.set_residual_history()
        hists, its = [], []
        for stage in range(2):
            if stage:
                A.zero_entries()
                for i in range(m):
                    for j in range(2 * rank, 2 * rank + 2):
                        stencil(j + n * i, 6.0)
                A.assemble()
            A.mult(u, b)
            ksp.set_operators(A)
            res = ksp.solve(b, x)
            hists.append([monitor_short(v) for v in res["history"]])
            its.append(res["its"])
        for o in (ksp, u, b, x, A):
            o.destroy()
        return hists, its

    hists, its = run_ranks(size, body)[0]
    assert hists == EX5_2_OUT and its == [10, 9]


@pytest.mark.parametrize("size", [1, 2])
def test_ex2f_golden_history(size):
    """src/ksp/ksp/examples/tutorials/ex2f.F (the 5-point Laplacian of ex2 on a 3 x 3 grid, u = 1, b = A u, GMRES + Jacobi with
    refine_always, default tolerances; makefile runex2f on 2 ranks): tutorials/output/ex2f_1.out, 3 iterations, error below 1e-12."""
    from petsc_amd import petsc as P
    m = n = 3

    def body(comm):
        A = P.Mat(comm, P.PETSC_DECIDE, P.PETSC_DECIDE, m * n, m * n, "aijhip")
        A.set_up()
        rs, re = A.ownership_range()
        for II in range(rs, re):
            i, j = II // n, II % n
            if i > 0: A.set_values([II], [II - n], [-1.0], P.INSERT_VALUES)
            if i < m - 1: A.set_values([II], [II + n], [-1.0], P.INSERT_VALUES)
            if j > 0: A.set_values([II], [II - 1], [-1.0], P.INSERT_VALUES)
            if j < n - 1: A.set_values([II], [II + 1], [-1.0], P.INSERT_VALUES)
            A.set_values([II], [II], [4.0], P.INSERT_VALUES)
        A.assemble()
        u, b = A.create_vecs()
        u.set(1.0)
        A.mult(u, b)
        x = u.duplicate()
        ksp = P.KSP(comm, "gmres", "jacobi")
        ksp.gmres_set_cgs_refinement(P.REFINE_ALWAYS)
        ksp.set_operators(A)
        ksp.set_residual_history()
        res = ksp.solve(b, x)
        res["x"] = x.array()
        for o in (ksp, u, b, x, A):
            o.destroy()
        return res

    res = run_ranks(size, body)
    assert [monitor_short(v) for v in res[0]["history"]] == EX2F_1_OUT and res[0]["its"] == 3
    assert np.linalg.norm(gather([r["x"] for r in res]) - 1.0) < 1e-12


@pytest.mark.parametrize("name", ["afiro", "golan"])
@pytest.mark.parametrize("pc", ["none", "jacobi"])
def test_ex25_indefinite_matrices(orc, golden_dir, name, pc):
    """src/ksp/ksp/examples/tests/ex25.c: KSPCG with rtol 1e-7 on the symmetric INDEFINITE matrices afiro and golan (right-hand side and
    exact solution come with the files).  ex25_1.out / ex25_2.out (158 / 56 iterations to convergence) predate the test of cg.c:215-219
    and cannot come out of this source: KSPSolve_CG stops with KSP_DIVERGED_INDEFINITE_MAT (or _PC, cg.c:187-190) as soon as p'Ap changes
    sign.  Error behaviour is part of the contract: same reason and iteration as the oracle, same residual at the stop."""
    from petsc_amd import petsc as P
    g = load(golden_dir, name)
    ai, aj, aa, b = g["i"], g["j"], g["a"], g["vec0"]
    n = len(ai) - 1

    def body(comm):
        A = P.Mat.seqaij_with_arrays(comm, n, n, ai, aj, aa)
        bv = P.Vec.from_array(comm, b, n)
        x = bv.duplicate()
        ksp = P.KSP(comm, "cg", pc)
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=1e-7)
        ksp.set_residual_history()
        x.set(0.0)
        res = ksp.solve(bv, x)
        for o in (ksp, x, bv, A):
            o.destroy()
        return res

    res = run_ranks(1, body)[0]
    ref = orc.operator(ai, aj, aa, 1).solve(b, "cg", pc, rtol=1e-7)
    assert ref["reason"] in (-8, -10)  # KSP_DIVERGED_INDEFINITE_PC / _MAT (petscksp.h)
    assert (res["reason"], res["its"]) == (ref["reason"], ref["its"])
    k = min(len(res["history"]), len(ref["history"]))
    assert np.abs(res["history"][:k] - ref["history"][:k]).max() <= 1e-9 * ref["history"][0]


@pytest.mark.parametrize("size", [1, 2, 3])
def test_byproduct_cache(orc, size):
    """VecTDot(P,W) after MatMult(A,P,W), VecNorm(Z) and VecTDot(Z,R) after VecPointwiseMult(Z,R,D) are served from by-products
    of the producing kernels; any later write to an operand invalidates them; results agree with the stand-alone kernels."""
    from petsc_amd import generators as G, petsc as P
    ai, aj, aa = G.laplace3d_7pt(11, 9, 13, dirichlet_rows=False)
    N = len(ai) - 1
    xg = G.random_vector(N)

    def body(comm):
        rs, re = local_slice(comm, N)
        A = P.Mat.from_global_csr(comm, ai, aj, aa)
        x = P.Vec.from_array(comm, xg[rs:re], N)
        y, z, d = x.duplicate(), x.duplicate(), x.duplicate()
        d.set_array(np.linspace(0.5, 1.5, N)[rs:re])
        out = {}
        for on in (True, False):
            comm.set_byproducts(on)
            A.mult(x, y)
            out["xy", on] = x.dot(y)
            out["yx", on] = y.tdot(x)
            z.pointwise_mult(y, d)
            out["nz", on] = z.norm()
            out["zy", on] = z.tdot(y)
            out["dz", on] = d.dot(z)
            # writes invalidate: after scaling y, dot(x, y) must be the new value
            y.scale(2.0)
            out["xy2", on] = x.dot(y)
            out["zy2", on] = z.dot(y)
            z.axpy(1.0, y)
            out["nz2", on] = z.norm()
            # VecMAXPY leaves sum z^2 behind (the VecNormalize after the GMRES orthogonalisation); a later write invalidates it
            z.maxpy([0.5, -0.25], [x, y])
            out["nz3", on] = z.norm()
            out["nz3n", on] = z.normalize()
            out["nz4", on] = z.norm()
        comm.set_byproducts(True)
        for o in (x, y, z, d, A):
            o.destroy()
        return out

    res = run_ranks(size, body)[0]
    yo = orc.operator(ai, aj, aa, 1).mult(xg)
    dg = np.linspace(0.5, 1.5, N)
    zo = yo * dg
    ref = {"xy": xg @ yo, "yx": xg @ yo, "nz": np.linalg.norm(zo), "zy": zo @ yo, "dz": dg @ zo, "xy2": 2 * (xg @ yo), "zy2": 2 * (zo @ yo),
           "nz2": np.linalg.norm(zo + 2 * yo), "nz3": np.linalg.norm(zo + 2 * yo + 0.5 * xg - 0.5 * yo),
           "nz3n": np.linalg.norm(zo + 2 * yo + 0.5 * xg - 0.5 * yo), "nz4": 1.0}
    for key, val in ref.items():
        for on in (True, False):
            assert abs(res[key, on] - val) <= 1e-12 * max(abs(val), 1.0), (key, on, res[key, on], val)


@pytest.mark.parametrize("n,ipc", [(2, 0), (3, 0), (2, 1), (3, 1)])
def test_multiprocess_shared_gpu(n, ipc):
    """One process per rank (as under torch.distributed.run), all on GPU 0, joined by the "shm" backend: products, transposed products,
    reductions through the cross-process shared-memory ring, GMRES and CG against the oracle.  ipc: the forward ghost exchange as
    peer writes (PHIP_HALO_IPC=1: hipIpc-mapped receive buffers, flag and acknowledgement words) instead of staging through the host."""
    import subprocess
    import sys
    key = os.urandom(128).hex()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PHIP_HOST_RING_TIMEOUT="60", PHIP_HALO_IPC=str(ipc))
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "_gpu_mp_worker.py"), key, str(n), str(r)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=100)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, _ = p.communicate()
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "mp ok rank %d" % r in o, o[-3000:]


@pytest.mark.parametrize("device_allreduce,halo", [(False, "0"), (True, "0"), (False, "1"), (False, None)])
def test_multiprocess_rccl(device_allreduce, halo):
    """The route bench.py --gpus N takes: one process and one GPU per rank, backend "rccl" -- set-up handshake (ncclAllGather, grouped
    send/recv all-to-all), host-name check and shared-memory ring attach, the grouped ncclSend/ncclRecv ghost exchange forward and in
    reverse with ADD, reductions through the host ring or (PETSCHIP_DEVICE_ALLREDUCE=1) ncclAllReduce -- against the oracle.
    halo: the ghost exchange as grouped ncclSend/ncclRecv ("0"), as peer writes ("1") or as the library chooses (None: peer writes if
    its trial exchanges pass).  Needs two GPUs: skipped on a one-GPU box (RCCL refuses two ranks on one device)."""
    import subprocess
    import sys
    import ctypes as C
    from petsc_amd import capi, petsc as P
    ndev = C.c_int()
    capi.load().phip_device_count(C.byref(ndev))
    if ndev.value < 2:
        pytest.skip("needs 2 GPUs (RCCL refuses two ranks on one device); this box has %d" % ndev.value)
    n = 2
    key = bytes(P.Comm.unique_id()).hex()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, PHIP_HOST_RING_TIMEOUT="60")
    if device_allreduce:
        env["PETSCHIP_DEVICE_ALLREDUCE"] = "1"
    else:
        env.pop("PETSCHIP_DEVICE_ALLREDUCE", None)
    env.pop("PHIP_HALO_IPC", None)          # halo: "0" grouped ncclSend/ncclRecv, "1" peer writes, None the library's own choice
    if halo is not None:
        env["PHIP_HALO_IPC"] = halo
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "_gpu_mp_worker.py"), key, str(n), str(r), "rccl", str(r)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(n)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=150)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, _ = p.communicate()
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "mp ok rank %d" % r in o, o[-3000:]

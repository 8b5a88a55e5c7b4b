"""CPU tests that pin the oracle (oracle/petsc_oracle.c) before anything trusts it.

 * SpMV / diagonal / MPIAIJ split / scatter plan against tests/golden/mat_*.npz, which
   tests/golden/make_golden.py derived from the reference's own data files with the reference's
   own reader + scipy + numpy (independent of the C oracle).
 * GMRES(30)+Jacobi driver against the reference's golden stdout files
   src/ksp/ksp/examples/tests/output/ex3_1.out and tutorials/output/ex5_1.out (-ksp_monitor_short,
   i.e. %g with 6 significant digits; "< 1.e-11" for tiny residuals).
"""
import os

import numpy as np
import pytest

from oracle.oracle import REFINE_ALWAYS, NORM_1, NORM_2, NORM_INFINITY, NORM_1_AND_2

NAMES = ["tiny", "small", "medium", "arco1", "poisson1", "spd-real-int32-float64", "ns-real-int32-float64", "poisson2", "fem1", "afiro", "golan", "A4", "P4"]


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, "mat_%s.npz" % name), allow_pickle=False)


@pytest.mark.parametrize("name", NAMES)
def test_matmult_vs_scipy(orc, golden_dir, name):
    g = load(golden_dir, name)
    y = orc.matmult(g["i"], g["j"], g["a"], g["x"])
    scale = np.abs(g["y"]).max() + 1e-300
    assert np.abs(y - g["y"]).max() <= 4e-16 * scale
    # MatMultAdd: sum starts at y0[i] (aij.c:1482)
    z = orc.matmultadd(g["i"], g["j"], g["a"], g["x"], g["y0"])
    assert np.abs(z - (g["y0"] + g["y"])).max() <= 1e-14 * (scale + 1)


@pytest.mark.parametrize("name", NAMES)
def test_diagonal(orc, golden_dir, name):
    g = load(golden_dir, name)
    np.testing.assert_array_equal(orc.getdiagonal(g["i"], g["j"], g["a"]), g["diag"])
    np.testing.assert_array_equal(orc.mark_diagonal(g["i"], g["j"]), g["diagpos"])


@pytest.mark.parametrize("name", NAMES)
def test_assembly_roundtrip(orc, golden_dir, name):
    """MatSetValues in shuffled order + MatAssemblyEnd reproduces the file's CSR bit for bit."""
    g = load(golden_dir, name)
    ai, aj, aa = g["i"], g["j"], g["a"]
    M, N = int(g["M"]), int(g["N"])
    rows = np.repeat(np.arange(M, dtype=np.int32), np.diff(ai))
    perm = np.random.default_rng(7).permutation(len(aj))
    A = orc.assemble(M, N, rows[perm], aj[perm], aa[perm], add=False, nz_prealloc=3)
    np.testing.assert_array_equal(A["i"], ai)
    np.testing.assert_array_equal(A["j"], aj)
    np.testing.assert_array_equal(A["a"], aa)
    np.testing.assert_array_equal(A["diag"], g["diagpos"])
    assert A["rmax"] == int(np.diff(ai).max())
    assert A["nonzerorowcnt"] == int(np.count_nonzero(np.diff(ai)))


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("size", [2, 3, 4])
def test_split_and_plan(orc, golden_dir, name, size):
    g = load(golden_dir, name)
    if "s%d_ranges" % size not in g.files:
        pytest.skip("rectangular: the fixture holds no split products")
    N = int(g["N"])
    ai, aj, aa = g["i"], g["j"], g["a"]
    rg = orc.ownership_ranges(N, size)
    np.testing.assert_array_equal(rg, g["s%d_ranges" % size])
    garrays = []
    for r in range(size):
        rs, re = int(rg[r]), int(rg[r + 1])
        sp = orc.mpiaij_split(ai[rs:re + 1] - ai[rs], aj[ai[rs]:ai[re]], aa[ai[rs]:ai[re]], N, rs, re)
        np.testing.assert_array_equal(sp["garray"], g["s%d_garray_%d" % (size, r)])
        np.testing.assert_array_equal(sp["Bj"], g["s%d_Bj_%d" % (size, r)])
        np.testing.assert_array_equal(sp["Bi"], g["s%d_Bi_%d" % (size, r)])
        garrays.append(sp["garray"])
    plans = orc.scatter_plan(rg, garrays)
    for r, p in enumerate(plans):
        for key in ("from_procs", "from_starts", "from_indices", "to_procs", "to_starts", "to_indices"):
            np.testing.assert_array_equal(p[key], g["s%d_%s_%d" % (size, key, r)], err_msg=key)
        assert p["from_contiq"]
        assert p["nlocal"] == 0


@pytest.mark.parametrize("name", ["small", "medium", "arco1"])
@pytest.mark.parametrize("size", [2, 4])
def test_distributed_mult_equals_sequential(orc, golden_dir, name, size):
    g = load(golden_dir, name)
    y1 = orc.operator(g["i"], g["j"], g["a"], 1).mult(g["x"])
    yp = orc.operator(g["i"], g["j"], g["a"], size).mult(g["x"])
    assert np.abs(y1 - yp).max() <= 1e-13 * (np.abs(y1).max() + 1e-300)


@pytest.mark.parametrize("name", NAMES)
def test_matmulttranspose_vs_scipy(orc, golden_dir, name):
    """MatMultTranspose(Add)_SeqAIJ (aij.c:1207-1266) and the MPIAIJ form with its reverse ADD scatter (mpiaij.c:1029-1056,
    1102-1117) on the reference's data files, against scipy's A.T @ x (an independent implementation)."""
    import scipy.sparse as sp
    g = load(golden_dir, name)
    M, N = int(g["M"]), int(g["N"])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(M), rng.standard_normal(N)
    ref = sp.csr_matrix((g["a"], g["j"], g["i"]), shape=(M, N)).T @ x
    scale = np.abs(ref).max() + 1e-300
    assert np.abs(orc.matmulttranspose(g["i"], g["j"], g["a"], N, x) - ref).max() <= 1e-14 * scale
    assert np.abs(orc.matmulttranspose(g["i"], g["j"], g["a"], N, x, y=y0) - (y0 + ref)).max() <= 1e-13 * (scale + 1)
    if M == N:
        for size in (1, 2, 3, 4):
            op = orc.operator(g["i"], g["j"], g["a"], size)
            assert np.abs(op.mult_transpose(x) - ref).max() <= 1e-13 * scale
            assert np.abs(op.mult_transpose(x, y0) - (y0 + ref)).max() <= 1e-13 * (scale + 1)


def test_compressed_row(orc):
    ai = np.array([0, 0, 2, 2, 2, 3, 3, 3, 3, 3, 3], dtype=np.int32)  # 2 of 10 rows non-empty
    use, cpi, ridx = orc.check_compressed_row(ai)
    assert use
    np.testing.assert_array_equal(cpi, [0, 2, 3])
    np.testing.assert_array_equal(ridx, [1, 4])
    use, _, _ = orc.check_compressed_row(np.arange(11, dtype=np.int32))
    assert not use
    # exactly at the ratio: zero rows 6 of 10 -> 6 < 0.6*10 is False -> use (compressedrow.c:49)
    ai = np.array([0, 1, 2, 3, 4, 4, 4, 4, 4, 4, 4], dtype=np.int32)
    assert orc.check_compressed_row(ai)[0]


def test_vec_closed_forms(orc):
    """The closed forms of src/vec/vec/examples/tutorials/ex1.c:71-153 (golden output: all zeros)."""
    n = 20
    one, two, three = 1.0, 2.0, 3.0
    x = np.full(n, one); y = np.full(n, two); z = np.full(n, one)
    assert orc.dot(x, x) == n
    np.testing.assert_array_equal(orc.mdot(x, [x, y, z]), [n, 2 * n, n])
    x = orc.scale(x, two)
    assert orc.norm(x) == pytest.approx(two * np.sqrt(n), abs=1e-14)
    w = orc.axpy(y, three, x)  # y = 2 + 3*2 = 8
    assert orc.norm(w) == pytest.approx(8 * np.sqrt(n), abs=1e-13)
    w = orc.aypx(w, two, x)  # 2 + 2*8 = 18
    assert orc.norm(w) == pytest.approx(18 * np.sqrt(n), abs=1e-13)
    w2 = orc.waxpy(two, x, w)  # 18 + 2*2 = 22
    assert orc.norm(w2) == pytest.approx(22 * np.sqrt(n), abs=1e-13)
    assert orc.norm(orc.pointwisemult(w, x)) == pytest.approx(36 * np.sqrt(n), abs=1e-12)
    assert orc.norm(orc.pointwisedivide(x, w)) == pytest.approx(np.sqrt(n) / 9, abs=1e-14)
    xm = orc.maxpy(np.full(n, one), [one, two, three], [x, w, w2])  # 1 + 2 + 36 + 66
    assert orc.norm(xm) == pytest.approx(105 * np.sqrt(n), abs=1e-12)
    assert orc.norm(x, NORM_1) == 2 * n and orc.norm(x, NORM_INFINITY) == 2.0
    assert orc.norm(x, NORM_1_AND_2) == (2.0 * n, pytest.approx(2 * np.sqrt(n)))
    assert np.isnan(orc.norm(np.array([1.0, np.nan, 3.0]), NORM_INFINITY))


def test_mdot_matches_dot(orc):
    """src/vec/vec/examples/tests/ex43.c:31-51: VecMDot vs per-vector VecDot (there: rel 1e-5)."""
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 3, 5, 100, 1003):
        x = rng.standard_normal(n)
        for nv in range(1, 10):
            ys = [rng.standard_normal(n) for _ in range(nv)]
            z = orc.mdot(x, ys)
            ref = np.array([orc.dot(x, y) for y in ys])
            assert np.abs(z - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
            al = rng.standard_normal(nv)
            w = orc.maxpy(x, al, ys)
            ref = x + sum(a * y for a, y in zip(al, ys))
            assert np.abs(w - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()) if n else True


def test_jacobi_zero_diag(orc):
    d = orc.jacobi_setup(np.array([2.0, 0.0, -4.0]))
    np.testing.assert_array_equal(d, [0.5, 1.0, -0.25])  # jacobi.c:188-193 zero -> 1
    d = orc.jacobi_setup(np.array([2.0, 0.0, -4.0]), useabs=True)
    np.testing.assert_array_equal(d, [0.5, 1.0, 0.25])


def monitor_short(v):
    """-ksp_monitor_short formatting (src/ksp/ksp/interface/iterativ.c KSPMonitorDefaultShort)."""
    if v > 1e-9:
        return "%g" % v
    if v > 1e-11:
        return "%5.3e" % v
    return "< 1.e-11"


def ex3_system(orc, m=5):
    """The linear system of src/ksp/ksp/examples/tests/ex3.c (bilinear Laplacian, Dirichlet rows)."""
    N = (m + 1) ** 2
    h = 1.0 / m
    H = h * h
    Ke = np.array([H / 6, -.125 * H, H / 12, -.125 * H, -.125 * H, H / 6, -.125 * H, H / 12,
                   H / 12, -.125 * H, H / 6, -.125 * H, -.125 * H, H / 12, -.125 * H, H / 6])
    blocks = []
    for e in range(m * m):
        i0 = (m + 1) * (e // m) + (e % m)
        idx = [i0, i0 + 1, i0 + m + 2, i0 + m + 1]
        blocks.append((idx, idx, Ke))
    A = orc.assemble_blocks(N, N, blocks, add=True, nz_prealloc=9)
    brows = [r for r in range(N) if r // (m + 1) in (0, m) or r % (m + 1) in (0, m)]
    aa = A["a"].copy()
    u = np.zeros(N)
    b = np.zeros(N)
    for r in brows:
        u[r] = b[r] = h * (r // (m + 1))
        aa[A["i"][r]:A["i"][r + 1]] = 0.0  # MatZeroRows keeps the pattern, diag 1.0
        aa[A["diag"][r]] = 1.0
    ustar = np.array([h * (i // (m + 1)) for i in range(N)])
    return A["i"], A["j"], aa, b, u, ustar


EX3_1_OUT = ["0.541987", "0.327572", "0.201868", "0.146074", "0.102129", "0.0549947", "0.0216871", "< 1.e-11"]
EX5_1_OUT = [["4.16083", "1.32287", "0.625076", "0.211511", "0.0201553", "< 1.e-11"],
             ["5.16667", "1.01143", "0.241047", "0.0452703", "0.00290094", "< 1.e-11"]]


@pytest.mark.parametrize("size", [1, 2])
def test_gmres_jacobi_ex3_golden(orc, size):
    ai, aj, aa, b, u, ustar = ex3_system(orc)
    res = orc.operator(ai, aj, aa, size).solve(b, "gmres", "jacobi", x0=u, cgs_refine=REFINE_ALWAYS)
    assert [monitor_short(v) for v in res["history"]] == EX3_1_OUT  # ex3_1.out and ex3_2.out are identical
    assert res["its"] == 7 and res["reason"] == 2
    assert np.linalg.norm(res["x"] - ustar) < 1e-14  # ex3 prints nothing when the error is below 1e-14


def ex5_matrix(orc, diagv, m=3, n=2):
    r, c, v = [], [], []
    for Ii in range(m * n):
        i, j = Ii // n, Ii % n
        if i > 0: r.append(Ii); c.append(Ii - n); v.append(-1.0)
        if i < m - 1: r.append(Ii); c.append(Ii + n); v.append(-1.0)
        if j > 0: r.append(Ii); c.append(Ii - 1); v.append(-1.0)
        if j < n - 1: r.append(Ii); c.append(Ii + 1); v.append(-1.0)
        r.append(Ii); c.append(Ii); v.append(diagv)
    return orc.assemble(m * n, m * n, r, c, v)


def test_gmres_jacobi_ex5_golden(orc):
    for k, dv in enumerate((4.0, 6.0)):
        A = ex5_matrix(orc, dv)
        op = orc.operator(A["i"], A["j"], A["a"], 1)
        u = np.arange(6, dtype=float)
        res = op.solve(op.mult(u), "gmres", "jacobi", cgs_refine=REFINE_ALWAYS)
        assert [monitor_short(v) for v in res["history"]] == EX5_1_OUT[k]
        assert res["its"] == 5


EX5_2_OUT = [["142.26", "56.7664", "28.4652", "16.0205", "6.37762", "2.18849", "0.698897", "0.1896", "0.0311023", "0.00173053", "< 1.e-11"],
             ["171.949", "51.7756", "16.2674", "5.93715", "0.787126", "0.156071", "0.0363183", "0.005189", "0.000484311", "1.74022e-05"]]


def test_gmres_jacobi_ex5_2_golden(orc):
    """tutorials/output/ex5_2.out: ex5 on 2 ranks (n = 2 * size = 4, 12 unknowns, u = local index + 100 * rank), -ksp_rtol .000001."""
    u = np.concatenate([np.arange(6.0), np.arange(6.0) + 100.0])
    for k, dv in enumerate((4.0, 6.0)):
        A = ex5_matrix(orc, dv, 3, 4)
        op = orc.operator(A["i"], A["j"], A["a"], 2)
        res = op.solve(op.mult(u), "gmres", "jacobi", rtol=1e-6, cgs_refine=REFINE_ALWAYS)
        assert [monitor_short(v) for v in res["history"]] == EX5_2_OUT[k]
        assert res["its"] == (10, 9)[k]


@pytest.mark.parametrize("size", [1, 2])
def test_gmres_jacobi_ex2f_golden(orc, size):
    """tutorials/output/ex2f_1.out: the 3 x 3 five-point Laplacian of ex2f.F, u = 1, b = A u, GMRES + Jacobi, refine_always."""
    A = ex5_matrix(orc, 4.0, 3, 3)
    op = orc.operator(A["i"], A["j"], A["a"], size)
    res = op.solve(op.mult(np.ones(9)), "gmres", "jacobi", cgs_refine=REFINE_ALWAYS)
    assert [monitor_short(v) for v in res["history"]] == ["1.11803", "0.591608", "0.316228", "< 1.e-11"] and res["its"] == 3
    assert np.linalg.norm(res["x"] - 1.0) < 1e-12


def test_vecdot_known_answer_ex18(orc):
    """src/vec/vec/examples/tests/ex18.c:27-35 and tests/output/ex18_1.out: a 13-digit known answer for VecDot."""
    i = np.arange(15, dtype=np.float64)
    x = i + 1.0 / (i + .35)
    y = x + 1.375547826473644376
    assert "Vector inner product %16.12e" % orc.dot(x, y) == "Vector inner product 1.202435522115e+03"


def test_cg_jacobi_poisson(orc, golden_dir):
    """No reference golden exists for KSPCG+PCJACOBI; check the driver against numpy algebra."""
    g = load(golden_dir, "spd-real-int32-float64")
    op = orc.operator(g["i"], g["j"], g["a"], 1)
    b = g["vec0"]
    res = op.solve(b, "cg", "jacobi", rtol=1e-12)
    assert res["reason"] == 2
    assert np.linalg.norm(op.mult(res["x"]) - b) <= 1e-10 * np.linalg.norm(b)
    res2 = orc.operator(g["i"], g["j"], g["a"], 3).solve(b, "cg", "jacobi", rtol=1e-12)
    assert abs(res2["its"] - res["its"]) <= 1


def test_threaded_ranks_same_bits(tmp_path):
    """The CPU baseline of bench.py spreads the emulated MPIAIJ ranks over OpenMP threads (fast build, -fopenmp): ranks are independent
    and reductions are summed in rank order, so the results are bit-identical to the one-thread emulation."""
    from oracle import oracle
    from petsc_amd import generators as G
    o = oracle.Oracle(path=oracle.build(fast=True, outdir=str(tmp_path)))
    ai, aj, aa = G.laplace3d_7pt(20, dirichlet_rows=False)
    n = len(ai) - 1
    b = o.operator(ai, aj, aa, 1).mult(np.ones(n))
    seq = o.operator(ai, aj, aa, 4, 1).solve(b, "cg", "jacobi", rtol=1e-8)
    par = o.operator(ai, aj, aa, 4, 4).solve(b, "cg", "jacobi", rtol=1e-8)
    assert seq["its"] == par["its"] and seq["rnorm"] == par["rnorm"]
    np.testing.assert_array_equal(seq["x"], par["x"])
    np.testing.assert_array_equal(seq["history"], par["history"])

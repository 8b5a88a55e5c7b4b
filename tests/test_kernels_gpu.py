"""GPU parity tests of the kernel C ABI (libpetschip.so) against the CPU oracle and the golden fixtures.

Tolerances (BASELINE.json): SpMV <= 1e-12 relative in double; index work bit-exact; reductions are compared at
1e-13 relative to sum|x_i y_i| (the reference pins dot/norm only to 1e-5, src/vec/vec/examples/tests/ex43.c:41-46).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ["tiny", "small", "medium", "arco1", "poisson1", "spd-real-int32-float64", "ns-real-int32-float64", "poisson2", "fem1", "afiro", "golan", "A4", "P4"]


@pytest.fixture(scope="module")
def ctx():
    from petsc_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, "mat_%s.npz" % name), allow_pickle=False)


def relerr(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


def mat_format(ctx, A):
    import ctypes as C
    f, nd, pad = C.c_int(), C.c_int32(), C.c_int64()
    ctx.check(ctx.lib.phip_mat_format_info(A, C.byref(f), C.byref(nd), C.byref(pad)))
    return f.value, nd.value, pad.value


def gpu_mult(ctx, ai, aj, aa, n, x, cprow=None, y0=None, inplace=False, tpr=0, fmt=0):
    """fmt 1: CSR row-block kernel, 2: SELL-dict kernel (returns None when the matrix is not eligible), 0: the library's choice."""
    m = len(ai) - 1
    A = ctx.mat_create(m, n, ai, aj, aa, cprow)
    if tpr:
        fmt = 1  # lanes per row is a knob of the CSR row-block kernel
        ctx.check(ctx.lib.phip_mat_set_threads_per_row(A, tpr))
    if fmt == 2 and ctx.lib.phip_mat_set_format(A, 2) != 0:
        ctx.check(ctx.lib.phip_mat_destroy(A))
        return None
    if fmt == 1:
        ctx.check(ctx.lib.phip_mat_set_format(A, 1))
    dx = ctx.to_device(x)
    dy = ctx.to_device(np.full(max(m, 1), 7.25))
    if y0 is None:
        ctx.check(ctx.lib.phip_mat_mult(A, dx, dy))
        out = ctx.to_host(dy, m)
    else:
        dy0 = ctx.to_device(y0 if m else np.zeros(1))
        if inplace:
            ctx.check(ctx.lib.phip_mat_multadd(A, dx, dy0, dy0))
            out = ctx.to_host(dy0, m)
        else:
            ctx.check(ctx.lib.phip_mat_multadd(A, dx, dy0, dy))
            out = ctx.to_host(dy, m)
        ctx.free(dy0)
    ctx.free(dx); ctx.free(dy)
    ctx.check(ctx.lib.phip_mat_destroy(A))
    return out


@pytest.mark.parametrize("name", NAMES)
def test_spmv_golden(ctx, orc, golden_dir, name):
    g = load(golden_dir, name)
    y = gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"])
    assert relerr(y, g["y"]) <= 1e-12
    # one lane per row reproduces the reference's k-ascending sum bit for bit (no FMA contraction on either side)
    yo = orc.matmult(g["i"], g["j"], g["a"], g["x"])
    y1 = gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"], tpr=1)
    np.testing.assert_array_equal(y1, yo)
    for tpr in (2, 8, 64):
        assert relerr(gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"], tpr=tpr), g["y"]) <= 1e-12
    z = gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"], y0=g["y0"])
    assert relerr(z, orc.matmultadd(g["i"], g["j"], g["a"], g["x"], g["y0"])) <= 1e-12
    z = gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"], y0=g["y0"], inplace=True)
    assert relerr(z, orc.matmultadd(g["i"], g["j"], g["a"], g["x"], g["y0"])) <= 1e-12


@pytest.mark.parametrize("name", NAMES)
def test_getdiagonal_golden(ctx, golden_dir, name):
    g = load(golden_dir, name)
    m = int(g["M"])
    A = ctx.mat_create(m, int(g["N"]), g["i"], g["j"], g["a"])
    dv = ctx.to_device(np.full(m, -3.0))
    ctx.check(ctx.lib.phip_mat_getdiagonal(A, dv))
    np.testing.assert_array_equal(ctx.to_host(dv, m), g["diag"])
    ctx.free(dv); ctx.check(ctx.lib.phip_mat_destroy(A))


def test_spmv_stencils_and_rmat(ctx, orc):
    from petsc_amd import generators as G
    cases = [G.laplace3d_7pt(24), G.box3d_27pt(17), G.convdiff3d_7pt(19, 13, 11), G.laplace2d_5pt(57, 31), G.rmat(13)]
    for ai, aj, aa in cases:
        n = len(ai) - 1
        x = G.random_vector(n)
        y = gpu_mult(ctx, ai, aj, aa, n, x)
        assert relerr(y, orc.matmult(ai, aj, aa, x)) <= 1e-12


def test_spmv_sell_dict_layout(ctx, orc, golden_dir):
    """The SELL-dict layout (one-byte diagonal index) gives the reference's bits, like the one-lane-per-row CSR kernel; it is
    chosen for the stencil matrices and refused for irregular ones."""
    from petsc_amd import generators as G
    rng = np.random.default_rng(3)
    cases = [G.laplace3d_7pt(24), G.laplace3d_7pt(9, 31, 5, dirichlet_rows=False), G.box3d_27pt(17), G.convdiff3d_7pt(19, 13, 11),
             G.laplace2d_5pt(57, 31), G.laplace2d_5pt(1, 700)]
    for ai, aj, aa in cases:
        n = len(ai) - 1
        aa = aa * rng.uniform(0.5, 1.5, len(aa))
        x = G.random_vector(n)
        yo = orc.matmult(ai, aj, aa, x)
        A = ctx.mat_create(n, n, ai, aj, aa)
        f, nd, pad = mat_format(ctx, A)
        assert f == 2 and nd <= 27 and pad >= len(aj)
        ctx.check(ctx.lib.phip_mat_destroy(A))
        np.testing.assert_array_equal(gpu_mult(ctx, ai, aj, aa, n, x, fmt=2), yo)
        np.testing.assert_array_equal(gpu_mult(ctx, ai, aj, aa, n, x, tpr=1), yo)
        y0 = rng.standard_normal(n)
        zo = orc.matmultadd(ai, aj, aa, x, y0)
        np.testing.assert_array_equal(gpu_mult(ctx, ai, aj, aa, n, x, y0=y0, fmt=2), zo)
        np.testing.assert_array_equal(gpu_mult(ctx, ai, aj, aa, n, x, y0=y0, inplace=True, fmt=2), zo)
    # rectangular, odd row count, empty rows in the middle, x containing Inf where no entry points (padding must not touch it)
    ai = np.array([0, 2, 2, 3, 5, 5], np.int32); aj = np.array([0, 1, 2, 3, 4], np.int32); aa = np.arange(1.0, 6.0)
    x = np.array([1.0, 2.0, 3.0, 4.0, 5.0, np.inf, np.nan])
    np.testing.assert_array_equal(gpu_mult(ctx, ai, aj, aa, 7, x, fmt=2), orc.matmult(ai, aj, aa, x))
    # irregular matrices stay on the CSR row-block kernel
    ai, aj, aa = G.rmat(12)
    assert gpu_mult(ctx, ai, aj, aa, len(ai) - 1, np.ones(len(ai) - 1), fmt=2) is None
    for name in ("arco1", "medium", "small", "tiny"):
        g = np.load(os.path.join(golden_dir, "mat_%s.npz" % name), allow_pickle=False)
        for fmt in (1, 2):
            y = gpu_mult(ctx, g["i"], g["j"], g["a"], int(g["N"]), g["x"], fmt=fmt)
            if y is not None:
                assert relerr(y, g["y"]) <= 1e-12
    # values-only update keeps both layouts in step
    ai, aj, aa = G.laplace3d_7pt(10)
    n = len(ai) - 1
    A = ctx.mat_create(n, n, ai, aj, aa)
    aa2 = aa * 1.75 + 0.125
    ctx.check(ctx.lib.phip_mat_update_values(A, aa2.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double))))
    x = G.random_vector(n); dx = ctx.to_device(x); dy = ctx.to_device(np.zeros(n))
    for fmt in (2, 1):
        ctx.check(ctx.lib.phip_mat_set_format(A, fmt))
        ctx.check(ctx.lib.phip_mat_mult(A, dx, dy))
        got = ctx.to_host(dy, n)
        if fmt == 2:
            np.testing.assert_array_equal(got, orc.matmult(ai, aj, aa2, x))
        else:
            assert relerr(got, orc.matmult(ai, aj, aa2, x)) <= 1e-12
    ctx.free(dx); ctx.free(dy); ctx.check(ctx.lib.phip_mat_destroy(A))


def test_spmv_long_rows_and_edges(ctx, orc):
    rng = np.random.default_rng(5)
    # rows of 0, 1, 2015 ... 2017 (the block capacity is 2016 behind up to 31 entries of alignment slack), 2046, 2047, 5000 and 40000
    # nonzeros; every start alignment of a full block (the 1..31 nonzeros in front); empty leading/trailing rows
    n = 50000
    lens = [0, 0, 1, 2046, 2047, 3, 0, 5000, 2045, 2, 40000, 0, 17, 0, 2016, 2017, 2015, 1, 2016, 31, 2016, 7, 2010, 6, 5, 2016]
    ai = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aj = np.concatenate([np.sort(rng.choice(n, L, replace=False)) for L in lens]).astype(np.int32)
    aa = rng.standard_normal(len(aj))
    x = rng.standard_normal(n)
    y = gpu_mult(ctx, ai, aj, aa, n, x)
    yo = orc.matmult(ai, aj, aa, x)
    assert np.abs(y - yo).max() <= 1e-12 * np.abs(aa).max() * np.abs(x).max() * 40000 ** 0.5
    # empty matrix, matrix with no nonzeros
    assert len(gpu_mult(ctx, np.zeros(1, np.int32), np.zeros(0, np.int32), np.zeros(0), 0, np.zeros(1))) == 0
    y = gpu_mult(ctx, np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0), 5, np.ones(5))
    np.testing.assert_array_equal(y, np.zeros(5))
    # no columns at all and a null x (allowed when there are no nonzeros): the row-block kernel must not gather anything
    A = ctx.mat_create(5, 0, np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0))
    dy = ctx.to_device(np.full(5, 7.0))
    ctx.check(ctx.lib.phip_mat_mult(A, None, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, 5), np.zeros(5))
    dz = ctx.to_device(np.arange(5.0))
    ctx.check(ctx.lib.phip_mat_multadd(A, None, dz, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, 5), np.arange(5.0))
    ctx.free(dy); ctx.free(dz); ctx.check(ctx.lib.phip_mat_destroy(A))


def test_spmv_compressed_row(ctx, orc):
    """Off-diagonal-block shape: >= 60% empty rows -> Mat_CompressedRow (compressedrow.c:49-72)."""
    rng = np.random.default_rng(11)
    m, n = 1000, 300
    lens = np.where(rng.random(m) < 0.15, rng.integers(1, 9, m), 0)
    ai = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aj = np.concatenate([np.sort(rng.choice(n, L, replace=False)) for L in lens if L]).astype(np.int32)
    aa = rng.standard_normal(len(aj))
    use, cpi, ridx = orc.check_compressed_row(ai)
    assert use
    x = rng.standard_normal(n); y0 = rng.standard_normal(m)
    y = gpu_mult(ctx, ai, aj, aa, n, x, cprow=(cpi, ridx))
    np.testing.assert_array_equal(y, orc.matmult(ai, aj, aa, x, cprow=(cpi, ridx)))
    for inplace in (False, True):
        z = gpu_mult(ctx, ai, aj, aa, n, x, cprow=(cpi, ridx), y0=y0, inplace=inplace)
        np.testing.assert_array_equal(z, orc.matmultadd(ai, aj, aa, x, y0, cprow=(cpi, ridx)))


def gpu_mult_transpose(ctx, ai, aj, aa, n, x, cprow=None, y0=None, inplace=False, tpr=0, fmt=0, new_values=None):
    m = len(ai) - 1
    A = ctx.mat_create(m, n, ai, aj, aa, cprow)
    if tpr:
        fmt = 1
        ctx.check(ctx.lib.phip_mat_set_threads_per_row(A, tpr))
    if fmt == 1:
        ctx.check(ctx.lib.phip_mat_set_format(A, 1))
    ai32, aj32, aa64 = np.ascontiguousarray(ai, np.int32), np.ascontiguousarray(aj, np.int32), np.ascontiguousarray(aa, np.float64)
    import ctypes as C
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ctx.check(ctx.lib.phip_mat_transpose_setup(A, ip(ai32), ip(aj32), dp(aa64)))
    if new_values is not None:
        nv = np.ascontiguousarray(new_values, np.float64)
        ctx.check(ctx.lib.phip_mat_update_values(A, dp(nv)))
    dx = ctx.to_device(x if m else np.zeros(1))
    dy = ctx.to_device(np.full(max(n, 1), 7.25))
    if y0 is None:
        ctx.check(ctx.lib.phip_mat_mult_transpose(A, dx, dy))
        out = ctx.to_host(dy, n)
    else:
        dy0 = ctx.to_device(y0 if n else np.zeros(1))
        ctx.check(ctx.lib.phip_mat_mult_transpose_add(A, dx, dy0, dy0 if inplace else dy))
        out = ctx.to_host(dy0 if inplace else dy, n)
        ctx.free(dy0)
    ctx.free(dx); ctx.free(dy)
    ctx.check(ctx.lib.phip_mat_destroy(A))
    return out


@pytest.mark.parametrize("name", NAMES)
def test_mult_transpose_golden(ctx, orc, golden_dir, name):
    """y = A^T x and z = y + A^T x (aij.c:1207-1266) on the reference's data files: within 1e-12 of scipy's A.T @ x, and bit for bit
    the oracle's scatter-add order when a row of the transposed copy is summed by one lane."""
    import scipy.sparse as sp
    g = load(golden_dir, name)
    m, n = int(g["M"]), int(g["N"])
    rng = np.random.default_rng(5)
    x, y0 = rng.standard_normal(m), rng.standard_normal(n)
    ref = sp.csr_matrix((g["a"], g["j"], g["i"]), shape=(m, n)).T @ x
    yo = orc.matmulttranspose(g["i"], g["j"], g["a"], n, x)
    assert relerr(yo, ref) <= 1e-13
    assert relerr(gpu_mult_transpose(ctx, g["i"], g["j"], g["a"], n, x), ref) <= 1e-12
    np.testing.assert_array_equal(gpu_mult_transpose(ctx, g["i"], g["j"], g["a"], n, x, tpr=1), yo)
    zo = orc.matmulttranspose(g["i"], g["j"], g["a"], n, x, y=y0)
    for inplace in (False, True):
        np.testing.assert_array_equal(gpu_mult_transpose(ctx, g["i"], g["j"], g["a"], n, x, y0=y0, inplace=inplace, tpr=1), zo)
        assert relerr(gpu_mult_transpose(ctx, g["i"], g["j"], g["a"], n, x, y0=y0, inplace=inplace), zo) <= 1e-12


def test_mult_transpose_shapes(ctx, orc):
    """Stencil matrices (the transposed copy is stencil-like too and takes the SELL-dict layout when eligible), a rectangular compressed-row matrix (the
    off-diagonal block's shape), R-MAT, new values through phip_mat_update_values, and the call-order error."""
    from petsc_amd import generators as G
    from petsc_amd.capi import PhipError
    rng = np.random.default_rng(6)
    for ai, aj, aa in (G.convdiff3d_7pt(19, 13, 11), G.box3d_27pt(12), G.laplace2d_5pt(40, 23)):
        n = len(ai) - 1
        x = rng.standard_normal(n)
        assert relerr(gpu_mult_transpose(ctx, ai, aj, aa, n, x), orc.matmulttranspose(ai, aj, aa, n, x)) <= 1e-12  # the library's layout choice
        np.testing.assert_array_equal(gpu_mult_transpose(ctx, ai, aj, aa, n, x, tpr=1), orc.matmulttranspose(ai, aj, aa, n, x))
        aa2 = aa * rng.uniform(0.5, 2.0, len(aa))
        np.testing.assert_array_equal(gpu_mult_transpose(ctx, ai, aj, aa, n, x, tpr=1, new_values=aa2), orc.matmulttranspose(ai, aj, aa2, n, x))
        assert relerr(gpu_mult_transpose(ctx, ai, aj, aa, n, x, new_values=aa2), orc.matmulttranspose(ai, aj, aa2, n, x)) <= 1e-12
    ai, aj, aa = G.rmat(12)
    n = len(ai) - 1
    x = rng.standard_normal(n)
    assert relerr(gpu_mult_transpose(ctx, ai, aj, aa, n, x), orc.matmulttranspose(ai, aj, aa, n, x)) <= 1e-12
    m, n = 1000, 300
    lens = np.where(rng.random(m) < 0.15, rng.integers(1, 9, m), 0)
    ai = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    aj = np.concatenate([np.sort(rng.choice(n, L, replace=False)) for L in lens if L]).astype(np.int32)
    aa = rng.standard_normal(len(aj))
    use, cpi, ridx = orc.check_compressed_row(ai)
    assert use
    x, y0 = rng.standard_normal(m), rng.standard_normal(n)
    np.testing.assert_array_equal(gpu_mult_transpose(ctx, ai, aj, aa, n, x, cprow=(cpi, ridx), tpr=1), orc.matmulttranspose(ai, aj, aa, n, x, cprow=(cpi, ridx)))
    np.testing.assert_array_equal(gpu_mult_transpose(ctx, ai, aj, aa, n, x, cprow=(cpi, ridx), y0=y0, tpr=1), orc.matmulttranspose(ai, aj, aa, n, x, y=y0, cprow=(cpi, ridx)))
    # empty matrix: A^T x = 0
    z = gpu_mult_transpose(ctx, np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0), 5, np.ones(3))
    np.testing.assert_array_equal(z, np.zeros(5))
    A = ctx.mat_create(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 1], np.int32), np.ones(2))
    d = ctx.to_device(np.ones(2)); e = ctx.to_device(np.ones(2))
    with pytest.raises(PhipError):
        ctx.check(ctx.lib.phip_mat_mult_transpose(A, d, e))
    ctx.free(d); ctx.free(e); ctx.check(ctx.lib.phip_mat_destroy(A))


def test_value_pair_dictionary(ctx, orc):
    """SELL-dict matrices with <= 256 distinct (diagonal, value) pairs are read through one byte per entry: same bits as with
    streamed values (and as the oracle), including -0.0 and denormal entries; phip_mat_update_values follows the values in and out
    of the dictionary."""
    import ctypes as C
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket
    L = ctx.lib
    rng = np.random.default_rng(5)

    def info(A):
        b, u, npairs = C.c_int(), C.c_int(), C.c_int32()
        ctx.check(L.phip_mat_value_dict_info(A, C.byref(b), C.byref(u), C.byref(npairs)))
        return b.value, u.value, npairs.value

    def products(A, m, x, v, y0):
        dx, dv, dy0 = ctx.to_device(x), ctx.to_device(v), ctx.to_device(y0)
        dy, dz, dt = (ctx.to_device(np.zeros(m)) for _ in range(3))
        ctx.check(L.phip_mat_mult(A, dx, dy))
        ctx.check(L.phip_mat_multadd(A, dx, dy0, dz))
        t = Ticket()
        val = np.zeros(1)
        ctx.check(L.phip_mat_mult_dot(A, dx, dt, dv, 1, C.byref(t)))
        ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
        out = [ctx.to_host(dy, m), ctx.to_host(dz, m), ctx.to_host(dt, m), val.copy()]
        ctx.check(L.phip_mat_mult_transpose(A, dv, dt))
        out.append(ctx.to_host(dt, m))
        for pz in (dx, dv, dy0, dy, dz, dt):
            ctx.free(pz)
        return out

    cases = [G.laplace3d_7pt(24), G.box3d_27pt(17), G.convdiff3d_7pt(19, 13, 11), G.laplace2d_5pt(57, 31), G.laplace2d_5pt(1, 700),
             G.laplace2d_5pt(700, 1), G.laplace3d_7pt(3, 5, 300)]
    for ci, (ai, aj, aa) in enumerate(cases):
        m = len(ai) - 1
        aa = aa.copy()
        if ci == 0:   # values whose bits matter: -0.0, a denormal, the largest double
            aa[5], aa[77], aa[300] = -0.0, 5e-324, 1.7976931348623157e308
        x, v, y0 = G.random_vector(m), rng.standard_normal(m), rng.standard_normal(m)
        A = ctx.mat_create(m, m, ai, aj, aa)
        built, used, npairs = info(A)
        assert (built, used) == (1, 1) and 1 <= npairs <= 256
        pi, pj, pa = (ai.ctypes.data_as(C.POINTER(C.c_int32)), aj.ctypes.data_as(C.POINTER(C.c_int32)), aa.ctypes.data_as(C.POINTER(C.c_double)))
        ctx.check(L.phip_mat_transpose_setup(A, pi, pj, pa))
        # three kernels over the same matrix: diagonal masks (<= 32 pairs), pair codes, streamed values
        db, du, dn = C.c_int(), C.c_int(), C.c_int32()
        ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
        assert (db.value, du.value) == (1, 1) and dn.value == npairs <= 32
        on = products(A, m, x, v, y0)
        ctx.check(L.phip_set_default_diag_mask(2))   # the fused dot through the masks too (default: only from 40 M nonzeros on)
        try:
            on2 = products(A, m, x, v, y0)
        finally:
            ctx.check(L.phip_set_default_diag_mask(1))
        for k, (a, b) in enumerate(zip(on, on2)):
            if k == 3:
                assert abs(a[0] - b[0]) <= 1e-13 * np.abs(v * on[2]).sum()
            else:
                np.testing.assert_array_equal(a, b)
        # the mask kernels (by default size and structure decide): bit sets per row; group descriptors with transposed masks
        for kind in (4, 5):
            ctx.check(L.phip_set_dmask_kernel(kind))
            ctx.check(L.phip_set_default_diag_mask(2))
            try:
                onk = products(A, m, x, v, y0)
            finally:
                ctx.check(L.phip_set_default_diag_mask(1))
                ctx.check(L.phip_set_dmask_kernel(0))
            for k, (a, b) in enumerate(zip(on, onk)):
                if k == 3:
                    assert abs(a[0] - b[0]) <= 1e-13 * np.abs(v * on[2]).sum()
                else:
                    np.testing.assert_array_equal(a, b)
        ctx.check(L.phip_mat_set_diag_mask(A, 0))
        ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
        assert (db.value, du.value) == (1, 0)
        pairs = products(A, m, x, v, y0)
        ctx.check(L.phip_mat_set_value_dict(A, 0))
        assert info(A)[1] == 0
        off = products(A, m, x, v, y0)
        for other in (pairs, off):
            for k, (a, b) in enumerate(zip(on, other)):
                if k == 3:   # the fused dot: the kernels map rows to lanes differently, so its partial sums are grouped differently
                    assert abs(a[0] - b[0]) <= 1e-13 * np.abs(v * on[2]).sum()
                else:
                    np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(on[0], orc.matmult(ai, aj, aa, x))
        ctx.check(L.phip_mat_set_value_dict(A, 1))
        ctx.check(L.phip_mat_set_diag_mask(A, 1))
        # values that no longer fit: the matrix goes back to streamed values, and returns to the dictionary when they fit again
        many = aa * rng.uniform(0.5, 1.5, len(aa))
        ctx.check(L.phip_mat_update_values(A, many.ctypes.data_as(C.POINTER(C.c_double))))
        if len(aa) > 2000:
            assert info(A)[:2] == (0, 0)
            assert L.phip_mat_set_value_dict(A, 1) != 0
        np.testing.assert_array_equal(products(A, m, x, v, y0)[0], orc.matmult(ai, aj, many, x))
        few = np.round(many / 2.0) * 2.0 + 0.25
        ctx.check(L.phip_mat_update_values(A, few.ctypes.data_as(C.POINTER(C.c_double))))
        assert info(A)[:2] == (1, 1)
        got = products(A, m, x, v, y0)
        np.testing.assert_array_equal(got[0], orc.matmult(ai, aj, few, x))
        np.testing.assert_array_equal(got[4], orc.matmulttranspose(ai, aj, few, m, v))
        ctx.check(L.phip_mat_destroy(A))
    # more than 32 pairs: pair codes, no masks
    ai, aj, aa = G.laplace3d_7pt(24)
    aa = aa.copy()
    aa[::7] = np.round(rng.uniform(1, 12, len(aa[::7])))       # a dozen more values on top of the stencil's
    m = len(ai) - 1
    A = ctx.mat_create(m, m, ai, aj, aa)
    db, du, dn = C.c_int(), C.c_int(), C.c_int32()
    ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
    assert info(A)[:2] == (1, 1) and 32 < info(A)[2] <= 256 and (db.value, du.value) == (0, 0)
    assert L.phip_mat_set_diag_mask(A, 1) != 0
    x = G.random_vector(m)
    dx, dy = ctx.to_device(x), ctx.to_device(np.zeros(m))
    ctx.check(L.phip_mat_mult(A, dx, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, m), orc.matmult(ai, aj, aa, x))
    ctx.free(dx); ctx.free(dy)
    ctx.check(L.phip_mat_destroy(A))
    # never built when switched off, nor for matrices outside the SELL-dict layout
    ctx.check(L.phip_set_default_value_dict(0))
    try:
        ai, aj, aa = cases[0]
        A = ctx.mat_create(len(ai) - 1, len(ai) - 1, ai, aj, aa)
        assert info(A) == (0, 0, 0)
        ctx.check(L.phip_mat_destroy(A))
    finally:
        ctx.check(L.phip_set_default_value_dict(1))
    ai, aj, aa = G.rmat(12)
    A = ctx.mat_create(len(ai) - 1, len(ai) - 1, ai, aj, aa)
    assert info(A) == (0, 0, 0)
    ctx.check(L.phip_mat_destroy(A))


def test_spmv_large_result_nontemporal(ctx, orc):
    """A product with more than 8 M rows stores its result with non-temporal stores (all three stencil kernels): same bits as the oracle,
    for the plain product, MultAdd and the product with the fused dot (which at this size also goes through the masks)."""
    import ctypes as C
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket
    L = ctx.lib
    ai, aj, aa = G.laplace2d_5pt(2900, 2900)
    m = len(ai) - 1
    assert m >= (8 << 20) and len(aj) >= 40000000
    x, y0 = G.random_vector(m), G.random_vector(m, 3)
    A = ctx.mat_create(m, m, ai, aj, aa)
    db, du, dn = C.c_int(), C.c_int(), C.c_int32()
    ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
    assert (db.value, du.value) == (1, 1)
    dx, dy, dy0 = ctx.to_device(x), ctx.to_device(np.zeros(m)), ctx.to_device(y0)
    ctx.check(L.phip_mat_mult(A, dx, dy))
    yo = orc.matmult(ai, aj, aa, x)
    np.testing.assert_array_equal(ctx.to_host(dy, m), yo)
    ctx.check(L.phip_mat_multadd(A, dx, dy0, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, m), orc.matmultadd(ai, aj, aa, x, y0))
    t = Ticket()
    val = np.zeros(1)
    ctx.check(L.phip_mat_mult_dot(A, dx, dy, dx, 1, C.byref(t)))
    ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
    np.testing.assert_array_equal(ctx.to_host(dy, m), yo)
    assert abs(val[0] - x @ yo) <= 1e-12 * np.abs(x * yo).sum()
    # the same through the pair codes and through streamed values
    ctx.check(L.phip_mat_set_diag_mask(A, 0))
    ctx.check(L.phip_mat_mult(A, dx, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, m), yo)
    ctx.check(L.phip_mat_set_value_dict(A, 0))
    ctx.check(L.phip_mat_mult(A, dx, dy))
    np.testing.assert_array_equal(ctx.to_host(dy, m), yo)
    for p in (dx, dy, dy0):
        ctx.free(p)
    ctx.check(L.phip_mat_destroy(A))


def test_hot_column_order(ctx, orc):
    """The hot-column order (columns numbered by descending use, x copied into that order by every product) leaves every product and
    every summation order as it was: bit-identical results for mult, multadd, the fused dot, the transpose and the compressed-row
    form; getdiagonal still reads the original columns; small matrices do not get the order unless it is forced."""
    import ctypes as C
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket
    L = ctx.lib
    rng = np.random.default_rng(77)

    def info(A):
        b, u, s0, s1 = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        ctx.check(L.phip_mat_hot_columns_info(A, C.byref(b), C.byref(u), C.byref(s0), C.byref(s1)))
        return b.value, u.value, s0.value, s1.value

    ai, aj, aa = G.rmat(13)
    n = len(ai) - 1
    A = ctx.mat_create(n, n, ai, aj, aa)
    assert info(A)[:2] == (0, 0)                      # default rule: x far too small
    assert L.phip_mat_set_hot_columns(A, 1) != 0
    ctx.check(L.phip_mat_destroy(A))
    ctx.check(L.phip_set_default_hot_columns(2))
    try:
        # a rectangular compressed-row block (the off-diagonal block's shape), a matrix with a row longer than a row block, R-MAT
        nz_rows = np.sort(rng.choice(500, 60, replace=False)).astype(np.int32)
        ri = np.zeros(501, dtype=np.int32)
        ri[nz_rows + 1] = rng.integers(1, 40, 60)
        ri = np.cumsum(ri).astype(np.int32)
        rj = np.concatenate([np.sort(rng.choice(3000, ri[r + 1] - ri[r], replace=False)) for r in range(500)]).astype(np.int32)
        ra = rng.standard_normal(len(rj))
        cp = (np.concatenate([[0], np.cumsum(np.diff(ri)[nz_rows])]).astype(np.int32), nz_rows)
        li = np.array([0, 3, 5003, 5004], dtype=np.int32)
        lj = np.concatenate([[0, 2, 5], np.arange(5000), [7]]).astype(np.int32)
        la = rng.standard_normal(len(lj))
        cases = [(ai, aj, aa, n, None), (ri, rj, ra, 3000, cp), (li, lj, la, 5000, None)]
        for ci, cj, ca, ncols, cprow in cases:
            m = len(ci) - 1
            x, v, y0 = rng.standard_normal(ncols), rng.standard_normal(m), rng.standard_normal(m)
            A = ctx.mat_create(m, ncols, ci, cj, ca, cprow)
            built, used, s0, s1 = info(A)
            assert (built, used) == (1, 1) and s1 >= s0 - 1e-15
            ctx.check(L.phip_mat_transpose_setup(A, ci.ctypes.data_as(C.POINTER(C.c_int32)), cj.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 ca.ctypes.data_as(C.POINTER(C.c_double))))
            dx, dv, dy0 = ctx.to_device(x), ctx.to_device(v), ctx.to_device(y0)
            dy, dz, dt, dd = (ctx.to_device(np.zeros(max(m, ncols))) for _ in range(4))
            out = {}
            for on in (1, 0):
                ctx.check(L.phip_mat_set_hot_columns(A, on))
                ctx.check(L.phip_mat_mult(A, dx, dy))
                ctx.check(L.phip_mat_multadd(A, dx, dy0, dz))
                t = Ticket()
                val = np.zeros(1)
                ctx.check(L.phip_mat_mult_dot(A, dx, dt, dv, 1, C.byref(t)))
                ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
                y, z, yd = ctx.to_host(dy, m), ctx.to_host(dz, m), ctx.to_host(dt, m)
                ctx.check(L.phip_mat_mult_transpose(A, dv, dt))
                yt = ctx.to_host(dt, ncols)
                ctx.check(L.phip_mat_getdiagonal(A, dd))
                out[on] = (y, z, yd, val.copy(), yt, ctx.to_host(dd, m))
            for a, b in zip(out[0], out[1]):
                np.testing.assert_array_equal(a, b)
            assert relerr(out[1][0], orc.matmult(ci, cj, ca, x)) <= 1e-12
            for pz in (dx, dv, dy0, dy, dz, dt, dd):
                ctx.free(pz)
            ctx.check(L.phip_mat_destroy(A))
        # the stencil layout has no use for it
        si, sj, sa = G.laplace3d_7pt(12)
        A = ctx.mat_create(len(si) - 1, len(si) - 1, si, sj, sa)
        assert info(A)[:2] == (0, 0)
        ctx.check(L.phip_mat_destroy(A))
    finally:
        ctx.check(L.phip_set_default_hot_columns(1))


def test_plan_covers_rows(ctx):
    from petsc_amd import generators as G
    for ai, aj, aa in (G.laplace3d_7pt(20), G.rmat(12)):
        m = len(ai) - 1
        A = ctx.mat_create(m, m, ai, aj, aa)
        p = ctx.mat_plan(A)
        br = p["block_row"]
        assert br[0] == 0 and br[-1] == m and np.all(np.diff(br) > 0)
        nnzb = ai[br[1:]] - ai[br[:-1]]
        single = np.diff(br) == 1
        assert np.all((nnzb <= p["cap"]) | single)
        assert np.all(np.diff(br) <= 256)
        ctx.check(ctx.lib.phip_mat_destroy(A))


def test_mat_create_rejects_bad_input(ctx):
    from petsc_amd.capi import PhipError
    with pytest.raises(PhipError):  # column out of range
        ctx.mat_create(2, 2, np.array([0, 1, 2], np.int32), np.array([0, 5], np.int32), np.ones(2))
    with pytest.raises(PhipError):  # decreasing row pointer
        ctx.mat_create(2, 2, np.array([0, 2, 1], np.int32), np.array([0, 1], np.int32), np.ones(2))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 255, 256, 257, 1000, 65537, 1 << 20])
def test_vec_ops(ctx, orc, n):
    L = ctx.lib
    rng = np.random.default_rng(n)
    x, y, w = rng.standard_normal(n), rng.standard_normal(n), rng.standard_normal(n)
    dx, dy, dw = ctx.to_device(x if n else np.zeros(1)), ctx.to_device(y if n else np.zeros(1)), ctx.to_device(w if n else np.zeros(1))

    def reset():
        if n:
            ctx.upload(dx, x); ctx.upload(dy, y); ctx.upload(dw, w)

    def tol(ref):
        return 1e-13 * max(1.0, np.abs(ref).sum() if np.ndim(ref) == 0 else 1.0)

    # reductions
    s = np.abs(x * y).sum() + 1e-300
    assert abs(ctx.dot(n, dx, dy) - orc.dot(x, y)) <= 1e-13 * s
    assert abs(ctx.norm(n, dx, 1) - orc.norm(x, 1)) <= 1e-13 * (np.linalg.norm(x) + 1e-300)
    assert abs(ctx.norm(n, dx, 0) - orc.norm(x, 0)) <= 1e-13 * (np.abs(x).sum() + 1e-300)
    assert ctx.norm(n, dx, 3) == orc.norm(x, 3)
    n1, n2 = ctx.norm(n, dx, 4)
    assert abs(n1 - orc.norm(x, 0)) <= 1e-13 * (np.abs(x).sum() + 1e-300) and abs(n2 - orc.norm(x, 1)) <= 1e-13 * (np.linalg.norm(x) + 1e-300)
    # elementwise, including the special-cased scalars of the reference
    for alpha in (0.0, 1.0, -1.0, 0.37):
        reset(); ctx.check(L.phip_vec_axpy(ctx.h, n, dy, alpha, dx))
        np.testing.assert_allclose(ctx.to_host(dy, n), orc.axpy(y, alpha, x), rtol=0, atol=1e-15 * 4)
        reset(); ctx.check(L.phip_vec_aypx(ctx.h, n, dy, alpha, dx))
        np.testing.assert_allclose(ctx.to_host(dy, n), orc.aypx(y, alpha, x), rtol=0, atol=1e-15 * 4)
        reset(); ctx.check(L.phip_vec_waxpy(ctx.h, n, dw, alpha, dx, dy))
        np.testing.assert_allclose(ctx.to_host(dw, n), orc.waxpy(alpha, x, y), rtol=0, atol=1e-15 * 4)
        reset(); ctx.check(L.phip_vec_scale(ctx.h, n, dx, alpha))
        np.testing.assert_array_equal(ctx.to_host(dx, n), orc.scale(x, alpha))
        for beta in (0.0, 1.0, -2.5):
            reset(); ctx.check(L.phip_vec_axpby(ctx.h, n, dy, alpha, beta, dx))
            np.testing.assert_allclose(ctx.to_host(dy, n), orc.axpby(y, alpha, beta, x), rtol=0, atol=1e-14)
            for gamma in (0.0, 1.0, 0.5):
                reset(); ctx.check(L.phip_vec_axpbypcz(ctx.h, n, dw, alpha, beta, gamma, dx, dy))
                np.testing.assert_allclose(ctx.to_host(dw, n), orc.axpbypcz(w, alpha, beta, gamma, x, y), rtol=0, atol=1e-14)
    reset(); ctx.check(L.phip_vec_pointwise_mult(ctx.h, n, dw, dx, dy))
    np.testing.assert_array_equal(ctx.to_host(dw, n), orc.pointwisemult(x, y))
    reset(); ctx.check(L.phip_vec_pointwise_mult(ctx.h, n, dx, dx, dy))  # w aliases x (bvec2.c:94-95)
    np.testing.assert_array_equal(ctx.to_host(dx, n), orc.pointwisemult(x, y))
    reset(); ctx.check(L.phip_vec_pointwise_mult(ctx.h, n, dy, dx, dy))  # w aliases y (bvec2.c:96-97)
    np.testing.assert_array_equal(ctx.to_host(dy, n), orc.pointwisemult(x, y))
    reset(); ctx.check(L.phip_vec_pointwise_divide(ctx.h, n, dw, dx, dy))
    np.testing.assert_allclose(ctx.to_host(dw, n), orc.pointwisedivide(x, y), rtol=2e-16, atol=0)
    reset(); ctx.check(L.phip_vec_copy(ctx.h, n, dx, dw))
    np.testing.assert_array_equal(ctx.to_host(dw, n), x)
    for a in (0.0, -1.75):
        ctx.check(L.phip_vec_set(ctx.h, n, dw, a))
        np.testing.assert_array_equal(ctx.to_host(dw, n), np.full(n, a))
    reset(); ctx.check(L.phip_vec_swap(ctx.h, n, dx, dy))
    np.testing.assert_array_equal(ctx.to_host(dx, n), y); np.testing.assert_array_equal(ctx.to_host(dy, n), x)
    xz = x.copy()
    if n > 2:
        xz[1] = 0.0
    if n:
        ctx.upload(dx, xz)
    ctx.check(L.phip_vec_reciprocal(ctx.h, n, dx))
    np.testing.assert_allclose(ctx.to_host(dx, n), orc.reciprocal(xz), rtol=2e-16, atol=0)
    for p in (dx, dy, dw):
        ctx.free(p)


def test_norm_inf_nan(ctx):
    x = np.ones(1000); x[777] = np.nan
    dx = ctx.to_device(x)
    assert np.isnan(ctx.norm(1000, dx, 3))
    ctx.free(dx)


@pytest.mark.parametrize("ny,nx", [(5, 13), (1, 257), (31, 33), (41, 25), (427, 3), (1, 1)])
def test_mask_kernels_partial_last_wave(ctx, orc, ny, nx):
    """Row counts that leave the last wave (256 rows) and the last workgroup (1024 rows) of the mask kernels partly or almost wholly
    without rows: 65, 257, 1023, 1025, 1281, 1.  spmv_dmaskg_kernel writes the execution mask from scalar registers (dmg_step); lanes
    without rows must neither store nor feed the fused dot.  Both mask kernels, product / MultAdd / product with fused dot, bit for bit
    against the oracle; the entries of y behind row m (a guard band) stay untouched."""
    import ctypes as C
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket
    L = ctx.lib
    ai, aj, aa = G.laplace2d_5pt(ny, nx)
    m = ny * nx
    assert len(ai) - 1 == m
    A = ctx.mat_create(m, m, ai, aj, aa)
    du = C.c_int()
    ctx.check(L.phip_mat_diag_mask_info(A, None, C.byref(du), None))
    assert du.value == 1 or m < 64   # a matrix of a few rows may stay in the CSR layout: the results are checked all the same
    x, y0 = G.random_vector(m), np.random.default_rng(m).standard_normal(m)
    guard = 2048
    dx, dy0 = ctx.to_device(x), ctx.to_device(y0)
    ref, refadd = orc.matmult(ai, aj, aa, x), orc.matmultadd(ai, aj, aa, x, y0)
    for kind in (4, 5):
        ctx.check(L.phip_set_dmask_kernel(kind))
        try:
            dy = ctx.to_device(np.full(m + guard, 7.0))
            ctx.check(L.phip_mat_mult(A, dx, dy))
            got = ctx.to_host(dy, m + guard)
            np.testing.assert_array_equal(got[:m], ref)
            np.testing.assert_array_equal(got[m:], np.full(guard, 7.0))
            ctx.check(L.phip_mat_multadd(A, dx, dy0, dy))
            got = ctx.to_host(dy, m + guard)
            np.testing.assert_array_equal(got[:m], refadd)
            np.testing.assert_array_equal(got[m:], np.full(guard, 7.0))
            t, val = Ticket(), np.zeros(8)
            ctx.check(L.phip_mat_mult_dot(A, dx, dy, dx, 1, C.byref(t)))
            ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
            got = ctx.to_host(dy, m + guard)
            np.testing.assert_array_equal(got[:m], ref)
            np.testing.assert_array_equal(got[m:], np.full(guard, 7.0))
            assert abs(val[0] - x @ ref) <= 1e-13 * np.abs(x * ref).sum() + 1e-300
            ctx.free(dy)
        finally:
            ctx.check(L.phip_set_dmask_kernel(0))
    ctx.free(dx); ctx.free(dy0)
    ctx.check(L.phip_mat_destroy(A))


def test_mask_kernels_repeated_diagonals(ctx, orc):
    """Diagonals that carry several values each (a coefficient with period 3 along the rows): the pair table then holds several pairs on
    one diagonal, which the group descriptors must place in groups of their own (a place of the window holds one pair).  All mask
    kernels against the oracle, bit for bit, on a 2-D 9-point stencil with 5 diagonals x 3 values + boundary variants <= 32 pairs."""
    import ctypes as C
    from petsc_amd import generators as G
    L = ctx.lib
    nx, ny = 70, 53
    m = nx * ny
    rows, cols, vals = [], [], []
    for j in range(ny):
        for i in range(nx):
            r = j * nx + i
            for dj, di in ((-1, 0), (0, -1), (0, 0), (0, 1), (1, 0)):
                ii, jj = i + di, j + dj
                if 0 <= ii < nx and 0 <= jj < ny:
                    rows.append(r); cols.append(jj * nx + ii)
                    vals.append((4.0 if (di, dj) == (0, 0) else -1.0) * (1.0 + 0.25 * (r % 3)) + (0.125 if dj else 0.0))
    import scipy.sparse as sp
    A0 = sp.csr_matrix((vals, (rows, cols)), shape=(m, m))
    A0.sort_indices()
    ai, aj, aa = A0.indptr.astype(np.int32), A0.indices.astype(np.int32), A0.data.astype(np.float64)
    A = ctx.mat_create(m, m, ai, aj, aa)
    db, du, dn = C.c_int(), C.c_int(), C.c_int32()
    ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
    assert (db.value, du.value) == (1, 1) and 12 <= dn.value <= 32, dn.value     # 5 diagonals x 3 values
    x, y0 = G.random_vector(m), np.random.default_rng(3).standard_normal(m)
    dx, dy, dy0 = ctx.to_device(x), ctx.to_device(np.zeros(m)), ctx.to_device(y0)
    ref = orc.matmult(ai, aj, aa, x)
    for kind in (4, 5):
        ctx.check(L.phip_set_dmask_kernel(kind))
        try:
            ctx.check(L.phip_mat_mult(A, dx, dy))
            np.testing.assert_array_equal(ctx.to_host(dy, m), ref)
            ctx.check(L.phip_mat_multadd(A, dx, dy0, dy))
            np.testing.assert_array_equal(ctx.to_host(dy, m), orc.matmultadd(ai, aj, aa, x, y0))
        finally:
            ctx.check(L.phip_set_dmask_kernel(0))
    for p_ in (dx, dy, dy0):
        ctx.free(p_)
    ctx.check(L.phip_mat_destroy(A))


def test_reductions_under_foreign_load(ctx):
    """The single-launch reductions let their last workgroup wait for the partial results of the others (phip_reduce.h: workgroups are
    dispatched in index order and at most 1024 of them exist, so all are resident).  Here that is exercised while ANOTHER PROCESS keeps
    the same GPU busy with bandwidth-bound launches: every result must arrive (no error from the fetching call, which a collector
    that gave up would cause through the context's error word), be finite, and equal the result obtained on the idle device bit for bit."""
    import subprocess
    import sys
    import time
    here = os.path.dirname(os.path.abspath(__file__))
    rng = np.random.default_rng(77)
    sizes = [1, 1000, 300001, 2097153, 8388609]
    data = {n: (rng.standard_normal(n), rng.standard_normal(n)) for n in sizes}
    dev = {n: (ctx.to_device(a), ctx.to_device(b)) for n, (a, b) in data.items()}
    idle = {n: (ctx.dot(n, dx, dy), ctx.norm(n, dx, 1), ctx.norm(n, dx, 3), ctx.mdot(n, dx, [dy, dx, dy])) for n, (dx, dy) in dev.items()}
    load = subprocess.Popen([sys.executable, os.path.join(here, "_gpu_load_worker.py"), "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        assert "load running" in load.stdout.readline()
        t0, rounds = time.time(), 0
        while time.time() - t0 < 2.5:
            for n, (dx, dy) in dev.items():
                got = (ctx.dot(n, dx, dy), ctx.norm(n, dx, 1), ctx.norm(n, dx, 3), ctx.mdot(n, dx, [dy, dx, dy]))
                assert got[0] == idle[n][0] and got[1] == idle[n][1] and got[2] == idle[n][2] and np.array_equal(got[3], idle[n][3]), (n, rounds)
            rounds += 1
        assert rounds >= 3
    finally:
        out, _ = load.communicate(timeout=60)
    assert load.returncode == 0 and "load done" in out, out[-2000:]
    for dx, dy in dev.values():
        ctx.free(dx); ctx.free(dy)


@pytest.mark.parametrize("n", [0, 5, 1003, 300001])
def test_mdot_maxpy(ctx, orc, n):
    """ex43.c-style: fused VecMDot equals per-vector VecDot; VecMAXPY equals the oracle's grouped sum."""
    rng = np.random.default_rng(n + 1)
    x = rng.standard_normal(n)
    dx = ctx.to_device(x if n else np.zeros(1))
    for nv in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 30, 31):
        ys = [rng.standard_normal(n) for _ in range(nv)]
        dys = [ctx.to_device(y if n else np.zeros(1)) for y in ys]
        z = ctx.mdot(n, dx, dys)
        zo = orc.mdot(x, ys)
        scale = max(1.0, max(np.abs(x * y).sum() for y in ys)) if n else 1.0
        assert np.abs(z - zo).max() <= 1e-13 * scale
        for k in range(nv):
            assert abs(z[k] - ctx.dot(n, dx, dys[k])) <= 1e-13 * scale
        al = rng.standard_normal(nv)
        if n:
            ctx.upload(dx, x)
        ctx.maxpy(n, dx, al, dys)
        ref = orc.maxpy(x, al, ys)
        assert np.abs(ctx.to_host(dx, n) - ref).max() <= 1e-13 * (1 + np.abs(al).sum() * 6) if n else True
        if n:
            ctx.upload(dx, x)
        for p in dys:
            ctx.free(p)
    ctx.free(dx)


def test_jacobi_setup(ctx, orc):
    d = np.array([2.0, 0.0, -4.0, 1e-300, 5.0])
    for useabs in (0, 1):
        dd = ctx.to_device(d)
        import ctypes
        flag = ctypes.c_int(-1)
        ctx.check(ctx.lib.phip_jacobi_setup(ctx.h, len(d), dd, useabs, 0, ctypes.byref(flag)))
        np.testing.assert_array_equal(ctx.to_host(dd, len(d)), orc.jacobi_setup(d, bool(useabs)))
        assert flag.value == 1
        ctx.free(dd)


def test_self_comm_and_reduce_local(ctx):
    import ctypes as C
    comm = C.c_void_p()
    ctx.check(ctx.lib.phip_comm_create(ctx.h, b"self", None, 1, 0, C.byref(comm)))
    x = np.arange(1, 101, dtype=float)
    dx = ctx.to_device(x)
    dout = ctx.malloc(8 * 4)
    ctx.check(ctx.lib.phip_vec_reduce_local(ctx.h, 1, 100, dx, 1, None, dout))
    ctx.check(ctx.lib.phip_comm_allreduce(comm, dout, 1, 0))
    z = (C.c_double * 1)()
    ctx.check(ctx.lib.phip_read_scalars(ctx.h, dout, 1, z))
    assert z[0] == float((x * x).sum())
    ctx.check(ctx.lib.phip_comm_destroy(comm))
    ctx.free(dx); ctx.free(dout)


def test_rccl_single_rank(ctx):
    """RCCL smoke on one GPU: unique id, init, all-reduce of one rank, a self send/recv ghost exchange."""
    import ctypes as C
    idb = C.create_string_buffer(128)
    ctx.check(ctx.lib.phip_comm_unique_id(idb))
    comm = C.c_void_p()
    ctx.check(ctx.lib.phip_comm_create(ctx.h, b"rccl", idb, 1, 0, C.byref(comm)))
    dbuf = ctx.to_device(np.array([1.5, -2.0, 3.25]))
    ctx.check(ctx.lib.phip_comm_allreduce(comm, dbuf, 3, 0))
    np.testing.assert_array_equal(ctx.to_host(dbuf, 3), [1.5, -2.0, 3.25])
    gathered = np.zeros(1, np.int32)
    mine = np.array([42], np.int32)
    ctx.check(ctx.lib.phip_comm_allgather_host(comm, mine.ctypes.data_as(C.c_void_p), gathered.ctypes.data_as(C.c_void_p), 4))
    assert gathered[0] == 42
    ctx.check(ctx.lib.phip_comm_destroy(comm))
    ctx.free(dbuf)


def test_byproduct_tickets(ctx, orc):
    """Reductions fused into the producing kernel: w = x.*y with {sum w^2, sum w x, sum w y}; y = A x with sum v_i y_i; the
    off-diagonal block's MultAdd adding sum v_i (increment)_i to the same ticket; expiry of old tickets."""
    import ctypes as C
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket, PhipError
    L = ctx.lib
    rng = np.random.default_rng(21)
    for n in (0, 1, 7, 1000, 300001):
        x, y = rng.standard_normal(n), rng.standard_normal(n)
        dx, dy, dw = ctx.to_device(x if n else np.zeros(1)), ctx.to_device(y if n else np.zeros(1)), ctx.to_device(np.zeros(max(n, 1)))
        t = Ticket()
        ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, n, dw, dx, dy, 1, C.byref(t)))
        vals = np.zeros(3)
        ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), vals.ctypes.data_as(C.POINTER(C.c_double))))
        w = x * y
        np.testing.assert_array_equal(ctx.to_host(dw, n), orc.pointwisemult(x, y))
        ref = np.array([w @ w, w @ x, w @ y])
        assert np.abs(vals - ref).max() <= 1e-13 * max(1.0, np.abs(w * w).sum(), np.abs(w * x).sum(), np.abs(w * y).sum())
        # the same values through the device slot (the multi-rank consumption path)
        dv = C.c_void_p()
        ctx.check(L.phip_ticket_values(ctx.h, C.byref(t), C.byref(dv)))
        np.testing.assert_array_equal(ctx.to_host(dv, 3), vals)
        with pytest.raises(PhipError):  # aliasing is refused: the plain kernel handles it
            ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, n or 1, dx, dx, dy, 1, C.byref(t)))
        for p in (dx, dy, dw):
            ctx.free(p)
    cases = [G.laplace3d_7pt(24), G.box3d_27pt(13), G.rmat(12), G.laplace2d_5pt(3, 2)]
    for ai, aj, aa in cases:
        n = len(ai) - 1
        x, v = G.random_vector(n), rng.standard_normal(n)
        for fmt in (1, 2):
            A = ctx.mat_create(n, n, ai, aj, aa)
            if L.phip_mat_set_format(A, fmt) != 0:
                ctx.check(L.phip_mat_destroy(A))
                continue
            dx, dv, dy = ctx.to_device(x), ctx.to_device(v), ctx.to_device(np.zeros(n))
            t = Ticket()
            ctx.check(L.phip_mat_mult_dot(A, dx, dy, dv, 1, C.byref(t)))
            val = np.zeros(1)
            ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
            yo = orc.matmult(ai, aj, aa, x)
            assert relerr(ctx.to_host(dy, n), yo) <= 1e-12
            assert abs(val[0] - v @ yo) <= 1e-12 * np.abs(v * yo).sum()
            # MultAdd in place adds its increment's share to the same ticket (no publish: read the device slot)
            t2 = Ticket()
            ctx.check(L.phip_mat_mult_dot(A, dx, dy, dv, 0, C.byref(t2)))
            ctx.check(L.phip_mat_multadd_dot(A, dx, dy, dv, 0, C.byref(t2)))
            dvp = C.c_void_p()
            ctx.check(L.phip_ticket_values(ctx.h, C.byref(t2), C.byref(dvp)))
            assert abs(ctx.to_host(dvp, 1)[0] - 2.0 * (v @ yo)) <= 1e-11 * np.abs(v * yo).sum()
            assert relerr(ctx.to_host(dy, n), 2.0 * yo) <= 1e-12
            # ... and with publish the accumulated value reaches the ticket's host slot (the multi-rank path: diag block, then off-diag)
            t3 = Ticket()
            ctx.check(L.phip_mat_mult_dot(A, dx, dy, dv, 0, C.byref(t3)))
            ctx.check(L.phip_mat_multadd_dot(A, dx, dy, dv, 1, C.byref(t3)))
            ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t3), val.ctypes.data_as(C.POINTER(C.c_double))))
            assert abs(val[0] - 2.0 * (v @ yo)) <= 1e-11 * np.abs(v * yo).sum()
            for p in (dx, dv, dy):
                ctx.free(p)
            ctx.check(L.phip_mat_destroy(A))
    # x += sum a_k y_k with {sum x_new^2}: same x bits as phip_vec_maxpy, every instantiation and the split above 16 vectors
    for n, nv in [(0, 1), (1, 1), (7, 3), (1000, 2), (1001, 16), (300001, 5), (300001, 31), (2097153, 17)]:
        x0 = rng.standard_normal(n)
        ys = [rng.standard_normal(n) for _ in range(nv)]
        al = rng.standard_normal(nv)
        dys = [ctx.to_device(y if n else np.zeros(1)) for y in ys]
        arr = (C.c_void_p * nv)(*dys)
        alp = al.ctypes.data_as(C.POINTER(C.c_double))
        dx1, dx2 = ctx.to_device(x0 if n else np.zeros(1)), ctx.to_device(x0 if n else np.zeros(1))
        ctx.check(L.phip_vec_maxpy(ctx.h, n, dx1, nv, alp, arr))
        t = Ticket()
        ctx.check(L.phip_vec_maxpy_reduce(ctx.h, n, dx2, nv, alp, arr, 1, C.byref(t)))
        val = np.zeros(1)
        ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), val.ctypes.data_as(C.POINTER(C.c_double))))
        x1 = ctx.to_host(dx1, n)
        np.testing.assert_array_equal(ctx.to_host(dx2, n), x1)
        if n:
            assert np.abs(x1 - orc.maxpy(x0, al, ys)).max() <= 1e-13 * (1 + np.abs(al).sum() * 6)
        assert abs(val[0] - x1 @ x1) <= 1e-13 * max(1.0, x1 @ x1), (n, nv)
        for p_ in dys + [dx1, dx2]:
            ctx.free(p_)
    t = Ticket()
    assert L.phip_vec_maxpy_reduce(ctx.h, 8, None, 0, None, None, 1, C.byref(t)) != 0   # needs at least one vector
    # a ticket expires after 63 newer ones
    dx = ctx.to_device(np.ones(8)); dw = ctx.to_device(np.zeros(8))
    dy = ctx.to_device(np.full(8, 2.0))
    first = Ticket()
    ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, 8, dw, dx, dy, 1, C.byref(first)))
    t = Ticket()
    for _ in range(64):
        ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, 8, dw, dx, dy, 1, C.byref(t)))
    vals = np.zeros(3)
    assert L.phip_ticket_wait(ctx.h, C.byref(first), vals.ctypes.data_as(C.POINTER(C.c_double))) == 73
    ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), vals.ctypes.data_as(C.POINTER(C.c_double))))
    np.testing.assert_array_equal(vals, [32.0, 16.0, 32.0])
    for p in (dx, dy, dw):
        ctx.free(p)


@pytest.mark.parametrize("n", [1, 2, 1001, 300000, 2097153, 8388609])   # the last: 64 MiB vectors, the non-temporal loads and stores
def test_cg_update_batch(orc, n):
    """VecAXPY(X,a,P); VecAXPY(R,b,W); Z = D.*R (or R.*D) with the three sums -- the batch of a KSPCG iteration -- runs in a kernel of its
    own (operands requested together, the new R handed over in a register): same bits as one launch per operation, sums within round-off;
    a batch that only looks similar (an operand shared between the operations) takes the generic kernel and gives the same."""
    import ctypes as C
    from petsc_amd import capi
    from petsc_amd.capi import Ticket
    rng = np.random.default_rng(9)
    x0, p0, r0, w0, d0 = (rng.standard_normal(n) for _ in range(5))

    def run(c):
        L = c.lib
        out = []
        for variant in range(4):
            dx, dp, dr, dw, dd = (c.to_device(v) for v in (x0, p0, r0, w0, d0))
            dz = dw if variant == 3 else c.to_device(np.zeros(n))                        # variant 3: Z is W, as in KSPCG (cg.c:106)
            c.check(L.phip_vec_axpy(c.h, n, dx, 0.375, dp))
            c.check(L.phip_vec_axpy(c.h, n, dr, -0.375, dp if variant == 2 else dw))      # variant 2: P appears twice -> generic kernel
            t = Ticket()
            if variant == 1:
                c.check(L.phip_vec_pointwise_mult_reduce(c.h, n, dz, dr, dd, 1, C.byref(t)))   # Z = R .* D
            else:
                c.check(L.phip_vec_pointwise_mult_reduce(c.h, n, dz, dd, dr, 1, C.byref(t)))   # Z = D .* R
            red = np.zeros(3)
            c.check(L.phip_ticket_wait(c.h, C.byref(t), red.ctypes.data_as(C.POINTER(C.c_double))))
            out.append([c.to_host(v, n) for v in (dx, dr, dz)] + [red])
            for v in (dx, dp, dr, dw, dd) + (() if variant == 3 else (dz,)):
                c.free(v)
        return out

    c1 = capi.Context(0)
    got = run(c1)
    c1.close()
    os.environ["PHIP_EW_BATCH"] = "0"
    try:
        c0 = capi.Context(0)
    finally:
        del os.environ["PHIP_EW_BATCH"]
    ref = run(c0)
    c0.close()
    for variant, (g, r) in enumerate(zip(got, ref)):
        for a, b in zip(g[:3], r[:3]):
            np.testing.assert_array_equal(a, b)
        x = x0 + 0.375 * p0
        rr = r0 - 0.375 * (p0 if variant == 2 else w0)
        z = d0 * rr
        f, gg = (rr, d0) if variant == 1 else (d0, rr)
        assert np.abs(g[0] - x).max() <= 1e-15 * max(1.0, np.abs(x).max()) * 4
        assert np.abs(g[2] - z).max() <= 1e-14 * max(1.0, np.abs(z).max())
        exact = np.array([z @ z, z @ f, z @ gg])
        scale = max(1.0, (z * z).sum(), np.abs(z * f).sum(), np.abs(z * gg).sum())
        assert np.abs(g[3] - exact).max() <= 1e-12 * scale and np.abs(g[3] - r[3]).max() <= 1e-12 * scale


@pytest.mark.parametrize("n", [1, 2, 1001, 300000, 8388609])
def test_cg_deferred_x_update(ctx, n):
    """KSPCG's VecAXPY(X,a,P) stays pending when the rest of its batch leaves with the reducing product, and is launched with the
    VecAYPX(P,b,Z) that follows in one kernel reading P once (xp_update_kernel): two iterations' worth of calls give the same bits in
    X, P, R, Z and the same sums as with the deferral switched off; a read of X in between sees the updated X."""
    import ctypes as C
    from petsc_amd.capi import Ticket
    L = ctx.lib
    rng = np.random.default_rng(5)
    x0, p0, r0, w0, d0 = (rng.standard_normal(n) for _ in range(5))
    res = {}
    for on in (1, 0):
        ctx.check(L.phip_set_cg_defer_x(on))
        try:
            dx, dp, dr, dw, dd = (ctx.to_device(v) for v in (x0, p0, r0, w0, d0))
            sums, peek = [], None
            for it in range(2):
                ctx.check(L.phip_vec_axpy(ctx.h, n, dx, 0.375 + it, dp))          # X += a P
                ctx.check(L.phip_vec_axpy(ctx.h, n, dr, -0.375 - it, dw))         # R -= a W
                t = Ticket()
                ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, n, dw, dd, dr, 1, C.byref(t)))   # Z = D .* R, Z is W (cg.c:106)
                red = np.zeros(3)
                ctx.check(L.phip_ticket_wait(ctx.h, C.byref(t), red.ctypes.data_as(C.POINTER(C.c_double))))
                sums.append(red)
                if it == 0:
                    ctx.check(L.phip_vec_aypx(ctx.h, n, dp, 0.625, dw))          # P = Z + b P: leaves together with the pending X update
                    ctx.check(L.phip_vec_scale(ctx.h, n, dw, 2.0))               # another operation behind them
                else:
                    peek = ctx.to_host(dx, n)                                    # the caller looks at X while its update is pending
            res[on] = [ctx.to_host(v, n) for v in (dx, dp, dr, dw)] + [peek] + sums
            for v in (dx, dp, dr, dw, dd):
                ctx.free(v)
        finally:
            ctx.check(L.phip_set_cg_defer_x(1))
    for a, b in zip(res[1], res[0]):
        np.testing.assert_array_equal(a, b)
    x1 = x0 + 0.375 * p0
    p1 = d0 * (r0 - 0.375 * w0) + 0.625 * p0
    assert np.abs(res[1][0] - (x1 + 1.375 * p1)).max() <= 1e-14 * max(1.0, np.abs(x1).max() + 2 * np.abs(p1).max())
    np.testing.assert_array_equal(res[1][4], res[1][0])


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 300001])
def test_batched_elementwise_ops(orc, n):
    """Elementwise operations wait in the context and leave in one launch (up to 4, call order kept per entry).  Chains in which each
    operation reads what the previous one wrote, operands that alias, sub-vectors that overlap without being equal (must not share a
    batch), different lengths, the reducing pointwise product joining a batch: same bits as unbatched execution (PHIP_EW_BATCH=0),
    oracle within round-off."""
    import ctypes as C
    from petsc_amd import capi
    from petsc_amd.capi import Ticket
    rng = np.random.default_rng(77)
    x0, y0, z0, w0 = (rng.standard_normal(n) for _ in range(4))

    def run(c):
        L = c.lib
        dx, dy, dz, dw = c.to_device(x0), c.to_device(y0), c.to_device(z0), c.to_device(w0)
        big = c.to_device(np.arange(2.0 * n + 8))
        c.check(L.phip_vec_axpy(c.h, n, dy, 0.5, dx))              # y += 0.5 x
        c.check(L.phip_vec_pointwise_mult(c.h, n, dw, dx, dy))     # w = x .* y        (reads the new y)
        c.check(L.phip_vec_aypx(c.h, n, dz, -2.0, dw))             # z = w - 2 z       (reads the new w)
        c.check(L.phip_vec_axpbypcz(c.h, n, dx, 2.0, 3.0, 0.5, dz, dw))  # x = 2 z + 3 w + 0.5 x
        c.check(L.phip_vec_scale(c.h, n, dx, 1.25))                # fifth operation: the first four leave before it
        c.check(L.phip_vec_waxpy(c.h, n, dw, -1.0, dx, dx))        # w = x - x = 0     (x twice, w rewritten)
        c.check(L.phip_vec_pointwise_mult(c.h, n, dy, dy, dy))     # y = y .* y        (all three alias)
        t = Ticket()
        c.check(L.phip_vec_pointwise_mult_reduce(c.h, n, dw, dy, dz, 1, C.byref(t)))  # joins the batch: w = y .* z + sums
        red = np.zeros(3)
        c.check(L.phip_ticket_wait(c.h, C.byref(t), red.ctypes.data_as(C.POINTER(C.c_double))))
        # overlapping, unequal operands: big[0:n] and big[1:n+1]
        b1 = C.c_void_p(big.value + 8)
        c.check(L.phip_vec_axpy(c.h, n, big, 1.0, dz))             # big[0:n] += z
        c.check(L.phip_vec_axpy(c.h, n, b1, 1.0, dz))              # big[1:n+1] += z   (must not run concurrently with the line above)
        c.check(L.phip_vec_scale(c.h, n + 3, big, 2.0))            # another length
        out = [c.to_host(p, n) for p in (dx, dy, dz, dw)] + [c.to_host(big, 2 * n + 8), red]
        for p in (dx, dy, dz, dw, big):
            c.free(p)
        return out

    c1 = capi.Context(0)
    got = run(c1)
    c1.close()
    os.environ["PHIP_EW_BATCH"] = "0"
    try:
        c0 = capi.Context(0)
    finally:
        del os.environ["PHIP_EW_BATCH"]
    ref = run(c0)
    c0.close()
    for a, b in zip(got[:5], ref[:5]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_allclose(got[5], ref[5], rtol=1e-13)
    # and against numpy
    x, y, z, w = x0.copy(), y0.copy(), z0.copy(), w0.copy()
    y = y + 0.5 * x; w = x * y; z = w - 2.0 * z; x = 2.0 * z + 3.0 * w + 0.5 * x; x = 1.25 * x; w = x - x; y = y * y; w = y * z
    for a, b in zip(got[:4], (x, y, z, w)):
        assert np.abs(a - b).max() <= 1e-13 * max(1.0, np.abs(b).max())
    big = np.arange(2.0 * n + 8); big[:n] += z; big[1:n + 1] += z; big[:n + 3] *= 2.0
    np.testing.assert_array_equal(got[4], big)
    assert np.abs(got[5] - np.array([w @ w, w @ y, w @ z])).max() <= 1e-12 * max(1.0, (w * w).sum(), np.abs(w * y).sum(), np.abs(w * z).sum())


def test_device_side_value_updates(ctx, orc):
    """phip_mat_scale / phip_mat_shift / phip_mat_diagonal_scale / phip_mat_zero_entries (MatScale_SeqAIJ aij.c:2475-2488, MatShift_Basic
    gcreate.c:6-24, MatDiagonalScale_SeqAIJ aij.c:2200-2235, MatZeroEntries_SeqAIJ aij.c:1058-1067) change the values where they live, on
    the device, in every layout at once: the CSR values read back are bit-identical to the host loops' (a * alpha; a(i,i) + alpha;
    (a * l_i) * r_j), every product afterwards equals the product of a matrix created from those values (bit for bit), the pair
    dictionary and the diagonal masks survive scaling and shifting, and a diagonal scaling falls back to streamed values."""
    import ctypes as C
    from petsc_amd import generators as G
    L = ctx.lib
    rng = np.random.default_rng(31)

    def flags(A):
        f, vb, vu, db, du = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
        ctx.check(L.phip_mat_format_info(A, C.byref(f), None, None))
        ctx.check(L.phip_mat_value_dict_info(A, C.byref(vb), C.byref(vu), None))
        ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), None))
        return f.value, vu.value, du.value

    def products(A, m, n, x, v):
        dx, dv, dy, dt = ctx.to_device(x), ctx.to_device(v), ctx.to_device(np.zeros(m)), ctx.to_device(np.zeros(n))
        ctx.check(L.phip_mat_mult(A, dx, dy))
        ctx.check(L.phip_mat_mult_transpose(A, dv, dt))
        out = (ctx.to_host(dy, m), ctx.to_host(dt, n))
        for p in (dx, dv, dy, dt):
            ctx.free(p)
        return out

    def fresh(ci, cj, ca, n, x, v):
        B = ctx.mat_create(len(ci) - 1, n, ci, cj, ca)
        ctx.check(L.phip_mat_transpose_setup(B, ci.ctypes.data_as(C.POINTER(C.c_int32)), cj.ctypes.data_as(C.POINTER(C.c_int32)), ca.ctypes.data_as(C.POINTER(C.c_double))))
        out = products(B, len(ci) - 1, n, x, v)
        ctx.check(L.phip_mat_destroy(B))
        return out

    ri, rj, ra = G.rmat(11)
    cases = [("7-point, diagonal masks", *G.laplace3d_7pt(14, 11, 9, dirichlet_rows=False), (2, 1, 1)),
             ("convection-diffusion, diagonal masks", *G.convdiff3d_7pt(10, 9, 8), (2, 1, 1)),
             ("R-MAT, CSR row blocks", ri, rj, ra, (1, 0, 0))]
    si, sj, sa = G.laplace3d_7pt(12, dirichlet_rows=False)
    cases.append(("7-point with values of its own, SELL-dict streamed", si, sj, sa * rng.uniform(0.5, 1.5, len(sa)), (2, 0, 0)))
    for name, ci, cj, ca, want in cases:
        ci, cj, ca = np.ascontiguousarray(ci, np.int32), np.ascontiguousarray(cj, np.int32), np.ascontiguousarray(ca, np.float64)
        m = n = len(ci) - 1
        rows = np.repeat(np.arange(m), np.diff(ci))
        x, v, l, r = rng.standard_normal(n), rng.standard_normal(m), rng.uniform(0.5, 2.0, m), rng.uniform(0.5, 2.0, n)
        A = ctx.mat_create(m, n, ci, cj, ca)
        ctx.check(L.phip_mat_transpose_setup(A, ci.ctypes.data_as(C.POINTER(C.c_int32)), cj.ctypes.data_as(C.POINTER(C.c_int32)), ca.ctypes.data_as(C.POINTER(C.c_double))))
        assert flags(A) == want, (name, flags(A))
        cur = ca.copy()
        got = np.zeros_like(cur)

        def check(tag, keep_layout):
            ctx.check(L.phip_mat_get_values(A, got.ctypes.data_as(C.POINTER(C.c_double))))
            np.testing.assert_array_equal(got, cur, err_msg="%s: %s" % (name, tag))
            if keep_layout:
                assert flags(A) == want, (name, tag, flags(A))
            y, t = products(A, m, n, x, v)
            yf, tf = fresh(ci, cj, cur, n, x, v)
            np.testing.assert_array_equal(y, yf, err_msg="%s: %s" % (name, tag))
            np.testing.assert_array_equal(t, tf, err_msg="%s: %s (transpose)" % (name, tag))
            assert relerr(y, orc.matmult(ci, cj, cur, x)) <= 1e-12

        ctx.check(L.phip_mat_scale(A, 2.5)); cur = cur * 2.5
        check("scale", True)
        if np.all(np.isin(np.arange(m), cj[cj == rows])):          # every row stores its diagonal entry
            ctx.check(L.phip_mat_shift(A, -0.75)); cur[cj == rows] = cur[cj == rows] + -0.75
            check("shift", True)
        dl, dr = ctx.to_device(l), ctx.to_device(r)
        ctx.check(L.phip_mat_diagonal_scale(A, dl, dr)); cur = (cur * l[rows]) * r[cj]
        check("diagonal scale", False)
        assert flags(A)[1:] == (0, 0)                               # position-dependent values: no dictionary any more
        ctx.check(L.phip_mat_diagonal_scale(A, None, dr)); cur = cur * r[cj]
        ctx.check(L.phip_mat_diagonal_scale(A, dl, None)); cur = cur * l[rows]
        check("one-sided diagonal scales", False)
        ctx.check(L.phip_mat_update_values(A, ca.ctypes.data_as(C.POINTER(C.c_double)))); cur = ca.copy()   # the host route brings the dictionary back
        check("update_values", True)
        ctx.check(L.phip_mat_zero_entries(A)); cur = np.zeros_like(cur)
        check("zero entries", False)
        ctx.free(dl); ctx.free(dr)
        ctx.check(L.phip_mat_destroy(A))

"""GPU tests of code paths that only sizes of the BASELINE benchmarks reach WITH DEFAULT SETTINGS (no knob forced): the automatic
hot-column rule of the CSR row-block layout (x of 48 MiB and more, skewed column use), the non-temporal matrix stream of
spmv_rowblock_kernel (matrices over 192 MB), the non-temporal loads / stores of the elementwise kernels (vectors of 48 MiB and more),
and BASELINE config 5 as written: an R-MAT matrix in the PetscBinary format, MatLoad -> MATMPIAIJHIP on 8 ranks -> MatMult.
Each default path is compared with the same call with the feature switched off (array-equal: the features never change a bit) and
with a strided sample of rows / entries computed on the host (bars of BASELINE.json: SpMV <= 1e-12 relative, index work bit-exact)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_host_gpu import run_ranks  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from petsc_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def skewed_matrix(m, n, per_row, seed):
    """m x n, per_row entries a row; column use follows a power law (u^4) and the often-used columns are scattered over the whole range"""
    rng = np.random.default_rng(seed)
    k = (n * rng.random((m, per_row)) ** 4).astype(np.int64)
    cols = (k * 2654435761 + 12345) % n            # scatter the hot columns
    cols.sort(axis=1)
    dup = cols[:, 1:] == cols[:, :-1]
    cols[:, 1:][dup] = np.minimum(cols[:, 1:][dup] + 1, n - 1)   # (rare) repeated columns: move on by one; a survivor is harmless
    cols.sort(axis=1)
    ai = (np.arange(m + 1, dtype=np.int64) * per_row).astype(np.int32)
    return ai, cols.reshape(-1).astype(np.int32), rng.uniform(-1.0, 1.0, m * per_row)


def test_default_hot_columns_and_nontemporal_csr(ctx):
    """n = 6.5 M columns (x = 52 MB >= 48 MiB) and 18.2 M nonzeros (218 MB > 192 MB): phip_mat_create builds the hot-column order by its
    own rule and the product streams the matrix with non-temporal loads, both by default."""
    L = ctx.lib
    m, n, per_row = 1_400_000, 6_500_000, 13
    ai, aj, aa = skewed_matrix(m, n, per_row, 5)
    assert 8 * n >= (48 << 20) and 12 * len(aa) > (192 << 20)
    A = ctx.mat_create(m, n, ai, aj, aa)
    built, used, s0, s1 = C.c_int(), C.c_int(), C.c_double(), C.c_double()
    ctx.check(L.phip_mat_hot_columns_info(A, C.byref(built), C.byref(used), C.byref(s0), C.byref(s1)))
    assert built.value == 1 and used.value == 1, "the automatic rule did not build the hot-column order (shares %.3f -> %.3f)" % (s0.value, s1.value)
    assert s1.value - s0.value >= 0.04
    f = C.c_int()
    ctx.check(L.phip_mat_format_info(A, C.byref(f), None, None))
    assert f.value == 1                                             # CSR row blocks: not a stencil matrix
    x = np.random.default_rng(6).uniform(-1.0, 1.0, n)
    y0 = np.random.default_rng(7).uniform(-1.0, 1.0, m)
    dx, dy, dy0 = ctx.to_device(x), ctx.to_device(np.zeros(m)), ctx.to_device(y0)
    res = {}
    for name, hot, nt in (("default", 1, -1), ("hot off", 0, -1), ("cached loads", 1, 0), ("both off", 0, 0)):
        ctx.check(L.phip_mat_set_hot_columns(A, hot))
        ctx.check(L.phip_mat_set_nontemporal(A, nt))
        ctx.check(L.phip_mat_mult(A, dx, dy))
        mult = ctx.to_host(dy, m)
        ctx.check(L.phip_mat_multadd(A, dx, dy0, dy))
        res[name] = (mult, ctx.to_host(dy, m))
    for name in ("hot off", "cached loads", "both off"):
        assert np.array_equal(res[name][0], res["default"][0]) and np.array_equal(res[name][1], res["default"][1]), name
    rows = np.arange(0, m, 997)
    ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], x[aj[ai[r]:ai[r + 1]]]) for r in rows])
    scale = np.abs(ref).max()
    assert np.abs(res["default"][0][rows] - ref).max() <= 1e-12 * scale
    assert np.abs(res["default"][1][rows] - (ref + y0[rows])).max() <= 1e-12 * (scale + 1.0)
    hist = (C.c_int64 * 8)()
    ctx.check(L.phip_mat_plan_lanes(A, hist))
    assert sum(hist) > 0 and hist[7] == 0
    for p in (dx, dy, dy0):
        ctx.free(p)
    ctx.check(L.phip_mat_destroy(A))


def test_default_group_descriptor_mask_kernel(ctx):
    """From 4 Mi rows on the diagonal-mask product runs spmv_dmaskg_kernel (group descriptors, transposed masks as execution masks) by
    default: 7-point 162^3 (4.25 M rows) against the per-row bit-set kernel forced on the same matrix (array-equal; the fused dot
    within round-off) and a strided sample of rows computed on the host."""
    from petsc_amd import generators as G
    from petsc_amd.capi import Ticket
    L = ctx.lib
    nn = 162
    m = nn ** 3
    assert m >= (4 << 20)
    ai, aj, aa = G.stencil_rows_native(7, nn, nn, nn, 0, m)
    A = ctx.mat_create(m, m, ai, aj, aa)
    db, du, dn = C.c_int(), C.c_int(), C.c_int32()
    ctx.check(L.phip_mat_diag_mask_info(A, C.byref(db), C.byref(du), C.byref(dn)))
    assert (db.value, du.value) == (1, 1) and dn.value <= 32
    ns, nw = C.c_int64(), C.c_int64()
    ctx.check(L.phip_mat_mask_sets_info(A, C.byref(ns), C.byref(nw)))
    assert nw.value == 4 * ((m + 1023) // 1024) and 1 <= ns.value <= nw.value // 10   # the lines of the grid share few sets of mask words
    x = np.random.default_rng(8).uniform(-1.0, 1.0, m)
    y0 = np.random.default_rng(9).uniform(-1.0, 1.0, m)
    dx, dy, dy0 = ctx.to_device(x), ctx.to_device(np.zeros(m)), ctx.to_device(y0)
    res = {}
    for kind in (0, 4, 5):
        ctx.check(L.phip_set_dmask_kernel(kind))
        try:
            ctx.check(L.phip_mat_mult(A, dx, dy))
            mult = ctx.to_host(dy, m)
            ctx.check(L.phip_mat_multadd(A, dx, dy0, dy))
            madd = ctx.to_host(dy, m)
            tk, val = Ticket(), np.zeros(1)
            ctx.check(L.phip_mat_mult_dot(A, dx, dy, dx, 1, C.byref(tk)))
            ctx.check(L.phip_ticket_wait(ctx.h, C.byref(tk), val.ctypes.data_as(C.POINTER(C.c_double))))
            res[kind] = (mult, madd, ctx.to_host(dy, m), val[0])
        finally:
            ctx.check(L.phip_set_dmask_kernel(0))
    for kind in (4, 5):
        for k in range(3):
            assert np.array_equal(res[kind][k], res[0][k]), (kind, k)
        assert abs(res[kind][3] - res[0][3]) <= 1e-12 * np.abs(x * res[0][0]).sum()
    rows = np.arange(0, m, 1009)
    ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], x[aj[ai[r]:ai[r + 1]]]) for r in rows])
    scale = np.abs(ref).max()
    assert np.abs(res[0][0][rows] - ref).max() <= 1e-12 * scale
    assert np.abs(res[0][1][rows] - (ref + y0[rows])).max() <= 1e-12 * (scale + 1.0)
    assert abs(res[0][3] - x @ res[0][0]) <= 1e-12 * np.abs(x * res[0][0]).sum()
    for p in (dx, dy, dy0):
        ctx.free(p)
    ctx.check(L.phip_mat_destroy(A))


def test_setup_threads_do_not_change_the_layout(ctx):
    """The host passes of phip_mat_create run on several threads from 2^20 stored entries on; tables numbered by first appearance
    (diagonals, pairs) must come out as on one thread: same counts, same products bit for bit -- for a stencil (pair dictionary, masks),
    a matrix with a value of its own per entry (streamed SELL values) and after phip_mat_update_values."""
    from petsc_amd import generators as G
    L = ctx.lib
    ai, aj, aa = G.stencil_rows_native(27, 40, 37, 41, 0, 40 * 37 * 41)
    m = len(ai) - 1
    assert len(aa) >= (1 << 20)
    rng = np.random.default_rng(12)
    varied = aa * rng.uniform(0.5, 1.5, len(aa))
    x = rng.uniform(-1.0, 1.0, m)
    dx, dy = ctx.to_device(x), ctx.to_device(np.zeros(m))
    out = {}
    for nt in (1, 5, 16):
        ctx.check(L.phip_set_setup_threads(nt))
        try:
            for name, vals in (("stencil", aa), ("varied", varied)):
                A = ctx.mat_create(m, m, ai, aj, vals)
                f, nd, pad = C.c_int(), C.c_int32(), C.c_int64()
                ctx.check(L.phip_mat_format_info(A, C.byref(f), C.byref(nd), C.byref(pad)))
                vb, vu, vn = C.c_int(), C.c_int(), C.c_int32()
                ctx.check(L.phip_mat_value_dict_info(A, C.byref(vb), C.byref(vu), C.byref(vn)))
                ctx.check(L.phip_mat_mult(A, dx, dy))
                y1 = ctx.to_host(dy, m)
                ctx.check(L.phip_mat_update_values(A, (varied if name == "stencil" else aa).ctypes.data_as(C.POINTER(C.c_double))))
                ctx.check(L.phip_mat_mult(A, dx, dy))
                out[nt, name] = ((f.value, nd.value, pad.value, vb.value, vn.value), y1, ctx.to_host(dy, m))
                ctx.check(L.phip_mat_destroy(A))
        finally:
            ctx.check(L.phip_set_setup_threads(0))
    assert out[1, "stencil"][0][3] == 1 and out[1, "varied"][0][3] == 0        # a pair dictionary / none
    for nt in (5, 16):
        for name in ("stencil", "varied"):
            assert out[nt, name][0] == out[1, name][0], (nt, name)
            assert np.array_equal(out[nt, name][1], out[1, name][1]) and np.array_equal(out[nt, name][2], out[1, name][2]), (nt, name)
    assert np.array_equal(out[1, "stencil"][2], out[1, "varied"][1])            # updated to the other's values: the other's product
    rows = np.arange(0, m, 211)
    ref = np.array([np.dot(aa[ai[r]:ai[r + 1]], x[aj[ai[r]:ai[r + 1]]]) for r in rows])
    assert np.abs(out[16, "stencil"][1][rows] - ref).max() <= 1e-12 * np.abs(ref).max()
    for p_ in (dx, dy):
        ctx.free(p_)


def test_default_nontemporal_elementwise(ctx):
    """vectors of 6.5 M entries (52 MB >= 48 MiB): the elementwise kernels, the batch kernel and the KSPCG batch read (and partly store)
    past the caches by default; the same calls with the feature off must give the same bits, and the closed forms on a sample."""
    from petsc_amd.capi import Ticket
    L = ctx.lib
    n = 6_500_001
    rng = np.random.default_rng(8)
    hx, hy, hz, hd = (rng.uniform(-1.0, 1.0, n) for _ in range(4))
    outs = {}
    for mode in (-1, 0):
        ctx.check(L.phip_set_ew_nontemporal(mode))
        x, y, z, d, w = (ctx.to_device(v) for v in (hx, hy, hz, hd, np.zeros(n)))
        ctx.check(L.phip_vec_axpy(ctx.h, n, y, 0.75, x)); ctx.sync()                     # single launches (ew_kernel)
        ctx.check(L.phip_vec_aypx(ctx.h, n, z, -0.5, x)); ctx.sync()
        ctx.check(L.phip_vec_waxpy(ctx.h, n, w, 1.25, x, y)); ctx.sync()
        r1 = [ctx.to_host(p, n) for p in (y, z, w)]
        tk = Ticket()                                                                    # the KSPCG batch (cg_update_kernel)
        ctx.check(L.phip_vec_axpy(ctx.h, n, x, 1e-3, z))
        ctx.check(L.phip_vec_axpy(ctx.h, n, y, -1e-3, w))
        ctx.check(L.phip_vec_pointwise_mult_reduce(ctx.h, n, w, d, y, 1, C.byref(tk)))
        sums = (C.c_double * 3)()
        ctx.check(L.phip_ticket_wait(ctx.h, C.byref(tk), sums))
        r2 = [ctx.to_host(p, n) for p in (x, y, w)]
        outs[mode] = (r1, r2, list(sums))
        for p in (x, y, z, d, w):
            ctx.free(p)
    ctx.check(L.phip_set_ew_nontemporal(-1))
    for a, b in zip(outs[-1][0] + outs[-1][1], outs[0][0] + outs[0][1]):
        assert np.array_equal(a, b)
    assert outs[-1][2] == outs[0][2]
    s = np.arange(0, n, 4099)
    y1 = hy[s] + 0.75 * hx[s]
    z1 = hx[s] + -0.5 * hz[s]
    w1 = y1 + 1.25 * hx[s]
    for got, want in zip(outs[-1][0], (y1, z1, w1)):
        assert np.abs(got[s] - want).max() <= 4e-16 * np.abs(want).max()
    x2 = hx[s] + 1e-3 * z1
    y2 = y1 + -1e-3 * w1
    w2 = hd[s] * y2
    for got, want in zip(outs[-1][1], (x2, y2, w2)):
        assert np.abs(got[s] - want).max() <= 1e-15 * np.abs(want).max()


def read_petscbin_mat(path):
    """the PetscBinary Mat format of MatView_SeqAIJ_Binary (aij.c:547-582), parsed with numpy alone"""
    raw = np.fromfile(path, dtype=">i4", count=4)
    assert raw[0] == 1211216
    M, N, nz = int(raw[1]), int(raw[2]), int(raw[3])
    lens = np.fromfile(path, dtype=">i4", count=M, offset=16).astype(np.int64)
    aj = np.fromfile(path, dtype=">i4", count=nz, offset=16 + 4 * M).astype(np.int32)
    aa = np.fromfile(path, dtype=">f8", count=nz, offset=16 + 4 * M + 4 * nz).astype(np.float64)
    ai = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    return M, N, ai, aj, aa


@pytest.mark.parametrize("size", [1, 8])
def test_rmat16_matload_mpiaij(orc, golden_dir, size):
    """BASELINE config 5 at the committed scale (2^16 rows, 955 094 nonzeros, written by the reference's PetscBinaryIO.py): MatLoad
    (mpiaij.c:2890-3089: row split m = M/size + ((M % size) > rank)) into MATMPIAIJHIP on 8 ranks, MatMult against the oracle's
    MatMult_MPIAIJ and against scipy's result stored with the fixture; garray and the scatter plan array-equal to their numpy
    derivation; per-rank nonzeros and the row-binning histogram of both blocks reported."""
    from petsc_amd import capi, petsc as P
    sys.path.insert(0, golden_dir)
    import plan_numpy
    src = os.path.join(golden_dir, "rmat16.petscbin")
    exp = np.load(os.path.join(golden_dir, "rmat16_expect.npz"), allow_pickle=False)
    M, N, ai, aj, aa = read_petscbin_mat(src)
    assert (M, N, len(aa)) == (int(exp["M"]), int(exp["N"]), int(exp["nz"]))
    xg = exp["x"]
    plan = plan_numpy.split_products(ai, aj, N, size) if size > 1 else None
    L = capi.load()

    def body(comm):
        A = P.Mat(comm)
        A.load(src)
        assert A.type() == ("mpiaijhip" if size > 1 else "seqaijhip") and A.sizes()[2:] == (M, N)
        rs, re = A.ownership_range()
        x = P.Vec.from_array(comm, xg[rs:re], N)
        y = x.duplicate()
        A.mult(x, y)
        out = {"y": y.array(), "range": (rs, re), "nnz": A.block(0)["nz"] + (A.block(1)["nz"] if size > 1 else 0), "hist": []}
        for which in range(2 if size > 1 else 1):
            h = (C.c_int64 * 8)()
            assert L.phip_mat_plan_lanes(A.device_mat(which), h) == 0
            out["hist"].append(list(h))
        if size > 1:
            gh, r = A.ghosts(), comm.rank
            np.testing.assert_array_equal(gh["garray"], plan["garray_%d" % r])
            for key in ("from_procs", "from_starts", "to_procs", "to_starts", "to_indices"):
                np.testing.assert_array_equal(gh[key], plan["%s_%d" % (key, r)], err_msg=key)
            blkB = A.block(1)
            np.testing.assert_array_equal(blkB["j"], plan["Bj_%d" % r])
            np.testing.assert_array_equal(blkB["i"], plan["Bi_%d" % r])
        for o in (x, y, A):
            o.destroy()
        return out

    res = run_ranks(size, body)
    y = np.concatenate([r["y"] for r in res])
    scale = np.abs(exp["y"]).max()
    assert np.abs(y - exp["y"]).max() <= 1e-12 * scale                                   # scipy csr_matvec, stored with the fixture
    yo = orc.operator(ai, aj, aa, size).mult(xg)                                          # MatMult_MPIAIJ restated: A x then B lvec added
    assert np.abs(y - yo).max() <= 1e-12 * scale
    rg = plan_numpy.ranges(M, size)
    assert [r["range"] for r in res] == [(int(rg[k]), int(rg[k + 1])) for k in range(size)]
    nnz = [r["nnz"] for r in res]
    assert sum(nnz) == len(aa) and all(sum(h) > 0 for r in res for h in r["hist"])       # every block has a non-empty binning histogram
    if size == 8:
        np.testing.assert_array_equal(nnz, exp["nnz_rank8"])
        print("rmat16 over 8 ranks: nnz per rank %s, imbalance %.2f; lanes-per-row histogram of rank 0's diagonal block %s"
              % (nnz, max(nnz) / (sum(nnz) / 8.0), res[0]["hist"][0]))


def test_full_size_config5(orc, tmp_path):
    """BASELINE config 5 at the bench leg's size (R-MAT 2^22 rows, 65 M nonzeros; 2^24 is the same code on 4 x the data and is
    measured by scripts/bench_configs.py): the matrix written in the format of MatView_SeqAIJ_Binary (aij.c:547-582), MatLoad on 8
    ranks (mpiaij.c:2890-3089) into MATMPIAIJHIP, MatMult_MPIAIJ against the oracle's restatement (A x, then B lvec added) and
    MatMult_SeqAIJ on one rank; the hot-column order of the row-block layout forced on and off gives the same bits."""
    from petsc_amd import capi, generators as G, petsc as P
    ai, aj, aa = G.rmat(22)
    M = len(ai) - 1
    assert (M, len(aj)) == (1 << 22, 65243634)
    path = str(tmp_path / "rmat22.petscbin")
    with open(path, "wb") as f:
        f.write(np.array([1211216, M, M, len(aj)], ">i4").tobytes())
        f.write(np.diff(ai).astype(">i4").tobytes()); f.write(aj.astype(">i4").tobytes()); f.write(aa.astype(">f8").tobytes())
    xg = G.random_vector(M)
    L = capi.load()

    def body(comm):
        A = P.Mat(comm)
        A.load(path)
        rs, re = A.ownership_range()
        x = P.Vec.from_array(comm, xg[rs:re], M)
        y = x.duplicate()
        A.mult(x, y)
        out = {"y": y.array(), "range": (rs, re), "nnz": A.block(0)["nz"] + (A.block(1)["nz"] if comm.size > 1 else 0)}
        for o in (x, y, A):
            o.destroy()
        return out

    for size in (8, 1):
        res = run_ranks(size, body)
        y = np.concatenate([r["y"] for r in res])
        yo = orc.operator(ai, aj, aa, size).mult(xg)
        assert np.abs(y - yo).max() <= 1e-12 * np.abs(yo).max(), size
        assert sum(r["nnz"] for r in res) == len(aj)
        assert [r["range"][0] for r in res] == [k * (M // size) for k in range(size)]
    os.remove(path)

    def hot(comm):
        outs = []
        for mode in (0, 2):
            assert L.phip_set_default_hot_columns(mode) == 0
            A = P.Mat.seqaij_with_arrays(comm, M, M, ai, aj, aa)
            x, y = A.create_vecs()
            x.set_array(xg)
            A.mult(x, y)
            outs.append(y.array())
            for o in (x, y, A):
                o.destroy()
        assert L.phip_set_default_hot_columns(1) == 0
        return outs

    y0, y2 = run_ranks(1, hot)[0]
    np.testing.assert_array_equal(y0, y2)
    assert np.abs(y0 - yo).max() <= 1e-12 * np.abs(yo).max()


"""ctypes front-end of the CPU oracle (oracle/petsc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the
product package petsc_amd never does.  Every function cites the reference through the C file.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_I = np.int32
_pI = C.POINTER(C.c_int32)
_pD = C.POINTER(C.c_double)

NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY, NORM_1_AND_2 = 0, 1, 2, 3, 4
KSP_NORM_PRECONDITIONED, KSP_NORM_UNPRECONDITIONED, KSP_NORM_NATURAL = 1, 2, 3
REFINE_NEVER, REFINE_IFNEEDED, REFINE_ALWAYS = 0, 1, 2


def build(fast=False, outdir=None):
    """Compile the oracle with gcc (recipe mirrors oracle/Makefile). Returns the .so path."""
    name = "liboracle_fast.so" if fast else "liboracle.so"
    outdir = outdir or _HERE
    out = os.path.join(outdir, name)
    src = os.path.join(_HERE, "petsc_oracle.c")
    if os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src) and not (fast and outdir == _HERE):
        return out
    # the pinned build has no OpenMP: its rank loops run one after the other.  The fast build (CPU baseline of bench.py) may spread
    # the emulated ranks over threads (OracleOperator(threads=...)).
    flags = ["-O3", "-march=native", "-fopenmp"] if fast else ["-O2", "-ffp-contract=off"]
    subprocess.check_call(["gcc", "-std=gnu99", *flags, "-fPIC", "-shared", "-fvisibility=hidden", "-o", out, src, "-lm"])
    return out


def _ip(a):
    return a.ctypes.data_as(_pI) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(_pD) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    def __init__(self, path=None, fast=False):
        path = path or build(fast=fast)
        L = self.lib = C.CDLL(path)
        L.orc_split_ownership.restype = C.c_int32
        L.orc_split_ownership.argtypes = [C.c_int32] * 3
        L.orc_vecdot.restype = C.c_double
        L.orc_seqaij_create.restype = C.c_void_p
        L.orc_operator_create.restype = C.c_void_p
        L.orc_ksp_create.restype = C.c_void_p
        L.orc_mpiaij_split.restype = C.c_int32

    # ---- partition -------------------------------------------------------------------------
    def split_ownership(self, N, size, rank):
        return int(self.lib.orc_split_ownership(N, size, rank))

    def ownership_ranges(self, N, size):
        r = np.zeros(size + 1, _I)
        self.lib.orc_ownership_ranges(C.c_int32(N), C.c_int32(size), _ip(r))
        return r

    # ---- assembly --------------------------------------------------------------------------
    def assemble(self, m, n, rows, cols, vals, add=True, nz_prealloc=5, one_by_one=True):
        """MatSetValues (one entry at a time, in the given order) + MatAssemblyEnd.
        Returns dict(i,j,a,diag,nonzerorowcnt,rmax,cprow_use,cprow_i,cprow_rindex,reallocs)."""
        L = self.lib
        A = C.c_void_p(L.orc_seqaij_create(C.c_int32(m), C.c_int32(n), C.c_int32(nz_prealloc), None))
        rows, cols, vals = _i32(rows), _i32(cols), _f64(vals)
        for k in range(len(rows)):
            rc = L.orc_seqaij_setvalues(A, C.c_int32(1), _ip(rows[k:k + 1]), C.c_int32(1), _ip(cols[k:k + 1]),
                                        _dp(vals[k:k + 1]), C.c_int(1 if add else 0))
            if rc:
                L.orc_seqaij_destroy(A)
                raise ValueError("orc_seqaij_setvalues error %d" % rc)
        L.orc_seqaij_assembly_end(A)
        out = self._seqaij_fetch(A)
        L.orc_seqaij_destroy(A)
        return out

    def assemble_blocks(self, m, n, blocks, add=True, nz_prealloc=5):
        """blocks: iterable of (rows[], cols[], dense row-major v) -> MatSetValues calls."""
        L = self.lib
        A = C.c_void_p(L.orc_seqaij_create(C.c_int32(m), C.c_int32(n), C.c_int32(nz_prealloc), None))
        for rws, cls, v in blocks:
            rws, cls, v = _i32(rws), _i32(cls), _f64(v)
            rc = L.orc_seqaij_setvalues(A, C.c_int32(len(rws)), _ip(rws), C.c_int32(len(cls)), _ip(cls), _dp(v), C.c_int(1 if add else 0))
            if rc:
                L.orc_seqaij_destroy(A)
                raise ValueError("orc_seqaij_setvalues error %d" % rc)
        L.orc_seqaij_assembly_end(A)
        out = self._seqaij_fetch(A)
        L.orc_seqaij_destroy(A)
        return out

    def _seqaij_fetch(self, A):
        L = self.lib
        info = np.zeros(8, _I)
        L.orc_seqaij_info(A, _ip(info))
        m, n, nz, nzr, rmax, cuse, cn, reallocs = [int(v) for v in info]
        ai = np.zeros(m + 1, _I); aj = np.zeros(nz, _I); aa = np.zeros(nz); diag = np.zeros(m, _I)
        cpi = np.zeros(cn + 1, _I); ridx = np.zeros(cn, _I)
        L.orc_seqaij_get(A, _ip(ai), _ip(aj), _dp(aa), _ip(diag), _ip(cpi), _ip(ridx))
        return dict(m=m, n=n, i=ai, j=aj, a=aa, diag=diag, nonzerorowcnt=nzr, rmax=rmax, cprow_use=bool(cuse),
                    cprow_i=cpi, cprow_rindex=ridx, reallocs=reallocs)

    def mark_diagonal(self, ai, aj):
        ai, aj = _i32(ai), _i32(aj)
        m = len(ai) - 1
        d = np.zeros(m, _I)
        self.lib.orc_mark_diagonal(C.c_int32(m), _ip(ai), _ip(aj), _ip(d))
        return d

    def check_compressed_row(self, ai, ratio=0.6):
        ai = _i32(ai)
        m = len(ai) - 1
        nzrows = int(np.count_nonzero(np.diff(ai) > 0))
        cn = C.c_int32(0)
        cpi = np.zeros(m + 1, _I); ridx = np.zeros(max(m, 1), _I)
        use = self.lib.orc_check_compressed_row(C.c_int32(m), _ip(ai), C.c_int32(nzrows), C.c_double(ratio), C.byref(cn), _ip(cpi), _ip(ridx))
        n = cn.value
        return bool(use), cpi[:n + 1].copy() if use else np.zeros(0, _I), ridx[:n].copy() if use else np.zeros(0, _I)

    # ---- SpMV ------------------------------------------------------------------------------
    def matmult(self, ai, aj, aa, x, cprow=None):
        ai, aj, aa, x = _i32(ai), _i32(aj), _f64(aa), _f64(x)
        m = len(ai) - 1
        y = np.empty(m)
        if cprow is not None:
            ci, cr = _i32(cprow[0]), _i32(cprow[1])
            self.lib.orc_matmult_seqaij(C.c_int32(m), _ip(ai), _ip(aj), _dp(aa), C.c_int(1), C.c_int32(len(cr)), _ip(ci), _ip(cr), _dp(x), _dp(y))
        else:
            self.lib.orc_matmult_seqaij(C.c_int32(m), _ip(ai), _ip(aj), _dp(aa), C.c_int(0), C.c_int32(0), None, None, _dp(x), _dp(y))
        return y

    def matmultadd(self, ai, aj, aa, x, y, cprow=None, inplace=False):
        ai, aj, aa, x, y = _i32(ai), _i32(aj), _f64(aa), _f64(x), _f64(y)
        m = len(ai) - 1
        z = y if inplace else np.empty(m)
        if cprow is not None:
            ci, cr = _i32(cprow[0]), _i32(cprow[1])
            self.lib.orc_matmultadd_seqaij(C.c_int32(m), _ip(ai), _ip(aj), _dp(aa), C.c_int(1), C.c_int32(len(cr)), _ip(ci), _ip(cr), _dp(x), _dp(y), _dp(z))
        else:
            self.lib.orc_matmultadd_seqaij(C.c_int32(m), _ip(ai), _ip(aj), _dp(aa), C.c_int(0), C.c_int32(0), None, None, _dp(x), _dp(y), _dp(z))
        return z

    def matmulttranspose(self, ai, aj, aa, n, x, y=None, cprow=None):
        """A^T x, or y + A^T x when y is given (n = number of columns)."""
        ai, aj, aa, x = _i32(ai), _i32(aj), _f64(aa), _f64(x)
        m = len(ai) - 1
        z = np.empty(n)
        cp = (C.c_int(0), C.c_int32(0), None, None)
        if cprow is not None:
            ci, cr = _i32(cprow[0]), _i32(cprow[1])
            cp = (C.c_int(1), C.c_int32(len(cr)), _ip(ci), _ip(cr))
        if y is None:
            self.lib.orc_matmulttranspose_seqaij(C.c_int32(m), C.c_int32(n), _ip(ai), _ip(aj), _dp(aa), *cp, _dp(x), _dp(z))
        else:
            y = _f64(y)
            self.lib.orc_matmulttransposeadd_seqaij(C.c_int32(m), C.c_int32(n), _ip(ai), _ip(aj), _dp(aa), *cp, _dp(x), _dp(y), _dp(z))
        return z

    def getdiagonal(self, ai, aj, aa):
        ai, aj, aa = _i32(ai), _i32(aj), _f64(aa)
        n = len(ai) - 1
        d = np.empty(n)
        self.lib.orc_getdiagonal_seqaij(C.c_int32(n), _ip(ai), _ip(aj), _dp(aa), _dp(d))
        return d

    def getrowmaxabs(self, ai, aa):
        ai, aa = _i32(ai), _f64(aa)
        m = len(ai) - 1
        d = np.empty(m)
        self.lib.orc_getrowmaxabs(C.c_int32(m), _ip(ai), _dp(aa), _dp(d))
        return d

    # ---- MPIAIJ split + scatter plan -------------------------------------------------------
    def mpiaij_split(self, gi, gj, ga, N, cstart, cend):
        """gi/gj/ga: this rank's rows (gi starts at 0), global sorted columns."""
        gi, gj, ga = _i32(gi), _i32(gj), _f64(ga)
        m = len(gi) - 1
        nzA, nzB = C.c_int32(0), C.c_int32(0)
        self.lib.orc_mpiaij_split_count(C.c_int32(m), _ip(gi), _ip(gj), C.c_int32(cstart), C.c_int32(cend), C.byref(nzA), C.byref(nzB))
        nzA, nzB = nzA.value, nzB.value
        Ai = np.zeros(m + 1, _I); Aj = np.zeros(nzA, _I); Aa = np.zeros(nzA)
        Bi = np.zeros(m + 1, _I); Bj = np.zeros(nzB, _I); Ba = np.zeros(nzB)
        garray = np.zeros(nzB, _I)
        ec = self.lib.orc_mpiaij_split(C.c_int32(m), C.c_int32(N), _ip(gi), _ip(gj), _dp(ga), C.c_int32(cstart), C.c_int32(cend),
                                       _ip(Ai), _ip(Aj), _dp(Aa), _ip(Bi), _ip(Bj), _dp(Ba), _ip(garray))
        return dict(Ai=Ai, Aj=Aj, Aa=Aa, Bi=Bi, Bj=Bj, Ba=Ba, garray=garray[:ec].copy())

    def scatter_plan(self, owners, garrays):
        """VecScatterCreate_PtoS plans for all ranks. Returns list of dicts."""
        owners = _i32(owners)
        size = len(owners) - 1
        garrays = [_i32(g) for g in garrays]
        ec_all = _i32([len(g) for g in garrays])
        ptrs = (_pI * size)(*[_ip(g) for g in garrays])
        tot = int(ec_all.sum())
        plans = []
        for r in range(size):
            ec = int(ec_all[r])
            fn, tn, nl, fc = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            fprocs = np.zeros(size, _I); fstarts = np.zeros(size + 1, _I); findices = np.zeros(max(ec, 1), _I)
            tprocs = np.zeros(size, _I); tstarts = np.zeros(size + 1, _I); tindices = np.zeros(max(tot, 1), _I)
            tloc = np.zeros(max(ec, 1), _I); floc = np.zeros(max(ec, 1), _I)
            self.lib.orc_scatter_plan_rank(C.c_int32(size), C.c_int32(r), _ip(owners), _ip(ec_all), ptrs,
                                           C.byref(fn), _ip(fprocs), _ip(fstarts), _ip(findices),
                                           C.byref(tn), _ip(tprocs), _ip(tstarts), _ip(tindices),
                                           C.byref(nl), _ip(tloc), _ip(floc), C.byref(fc))
            fn_, tn_, nl_ = fn.value, tn.value, nl.value
            plans.append(dict(from_n=fn_, from_procs=fprocs[:fn_].copy(), from_starts=fstarts[:fn_ + 1].copy(),
                              from_indices=findices[:int(fstarts[fn_])].copy(),
                              to_n=tn_, to_procs=tprocs[:tn_].copy(), to_starts=tstarts[:tn_ + 1].copy(),
                              to_indices=tindices[:int(tstarts[tn_])].copy(),
                              nlocal=nl_, to_local=tloc[:nl_].copy(), from_local=floc[:nl_].copy(), from_contiq=bool(fc.value)))
        return plans

    # ---- vectors ---------------------------------------------------------------------------
    def dot(self, x, y):
        x, y = _f64(x), _f64(y)
        return float(self.lib.orc_vecdot(C.c_int32(len(x)), _dp(x), _dp(y)))

    def norm(self, x, ntype=NORM_2):
        x = _f64(x)
        z = np.zeros(2)
        self.lib.orc_vecnorm(C.c_int32(len(x)), _dp(x), C.c_int(ntype), _dp(z))
        return (float(z[0]), float(z[1])) if ntype == NORM_1_AND_2 else float(z[0])

    def mdot(self, x, ys):
        x = _f64(x)
        ys = [_f64(y) for y in ys]
        nv = len(ys)
        ptrs = (_pD * max(nv, 1))(*[_dp(y) for y in ys])
        z = np.zeros(nv)
        self.lib.orc_vecmdot(C.c_int32(len(x)), _dp(x), C.c_int32(nv), ptrs, _dp(z))
        return z

    def maxpy(self, x, alpha, ys):
        x = _f64(x).copy()
        ys = [_f64(y) for y in ys]
        alpha = _f64(alpha)
        nv = len(ys)
        ptrs = (_pD * max(nv, 1))(*[_dp(y) for y in ys])
        self.lib.orc_vecmaxpy(C.c_int32(len(x)), _dp(x), C.c_int32(nv), _dp(alpha), ptrs)
        return x

    def axpy(self, y, alpha, x):
        y = _f64(y).copy(); x = _f64(x)
        self.lib.orc_vecaxpy(C.c_int32(len(y)), _dp(y), C.c_double(alpha), _dp(x))
        return y

    def aypx(self, y, alpha, x):
        y = _f64(y).copy(); x = _f64(x)
        self.lib.orc_vecaypx(C.c_int32(len(y)), _dp(y), C.c_double(alpha), _dp(x))
        return y

    def waxpy(self, alpha, x, y):
        x, y = _f64(x), _f64(y)
        w = np.empty_like(x)
        self.lib.orc_vecwaxpy(C.c_int32(len(x)), _dp(w), C.c_double(alpha), _dp(x), _dp(y))
        return w

    def axpby(self, y, a, b, x):
        y = _f64(y).copy(); x = _f64(x)
        self.lib.orc_vecaxpby(C.c_int32(len(y)), _dp(y), C.c_double(a), C.c_double(b), _dp(x))
        return y

    def axpbypcz(self, z, a, b, g, x, y):
        z = _f64(z).copy(); x = _f64(x); y = _f64(y)
        self.lib.orc_vecaxpbypcz(C.c_int32(len(z)), _dp(z), C.c_double(a), C.c_double(b), C.c_double(g), _dp(x), _dp(y))
        return z

    def scale(self, x, alpha):
        x = _f64(x).copy()
        self.lib.orc_vecscale(C.c_int32(len(x)), _dp(x), C.c_double(alpha))
        return x

    def pointwisemult(self, x, y):
        x, y = _f64(x), _f64(y)
        w = np.empty_like(x)
        self.lib.orc_vecpointwisemult(C.c_int32(len(x)), _dp(w), _dp(x), _dp(y))
        return w

    def pointwisedivide(self, x, y):
        x, y = _f64(x), _f64(y)
        w = np.empty_like(x)
        self.lib.orc_vecpointwisedivide(C.c_int32(len(x)), _dp(w), _dp(x), _dp(y))
        return w

    def reciprocal(self, x):
        x = _f64(x).copy()
        self.lib.orc_vecreciprocal(C.c_int32(len(x)), _dp(x))
        return x

    def jacobi_setup(self, diag, useabs=False):
        d = _f64(diag).copy()
        self.lib.orc_jacobi_setup(C.c_int32(len(d)), _dp(d), C.c_int(1 if useabs else 0))
        return d

    # ---- operator + KSP --------------------------------------------------------------------
    def operator(self, gi, gj, ga, size=1, threads=1):
        return OracleOperator(self, gi, gj, ga, size, threads)


class OracleOperator:
    """Row-block distributed (MPIAIJ-shaped) operator emulated in one process."""

    def __init__(self, orc, gi, gj, ga, size=1, threads=1):
        """threads > 1 (fast build only): the emulated ranks run on that many OpenMP threads -- same results, the CPU baseline of
        'size ranks of MPIAIJ on size cores'."""
        self.orc = orc
        self.gi, self.gj, self.ga = _i32(gi), _i32(gj), _f64(ga)
        self.N = len(self.gi) - 1
        self.size = size
        orc.lib.orc_operator_create_threaded.restype = C.c_void_p
        self.h = C.c_void_p(orc.lib.orc_operator_create_threaded(C.c_int32(self.N), C.c_int32(size), _ip(self.gi), _ip(self.gj), _dp(self.ga), C.c_int(threads)))

    def __del__(self):
        try:
            self.orc.lib.orc_operator_destroy(self.h)
        except Exception:
            pass

    def mult(self, x):
        x = _f64(x)
        y = np.empty(self.N)
        self.orc.lib.orc_operator_mult(self.h, _dp(x), _dp(y))
        return y

    def mult_transpose(self, x, y=None):
        x = _f64(x)
        z = np.empty(self.N)
        yy = _f64(y) if y is not None else z
        self.orc.lib.orc_operator_mult_transpose(self.h, _dp(x), _dp(yy), _dp(z), C.c_int(0 if y is None else 1))
        return z

    def diagonal(self):
        d = np.empty(self.N)
        self.orc.lib.orc_operator_getdiagonal(self.h, _dp(d))
        return d

    def solve(self, b, ksp_type="cg", pc="jacobi", rtol=1e-5, abstol=1e-50, divtol=1e4, max_it=10000, x0=None,
              normtype=KSP_NORM_PRECONDITIONED, restart=30, cgs_refine=REFINE_NEVER, useabs=False):
        L = self.orc.lib
        b = _f64(b)
        x = np.zeros(self.N) if x0 is None else _f64(x0).copy()
        d = self.orc.jacobi_setup(self.diagonal(), useabs) if pc == "jacobi" else None
        k = C.c_void_p(L.orc_ksp_create())
        L.orc_ksp_set(k, C.c_double(rtol), C.c_double(abstol), C.c_double(divtol), C.c_int32(max_it), C.c_int(0 if x0 is None else 1),
                      C.c_int(normtype), C.c_int32(restart), C.c_int(cgs_refine))
        hist = np.zeros(max_it + 2)
        L.orc_ksp_set_history(k, _dp(hist), C.c_int32(len(hist)))
        fn = {"cg": L.orc_ksp_cg, "pipecg": L.orc_ksp_pipecg, "gmres": L.orc_ksp_gmres}[ksp_type]
        fn(self.h, k, _dp(d) if d is not None else None, _dp(b), _dp(x))
        its, reason, rnorm, nh = C.c_int32(0), C.c_int(0), C.c_double(0), C.c_int32(0)
        L.orc_ksp_get(k, C.byref(its), C.byref(reason), C.byref(rnorm), C.byref(nh))
        L.orc_ksp_destroy(k)
        return dict(x=x, its=its.value, reason=reason.value, rnorm=rnorm.value, history=hist[:min(nh.value, len(hist))].copy())

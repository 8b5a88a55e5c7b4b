"""ctypes binding of libpetschip_host.so (include/petschip_host.h): the stand-alone mirror of the PETSc API slice.

Argument types are derived from the header itself, so the binding cannot drift from the C ABI.  Thin object wrappers
(Comm, Vec, Mat, KSP) keep the PETSc call names; every call raises PetscError on a non-zero PetscErrorCode.
There is no CPU fallback: a missing library raises ImportError.
"""
import ctypes as C
import os
import re
import sys

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "lib", "libpetschip_host.so")
HEADER = os.path.join(os.path.dirname(_HERE), "include", "petschip_host.h")
# the restated KSP drivers (test and benchmark harness, not product): loaded only when a KSP* function is first asked for
HARNESS_LIB_PATH = os.path.join(_HERE, "lib", "libpetschip_harness.so")
HARNESS_HEADER = os.path.join(os.path.dirname(_HERE), "harness", "ksp_restated.h")

INSERT_VALUES, ADD_VALUES = 1, 2
MAT_FINAL_ASSEMBLY, MAT_FLUSH_ASSEMBLY = 0, 1
NORM_1, NORM_2, NORM_FROBENIUS, NORM_INFINITY, NORM_1_AND_2 = 0, 1, 2, 3, 4
KSP_NORM_NONE, KSP_NORM_PRECONDITIONED, KSP_NORM_UNPRECONDITIONED, KSP_NORM_NATURAL = 0, 1, 2, 3
REFINE_NEVER, REFINE_IFNEEDED, REFINE_ALWAYS = 0, 1, 2
PC_JACOBI_DIAGONAL, PC_JACOBI_ROWMAX, PC_JACOBI_ROWSUM = 0, 1, 2
PETSC_DECIDE = -1
PETSC_DEFAULT = -2

_SCALAR = {"PetscInt": C.c_int32, "int32_t": C.c_int32, "int": C.c_int, "PetscBool": C.c_int, "PetscScalar": C.c_double,
           "PetscReal": C.c_double, "double": C.c_double, "int64_t": C.c_int64, "PetscErrorCode": C.c_int,
           "InsertMode": C.c_int, "MatAssemblyType": C.c_int, "NormType": C.c_int, "KSPNormType": C.c_int,
           "KSPGMRESCGSRefinementType": C.c_int, "PCJacobiType": C.c_int, "KSPConvergedReason": C.c_int,
           "unsigned": C.c_ulong}  # the only unsigned parameter is PetscRandomSetSeed's `unsigned long`
_HANDLES = {"PetscComm", "Vec", "Mat", "PC", "KSP", "phip_ctx", "phip_mat", "phip_comm", "PetscRandom"}


def _ctype(decl):
    decl = decl.strip()
    if "(*" in decl:
        return C.c_void_p
    if decl.endswith("]"):
        decl = decl[:decl.index("[")].rstrip()
        ptr = 1
    else:
        ptr = 0
    ptr += decl.count("*")
    words = [w for w in re.sub(r"[*]", " ", decl).split() if w != "const"]
    base = words[0]
    if base == "char" and ptr == 1:
        return C.c_char_p
    if ptr:
        return C.c_void_p
    if base in _HANDLES or base.endswith("_fn"):
        return C.c_void_p
    return _SCALAR[base]


def parse_header(path=None):
    path = path or HEADER
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    sigs = {}
    for m in re.finditer(r"\b(PetscErrorCode|const char\s*\*)\s*([A-Za-z0-9_]+)\s*\(([^;{]*)\)\s*;", txt):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        depth, cur, parts = 0, "", []
        for ch in args:
            if ch == "(":
                depth += 1
            if ch == ")":
                depth -= 1
            if ch == "," and depth == 0:
                parts.append(cur)
                cur = ""
            else:
                cur += ch
        parts.append(cur)
        parts = [p for p in parts if p.strip() and p.strip() != "void"]
        sigs[name] = (C.c_int if ret == "PetscErrorCode" else C.c_char_p, [_ctype(p) for p in parts])
    return sigs


class PetscError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("PetscErrorCode %d: %s" % (code, msg))
        self.code = code


_lib = None


def _typed(L, header):
    for name, (res, args) in parse_header(header).items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


class _Libs:
    """libpetschip_host.so (the product's PETSc-API mirror) and, behind it, libpetschip_harness.so: an entry point the first does not
    export (the KSP* drivers) is looked up in the second, which is loaded at that moment -- product-only users never load the harness."""

    def __init__(self, host):
        self._host, self._harness = host, None

    def harness(self):
        if self._harness is None:
            if not os.path.exists(HARNESS_LIB_PATH):
                raise ImportError("libpetschip_harness.so not found at %s -- build it with `python -m petsc_amd.build`" % HARNESS_LIB_PATH)
            self._harness = _typed(C.CDLL(HARNESS_LIB_PATH, mode=C.RTLD_GLOBAL), HARNESS_HEADER)
        return self._harness

    def __getattr__(self, name):
        try:
            fn = getattr(self._host, name)
        except AttributeError:
            fn = getattr(self.harness(), name)
        setattr(self, name, fn)
        return fn


def lib():
    global _lib
    if _lib is None:
        capi.load()  # libpetschip.so first (RTLD_GLOBAL), fails loudly if missing
        if not os.path.exists(HOST_LIB_PATH):
            raise ImportError("libpetschip_host.so not found at %s -- build it with `python -m petsc_amd.build`" % HOST_LIB_PATH)
        _lib = _Libs(_typed(C.CDLL(HOST_LIB_PATH, mode=C.RTLD_GLOBAL), HEADER))
    return _lib


def chk(rc):
    if rc:
        raise PetscError(rc, lib().PetscHipLastError().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Comm:
    """PetscHipInitialize: one rank per GPU.  backend 'self' | 'rccl' | 'threads'."""

    def __init__(self, device=0, backend="self", id128=None, nranks=1, rank=0):
        self.h = C.c_void_p()
        buf = C.create_string_buffer(bytes(id128), 128) if id128 is not None else None
        chk(lib().PetscHipInitialize(device, backend.encode(), buf, nranks, rank, C.byref(self.h)))
        self.rank, self.size = rank, nranks
        s = C.c_void_p()
        chk(lib().PetscCommGetSelf(self.h, C.byref(s)))
        self.self_h = s
        ctx = C.c_void_p()
        chk(lib().PetscCommGetContext(self.h, C.byref(ctx)))
        self.ctx = ctx

    @classmethod
    def from_callbacks(cls, device, nranks, rank, allgather, alltoallv):
        """PetscHipInitializeWithCallbacks: set-up collectives supplied by the application (see petsc_amd/torchcomm.py).
        allgather(send: bytes) -> bytes of all ranks in rank order; alltoallv(send: int32 array, send_counts) -> (recv, ...)."""
        self = cls.__new__(cls)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
        AV = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32))

        def _ag(user, send, recv, nbytes):
            try:
                data = C.string_at(send, nbytes) if nbytes else b""
                out = allgather(data)
                assert len(out) == nbytes * nranks
                C.memmove(recv, out, len(out))
                return 0
            except Exception as e:  # the C side turns this into PHIP_ERR_LIB
                sys.stderr.write("allgather callback failed: %r\n" % (e,))
                return 1

        def _av(user, send, scounts, recv, rcounts):
            try:
                sc = np.array([scounts[i] for i in range(nranks)], np.int32)
                rc = np.array([rcounts[i] for i in range(nranks)], np.int32)
                tot = int(sc.sum())
                sbuf = np.ctypeslib.as_array(send, shape=(max(tot, 1),))[:tot].copy() if tot else np.zeros(0, np.int32)
                out = np.ascontiguousarray(alltoallv(sbuf, sc, rc), np.int32)
                assert len(out) == int(rc.sum())
                if len(out):
                    C.memmove(recv, out.ctypes.data, 4 * len(out))
                return 0
            except Exception as e:
                sys.stderr.write("alltoallv callback failed: %r\n" % (e,))
                return 1

        self._cb = (AG(_ag), AV(_av))  # keep the thunks alive as long as the communicator
        self.h = C.c_void_p()
        chk(lib().PetscHipInitializeWithCallbacks(device, nranks, rank, C.cast(self._cb[0], C.c_void_p), C.cast(self._cb[1], C.c_void_p), None, C.byref(self.h)))
        self.rank, self.size = rank, nranks
        s = C.c_void_p()
        chk(lib().PetscCommGetSelf(self.h, C.byref(s)))
        self.self_h = s
        ctx = C.c_void_p()
        chk(lib().PetscCommGetContext(self.h, C.byref(ctx)))
        self.ctx = ctx
        return self

    @property
    def SELF(self):
        c = Comm.__new__(Comm)
        c.h, c.rank, c.size, c.self_h, c.ctx = self.self_h, 0, 1, self.self_h, self.ctx
        return c

    def hip_comm(self):
        """the phip_comm handle (include/petschip.h) behind this communicator"""
        c = C.c_void_p()
        chk(lib().PetscCommGetHipComm(self.h, C.byref(c)))
        return c

    def set_byproducts(self, flg):
        chk(lib().PetscHipSetByProducts(self.h, 1 if flg else 0))

    def set_host_reductions(self, flg):
        chk(lib().PetscHipSetHostReductions(self.h, 1 if flg else 0))

    def finalize(self):
        if self.h:
            chk(lib().PetscHipFinalize(C.byref(self.h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        rc = capi.load().phip_comm_unique_id(buf)
        if rc:
            raise capi.PhipError(rc, capi.load().phip_last_error().decode())
        return buf.raw


class Vec:
    def __init__(self, comm=None, n=PETSC_DECIDE, N=PETSC_DECIDE, vtype="hip", handle=None):
        self.comm = comm
        if handle is not None:
            self.h = handle
            return
        self.h = C.c_void_p()
        chk(lib().VecCreate(comm.h, C.byref(self.h)))
        chk(lib().VecSetSizes(self.h, n, N))
        chk(lib().VecSetType(self.h, vtype.encode()))

    @classmethod
    def from_array(cls, comm, arr, N=PETSC_DECIDE, vtype="hip"):
        arr = _f64(arr)
        v = cls(comm, len(arr), N, vtype)
        v.set_array(arr)
        return v

    def duplicate(self):
        w = C.c_void_p()
        chk(lib().VecDuplicate(self.h, C.byref(w)))
        return Vec(self.comm, handle=w)

    def destroy(self):
        if self.h:
            chk(lib().VecDestroy(C.byref(self.h)))

    @property
    def local_size(self):
        n = C.c_int32()
        chk(lib().VecGetLocalSize(self.h, C.byref(n)))
        return n.value

    @property
    def size(self):
        n = C.c_int32()
        chk(lib().VecGetSize(self.h, C.byref(n)))
        return n.value

    def ownership_range(self):
        lo, hi = C.c_int32(), C.c_int32()
        chk(lib().VecGetOwnershipRange(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def type(self):
        t = C.c_char_p()
        chk(lib().VecGetType(self.h, C.byref(t)))
        return t.value.decode()

    def array(self):
        p = C.c_void_p()
        n = self.local_size
        chk(lib().VecGetArrayRead(self.h, C.byref(p)))
        out = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(max(n, 1),))[:n].copy() if n else np.zeros(0)
        chk(lib().VecRestoreArrayRead(self.h, C.byref(p)))
        return out

    def set_array(self, arr):
        arr = _f64(arr)
        p = C.c_void_p()
        n = self.local_size
        assert len(arr) == n
        chk(lib().VecGetArray(self.h, C.byref(p)))
        if n:
            C.memmove(p, arr.ctypes.data, 8 * n)
        chk(lib().VecRestoreArray(self.h, C.byref(p)))

    def device_ptr(self):
        p = C.c_void_p()
        chk(lib().VecHipGetArrayRead(self.h, C.byref(p)))
        return p

    def set_values(self, ix, y, mode=INSERT_VALUES):
        ix, y = _i32(ix), _f64(y)
        chk(lib().VecSetValues(self.h, len(ix), _p(ix), _p(y), mode))

    def assemble(self):
        chk(lib().VecAssemblyBegin(self.h))
        chk(lib().VecAssemblyEnd(self.h))

    def set(self, a): chk(lib().VecSet(self.h, a))
    def copy_to(self, y): chk(lib().VecCopy(self.h, y.h))
    def swap(self, y): chk(lib().VecSwap(self.h, y.h))
    def scale(self, a): chk(lib().VecScale(self.h, a))
    def axpy(self, a, x): chk(lib().VecAXPY(self.h, a, x.h))
    def aypx(self, a, x): chk(lib().VecAYPX(self.h, a, x.h))
    def axpby(self, a, b, x): chk(lib().VecAXPBY(self.h, a, b, x.h))
    def waxpy(self, a, x, y): chk(lib().VecWAXPY(self.h, a, x.h, y.h))
    def axpbypcz(self, a, b, g, x, y): chk(lib().VecAXPBYPCZ(self.h, a, b, g, x.h, y.h))
    def pointwise_mult(self, x, y): chk(lib().VecPointwiseMult(self.h, x.h, y.h))
    def pointwise_divide(self, x, y): chk(lib().VecPointwiseDivide(self.h, x.h, y.h))
    def reciprocal(self): chk(lib().VecReciprocal(self.h))

    def maxpy(self, alpha, xs):
        al = _f64(alpha)
        arr = (C.c_void_p * max(len(xs), 1))(*[x.h for x in xs])
        chk(lib().VecMAXPY(self.h, len(xs), _p(al), arr))

    def dot(self, y):
        z = C.c_double()
        chk(lib().VecDot(self.h, y.h, C.byref(z)))
        return z.value

    def tdot(self, y):
        z = C.c_double()
        chk(lib().VecTDot(self.h, y.h, C.byref(z)))
        return z.value

    def mdot(self, ys):
        arr = (C.c_void_p * max(len(ys), 1))(*[y.h for y in ys])
        z = np.zeros(len(ys))
        chk(lib().VecMDot(self.h, len(ys), arr, _p(z)))
        return z

    def set_random(self, seed=None, interval=None):
        """VecSetRandom with a rand48 PetscRandom (default seed 0x12345678 + 76543*rank, randomc.c:347)."""
        r = C.c_void_p()
        chk(lib().PetscRandomCreate(self.comm.h, C.byref(r)))
        if seed is not None:
            chk(lib().PetscRandomSetSeed(r, seed))
            chk(lib().PetscRandomSeed(r))
        if interval is not None:
            chk(lib().PetscRandomSetInterval(r, float(interval[0]), float(interval[1])))
        chk(lib().VecSetRandom(self.h, r))
        chk(lib().PetscRandomDestroy(C.byref(r)))

    def place_array(self, a):
        """VecPlaceArray: `a` (float64, contiguous, local length) must stay alive until reset_array()."""
        assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"] and len(a) == self.local_size
        self._placed = a
        chk(lib().VecPlaceArray(self.h, a.ctypes.data_as(C.c_void_p)))

    def reset_array(self):
        chk(lib().VecResetArray(self.h))
        self._placed = None

    # split reductions (comb.c): begin ... other work ... end, in the same order
    def dot_begin(self, y): chk(lib().VecDotBegin(self.h, y.h, None))

    def dot_end(self, y):
        z = C.c_double()
        chk(lib().VecDotEnd(self.h, y.h, C.byref(z)))
        return z.value

    def norm_begin(self, ntype=NORM_2): chk(lib().VecNormBegin(self.h, ntype, None))

    def norm_end(self, ntype=NORM_2):
        z = (C.c_double * 2)()
        chk(lib().VecNormEnd(self.h, ntype, z))
        return (z[0], z[1]) if ntype == NORM_1_AND_2 else z[0]

    def mtdot(self, ys):
        arr = (C.c_void_p * max(len(ys), 1))(*[y.h for y in ys])
        z = np.zeros(len(ys))
        chk(lib().VecMTDot(self.h, len(ys), arr, _p(z)))
        return z

    def dot_norm2(self, t):
        """VecDotNorm2(self, t): (self . t, t . t)."""
        dp, nm = C.c_double(), C.c_double()
        chk(lib().VecDotNorm2(self.h, t.h, C.byref(dp), C.byref(nm)))
        return dp.value, nm.value

    def norm(self, ntype=NORM_2):
        z = (C.c_double * 2)()
        chk(lib().VecNorm(self.h, ntype, z))
        return (z[0], z[1]) if ntype == NORM_1_AND_2 else z[0]

    def normalize(self):
        z = C.c_double()
        chk(lib().VecNormalize(self.h, C.byref(z)))
        return z.value

    def load(self, filename, offset=0):
        off = C.c_int64(offset)
        chk(lib().VecLoad(self.h, filename.encode(), C.byref(off)))
        return off.value

    def view(self, filename, append=False):
        chk(lib().VecView(self.h, filename.encode(), 1 if append else 0))


class Mat:
    def __init__(self, comm=None, m=PETSC_DECIDE, n=PETSC_DECIDE, M=PETSC_DECIDE, N=PETSC_DECIDE, mtype="aijhip", handle=None, set_type=True):
        self.comm = comm
        if handle is not None:
            self.h = handle
            return
        self.h = C.c_void_p()
        chk(lib().MatCreate(comm.h, C.byref(self.h)))
        if not (m == n == M == N == PETSC_DECIDE):
            chk(lib().MatSetSizes(self.h, m, n, M, N))
            if set_type:
                chk(lib().MatSetType(self.h, mtype.encode()))

    @classmethod
    def seqaij_with_arrays(cls, comm, m, n, ai, aj, aa):
        ai, aj, aa = _i32(ai), _i32(aj), _f64(aa)
        h = C.c_void_p()
        chk(lib().MatCreateSeqAIJWithArrays(comm.h, m, n, _p(ai), _p(aj), _p(aa), C.byref(h)))
        return cls(comm, handle=h)

    @classmethod
    def mpiaij_with_arrays(cls, comm, m, n, M, N, ai, aj, aa):
        ai, aj, aa = _i32(ai), _i32(aj), _f64(aa)
        h = C.c_void_p()
        chk(lib().MatCreateMPIAIJWithArrays(comm.h, m, n, M, N, _p(ai), _p(aj), _p(aa), C.byref(h)))
        return cls(comm, handle=h)

    @classmethod
    def from_global_csr(cls, comm, ai, aj, aa, N=None):
        """Rows of a global CSR that MPIAIJ's ownership rule gives this rank (Seq when comm.size == 1)."""
        ai, aj, aa = _i32(ai), _i32(aj), _f64(aa)
        M = len(ai) - 1
        N = N or M
        if comm.size == 1:
            return cls.seqaij_with_arrays(comm, M, N, ai, aj, aa)
        m = M // comm.size + (1 if (M % comm.size) > comm.rank else 0)
        rs = sum(M // comm.size + (1 if (M % comm.size) > r else 0) for r in range(comm.rank))
        li = (ai[rs:rs + m + 1] - ai[rs]).astype(np.int32)
        return cls.mpiaij_with_arrays(comm, m, PETSC_DECIDE, M, N, li, aj[ai[rs]:ai[rs + m]], aa[ai[rs]:ai[rs + m]])

    def destroy(self):
        if self.h:
            chk(lib().MatDestroy(C.byref(self.h)))

    def type(self):
        t = C.c_char_p()
        chk(lib().MatGetType(self.h, C.byref(t)))
        return t.value.decode()

    def set_up(self): chk(lib().MatSetUp(self.h))

    def seqaij_set_preallocation(self, nz, nnz=None):
        nnz = _i32(nnz) if nnz is not None else None
        chk(lib().MatSeqAIJSetPreallocation(self.h, nz, _p(nnz)))

    def mpiaij_set_preallocation(self, d_nz, d_nnz=None, o_nz=0, o_nnz=None):
        d_nnz = _i32(d_nnz) if d_nnz is not None else None
        o_nnz = _i32(o_nnz) if o_nnz is not None else None
        chk(lib().MatMPIAIJSetPreallocation(self.h, d_nz, _p(d_nnz), o_nz, _p(o_nnz)))

    def set_values(self, rows, cols, v, mode=INSERT_VALUES):
        rows, cols, v = _i32(rows), _i32(cols), _f64(v)
        chk(lib().MatSetValues(self.h, len(rows), _p(rows), len(cols), _p(cols), _p(v), mode))

    def assemble(self, atype=MAT_FINAL_ASSEMBLY):
        chk(lib().MatAssemblyBegin(self.h, atype))
        chk(lib().MatAssemblyEnd(self.h, atype))

    def zero_entries(self): chk(lib().MatZeroEntries(self.h))

    def set_global_values(self, ai, aj, aa):
        """INSERT this rank's rows of a global CSR (MatSetValues row by row, global columns) and assemble: with an unchanged nonzero
        pattern only the values travel to the device (MatAssemblyEnd, aij.c:968-1024 keeps nonzerostate)."""
        ai, aj, aa = _i32(ai), _i32(aj), _f64(aa)
        lo, hi = self.ownership_range()
        for r in range(lo, hi):
            s, e = int(ai[r]), int(ai[r + 1])
            if e > s:
                self.set_values([r], aj[s:e], aa[s:e])
        self.assemble()

    def zero_rows(self, rows, diag=0.0):
        rows = _i32(rows)
        chk(lib().MatZeroRows(self.h, len(rows), _p(rows), diag, None, None))

    def sizes(self):
        m, n, M, N = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        chk(lib().MatGetLocalSize(self.h, C.byref(m), C.byref(n)))
        chk(lib().MatGetSize(self.h, C.byref(M), C.byref(N)))
        return m.value, n.value, M.value, N.value

    def ownership_range(self):
        lo, hi = C.c_int32(), C.c_int32()
        chk(lib().MatGetOwnershipRange(self.h, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def create_vecs(self):
        r, l = C.c_void_p(), C.c_void_p()
        chk(lib().MatCreateVecs(self.h, C.byref(r), C.byref(l)))
        return Vec(self.comm, handle=r), Vec(self.comm, handle=l)

    def mult(self, x, y): chk(lib().MatMult(self.h, x.h, y.h))
    def mult_add(self, x, y, z): chk(lib().MatMultAdd(self.h, x.h, y.h, z.h))
    def mult_transpose(self, x, y): chk(lib().MatMultTranspose(self.h, x.h, y.h))
    def mult_transpose_add(self, x, y, z): chk(lib().MatMultTransposeAdd(self.h, x.h, y.h, z.h))
    def get_diagonal(self, v): chk(lib().MatGetDiagonal(self.h, v.h))
    def scale(self, a): chk(lib().MatScale(self.h, float(a)))
    def diagonal_scale(self, l=None, r=None): chk(lib().MatDiagonalScale(self.h, l.h if l is not None else None, r.h if r is not None else None))
    def shift(self, a): chk(lib().MatShift(self.h, float(a)))
    def get_row_max_abs(self, v): chk(lib().MatGetRowMaxAbs(self.h, v.h, None))
    def get_row_sum(self, v): chk(lib().MatGetRowSum(self.h, v.h))

    def block(self, which=0):
        """Assembly products of the Seq matrix / diagonal block (0) or off-diagonal block (1)."""
        m, n, nz, nzr, rmax, cu, cn = [C.c_int32() for _ in range(7)]
        chk(lib().MatHipGetBlockInfo(self.h, which, C.byref(m), C.byref(n), C.byref(nz), C.byref(nzr), C.byref(rmax), C.byref(cu), C.byref(cn)))
        ai = np.zeros(m.value + 1, np.int32); aj = np.zeros(nz.value, np.int32); aa = np.zeros(nz.value)
        diag = np.zeros(m.value, np.int32); cpi = np.zeros(cn.value + 1, np.int32); cri = np.zeros(cn.value, np.int32)
        chk(lib().MatHipGetBlockArrays(self.h, which, _p(ai), _p(aj), _p(aa), _p(diag), _p(cpi), _p(cri)))
        return dict(m=m.value, n=n.value, nz=nz.value, nonzerorowcnt=nzr.value, rmax=rmax.value, cprow_use=bool(cu.value),
                    i=ai, j=aj, a=aa, diag=diag, cprow_i=cpi if cu.value else np.zeros(0, np.int32), cprow_rindex=cri)

    def device_mat(self, which=0):
        """phip_mat handle of the Seq matrix / diagonal block (0) or the off-diagonal block (1)."""
        d = C.c_void_p()
        chk(lib().MatHipGetDeviceMat(self.h, which, C.byref(d)))
        return d

    def ghosts(self):
        ec, tn, tl, fn = [C.c_int32() for _ in range(4)]
        chk(lib().MatHipGetGhostInfo(self.h, C.byref(ec), C.byref(tn), C.byref(tl), C.byref(fn)))
        g = np.zeros(ec.value, np.int32); tp = np.zeros(tn.value, np.int32); ts = np.zeros(tn.value + 1, np.int32)
        ti = np.zeros(tl.value, np.int32); fp = np.zeros(fn.value, np.int32); fs = np.zeros(fn.value + 1, np.int32)
        chk(lib().MatHipGetGhostArrays(self.h, _p(g), _p(tp), _p(ts), _p(ti), _p(fp), _p(fs)))
        return dict(garray=g, to_procs=tp, to_starts=ts, to_indices=ti, from_procs=fp, from_starts=fs)

    def load(self, filename, offset=0):
        off = C.c_int64(offset)
        chk(lib().MatLoad(self.h, filename.encode(), C.byref(off)))
        return off.value

    def view(self, filename, append=False):
        chk(lib().MatView(self.h, filename.encode(), 1 if append else 0))


class KSP:
    def __init__(self, comm, ksp_type="cg", pc_type="jacobi"):
        self.comm = comm
        self.h = C.c_void_p()
        chk(lib().KSPCreate(comm.h, C.byref(self.h)))
        chk(lib().KSPSetType(self.h, ksp_type.encode()))
        self.pc = C.c_void_p()
        chk(lib().KSPGetPC(self.h, C.byref(self.pc)))
        chk(lib().PCSetType(self.pc, pc_type.encode()))
        self._hist = None

    def set_operators(self, A, P=None): chk(lib().KSPSetOperators(self.h, A.h, (P or A).h))
    def set_tolerances(self, rtol=PETSC_DEFAULT, abstol=PETSC_DEFAULT, dtol=PETSC_DEFAULT, max_it=PETSC_DEFAULT): chk(lib().KSPSetTolerances(self.h, rtol, abstol, dtol, max_it))
    def set_initial_guess_nonzero(self, flg=True): chk(lib().KSPSetInitialGuessNonzero(self.h, 1 if flg else 0))
    def set_norm_type(self, t): chk(lib().KSPSetNormType(self.h, t))
    def gmres_set_restart(self, r): chk(lib().KSPGMRESSetRestart(self.h, r))
    def gmres_set_cgs_refinement(self, t): chk(lib().KSPGMRESSetCGSRefinementType(self.h, t))
    def cg_set_single_reduction(self, flg=True): chk(lib().KSPCGSetSingleReduction(self.h, 1 if flg else 0))
    def pc_jacobi_set_type(self, t): chk(lib().PCJacobiSetType(self.pc, t))
    def pc_jacobi_set_use_abs(self, flg=True): chk(lib().PCJacobiSetUseAbs(self.pc, 1 if flg else 0))

    def set_residual_history(self, n=20000):
        self._hist = np.zeros(n)
        chk(lib().KSPSetResidualHistory(self.h, _p(self._hist), n, 1))

    def solve(self, b, x):
        chk(lib().KSPSolve(self.h, b.h, x.h))
        its, reason, rn = C.c_int32(), C.c_int(), C.c_double()
        chk(lib().KSPGetIterationNumber(self.h, C.byref(its)))
        chk(lib().KSPGetConvergedReason(self.h, C.byref(reason)))
        chk(lib().KSPGetResidualNorm(self.h, C.byref(rn)))
        out = dict(its=its.value, reason=reason.value, rnorm=rn.value)
        if self._hist is not None:
            p, na = C.c_void_p(), C.c_int32()
            chk(lib().KSPGetResidualHistory(self.h, C.byref(p), C.byref(na)))
            out["history"] = self._hist[:na.value].copy()
        return out

    def destroy(self):
        if self.h:
            chk(lib().KSPDestroy(C.byref(self.h)))

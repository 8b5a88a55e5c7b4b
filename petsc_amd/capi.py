"""ctypes binding of libpetschip.so (include/petschip.h).  No fallback: a missing library raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libpetschip.so")

_pI = C.POINTER(C.c_int32)
_pD = C.POINTER(C.c_double)
_vp = C.c_void_p

# every symbol include/petschip.h declares: (name, restype, argtypes)
SIGNATURES = [
    ("phip_init", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("phip_finalize", C.c_int, [_vp]),
    ("phip_last_error", C.c_char_p, []),
    ("phip_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("phip_device_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.c_char_p, C.c_int]),
    ("phip_malloc", C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    ("phip_free", C.c_int, [_vp, _vp]),
    ("phip_memcpy_h2d", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("phip_memcpy_d2h", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("phip_memcpy_d2d", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("phip_sync", C.c_int, [_vp]),
    ("phip_device_sync", C.c_int, [_vp]),
    ("phip_stream", _vp, [_vp]),
    ("phip_event_create", C.c_int, [_vp, C.POINTER(_vp)]),
    ("phip_event_record", C.c_int, [_vp, _vp]),
    ("phip_event_elapsed_ms", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    ("phip_event_destroy", C.c_int, [_vp, _vp]),
    ("phip_profile_enable", C.c_int, [_vp, C.c_int]),
    ("phip_profile_read", C.c_int, [_vp, C.c_int, _pI, _pD, _pD, _pD]),
    ("phip_profile_durations", C.c_int, [_vp, C.c_int, C.c_int32, C.POINTER(C.c_float), _pI]),
    ("phip_mat_create", C.c_int, [_vp, C.c_int32, C.c_int32, _pI, _pI, _pD, C.c_int, C.c_int32, _pI, _pI, C.POINTER(_vp)]),
    ("phip_mat_update_values", C.c_int, [_vp, _pD]),
    ("phip_mat_scale", C.c_int, [_vp, C.c_double]),
    ("phip_mat_shift", C.c_int, [_vp, C.c_double]),
    ("phip_mat_diagonal_scale", C.c_int, [_vp, _vp, _vp]),
    ("phip_mat_zero_entries", C.c_int, [_vp]),
    ("phip_mat_get_values", C.c_int, [_vp, _pD]),
    ("phip_mat_destroy", C.c_int, [_vp]),
    ("phip_mat_mult", C.c_int, [_vp, _vp, _vp]),
    ("phip_mat_multadd", C.c_int, [_vp, _vp, _vp, _vp]),
    ("phip_mat_transpose_setup", C.c_int, [_vp, _pI, _pI, _pD]),
    ("phip_mat_mult_transpose", C.c_int, [_vp, _vp, _vp]),
    ("phip_mat_mult_transpose_add", C.c_int, [_vp, _vp, _vp, _vp]),
    ("phip_mat_getdiagonal", C.c_int, [_vp, _vp]),
    ("phip_mat_plan_info", C.c_int, [_vp, _pI, _pI, _pI, C.POINTER(C.c_int64)]),
    ("phip_mat_plan_get", C.c_int, [_vp, _pI]),
    ("phip_mat_plan_lanes", C.c_int, [_vp, C.POINTER(C.c_int64)]),
    ("phip_mat_set_nontemporal", C.c_int, [_vp, C.c_int]),
    ("phip_set_ew_nontemporal", C.c_int, [C.c_int]),
    ("phip_mat_set_threads_per_row", C.c_int, [_vp, C.c_int32]),
    ("phip_set_default_mat_format", C.c_int, [C.c_int]),
    ("phip_mat_set_format", C.c_int, [_vp, C.c_int]),
    ("phip_mat_format_info", C.c_int, [_vp, C.POINTER(C.c_int), _pI, C.POINTER(C.c_int64)]),
    ("phip_set_default_value_dict", C.c_int, [C.c_int]),
    ("phip_mat_set_value_dict", C.c_int, [_vp, C.c_int]),
    ("phip_mat_value_dict_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _pI]),
    ("phip_set_default_diag_mask", C.c_int, [C.c_int]),
    ("phip_mat_set_diag_mask", C.c_int, [_vp, C.c_int]),
    ("phip_mat_diag_mask_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _pI]),
    ("phip_set_default_hot_columns", C.c_int, [C.c_int]),
    ("phip_mat_set_hot_columns", C.c_int, [_vp, C.c_int]),
    ("phip_mat_hot_columns_info", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), _pD, _pD]),
    ("phip_set_dmask_kernel", C.c_int, [C.c_int]),
    ("phip_set_xcd_chunk", C.c_int, [C.c_int]),
    ("phip_set_setup_threads", C.c_int, [C.c_int]),
    ("phip_set_cg_defer_x", C.c_int, [C.c_int]),
    ("phip_mat_mask_sets_info", C.c_int, [_vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ("phip_halo_ipc_count", C.c_int, [C.POINTER(C.c_ulonglong)]),
    ("phip_vec_dot", C.c_int, [_vp, C.c_int32, _vp, _vp, _pD]),
    ("phip_vec_norm", C.c_int, [_vp, C.c_int32, _vp, C.c_int, _pD]),
    ("phip_vec_mdot", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp), _pD]),
    ("phip_vec_maxpy", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _pD, C.POINTER(_vp)]),
    ("phip_vec_axpy", C.c_int, [_vp, C.c_int32, _vp, C.c_double, _vp]),
    ("phip_vec_aypx", C.c_int, [_vp, C.c_int32, _vp, C.c_double, _vp]),
    ("phip_vec_waxpy", C.c_int, [_vp, C.c_int32, _vp, C.c_double, _vp, _vp]),
    ("phip_vec_axpby", C.c_int, [_vp, C.c_int32, _vp, C.c_double, C.c_double, _vp]),
    ("phip_vec_axpbypcz", C.c_int, [_vp, C.c_int32, _vp, C.c_double, C.c_double, C.c_double, _vp, _vp]),
    ("phip_vec_pointwise_mult", C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    ("phip_vec_pointwise_divide", C.c_int, [_vp, C.c_int32, _vp, _vp, _vp]),
    ("phip_vec_scale", C.c_int, [_vp, C.c_int32, _vp, C.c_double]),
    ("phip_vec_copy", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("phip_vec_set", C.c_int, [_vp, C.c_int32, _vp, C.c_double]),
    ("phip_vec_swap", C.c_int, [_vp, C.c_int32, _vp, _vp]),
    ("phip_vec_reciprocal", C.c_int, [_vp, C.c_int32, _vp]),
    ("phip_vec_pointwise_mult_reduce", C.c_int, [_vp, C.c_int32, _vp, _vp, _vp, C.c_int, _vp]),
    ("phip_vec_maxpy_reduce", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, _pD, C.POINTER(_vp), C.c_int, _vp]),
    ("phip_mat_mult_dot", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("phip_mat_multadd_dot", C.c_int, [_vp, _vp, _vp, _vp, C.c_int, _vp]),
    ("phip_ticket_values", C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    ("phip_ticket_wait", C.c_int, [_vp, _vp, _pD]),
    ("phip_jacobi_setup", C.c_int, [_vp, C.c_int32, _vp, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    ("phip_vec_reduce_local", C.c_int, [_vp, C.c_int, C.c_int32, _vp, C.c_int32, C.POINTER(_vp), _vp]),
    ("phip_vec_reduce_begin", C.c_int, [_vp, C.c_int, C.c_int32, _vp, C.c_int32, C.POINTER(_vp), _vp]),
    ("phip_vec_reduce_local_host", C.c_int, [_vp, C.c_int, C.c_int32, _vp, C.c_int32, C.POINTER(_vp), _pD]),
    ("phip_read_scalars", C.c_int, [_vp, _vp, C.c_int32, _pD]),
    ("phip_vec_mdot_static", C.c_int, [_vp, C.c_int32, _vp, C.c_int32, C.POINTER(_vp), _vp]),
    ("phip_comm_create_callbacks", C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.POINTER(_vp)]),
    ("phip_comm_unique_id", C.c_int, [_vp]),
    ("phip_comm_create", C.c_int, [_vp, C.c_char_p, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    ("phip_comm_destroy", C.c_int, [_vp]),
    ("phip_comm_rank_size", C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("phip_comm_allreduce", C.c_int, [_vp, _vp, C.c_int32, C.c_int]),
    ("phip_comm_allreduce_host", C.c_int, [_vp, _pD, C.c_int32, C.c_int]),
    ("phip_comm_allgather_host", C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    ("phip_comm_alltoallv_i32", C.c_int, [_vp, _pI, _pI, _pI, _pI]),
    ("phip_comm_barrier", C.c_int, [_vp]),
    ("phip_halo_create", C.c_int, [_vp, C.c_int32, _pI, _pI, _pI, C.c_int32, _pI, _pI, C.POINTER(_vp)]),
    ("phip_halo_destroy", C.c_int, [_vp]),
    ("phip_halo_begin", C.c_int, [_vp, _vp, _vp]),
    ("phip_halo_end", C.c_int, [_vp]),
    ("phip_halo_reverse_begin", C.c_int, [_vp, _vp]),
    ("phip_halo_reverse_end", C.c_int, [_vp, _vp]),
]


class Ticket(C.Structure):
    _fields_ = [("slot", C.c_int32), ("nvals", C.c_int32), ("seq", C.c_uint64)]


def last_error():
    """Text of the calling thread's last libpetschip error."""
    return load().phip_last_error().decode()


class PhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libpetschip error %d: %s" % (code, msg))
        self.code = code


def load(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise ImportError("libpetschip.so not found at %s -- build it with `python -m petsc_amd.build` (there is no CPU fallback)" % path)
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)  # raises AttributeError if the ABI lost a symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def _ip(a):
    return a.ctypes.data_as(_pI) if a is not None else None


def _dp(a):
    return a.ctypes.data_as(_pD) if a is not None else None


class Context:
    """One GPU context + convenience wrappers that move numpy arrays through the C ABI (tests / bench)."""

    def __init__(self, device=0, lib=None):
        self.lib = lib or load()
        self.h = _vp()
        self.check(self.lib.phip_init(device, C.byref(self.h)))

    def check(self, rc):
        if rc:
            raise PhipError(rc, self.lib.phip_last_error().decode())

    def close(self):
        if self.h:
            self.lib.phip_finalize(self.h)
            self.h = _vp()

    def device_info(self):
        dev, cus, mem = C.c_int(), C.c_int(), C.c_size_t()
        name = C.create_string_buffer(64)
        self.check(self.lib.phip_device_info(self.h, C.byref(dev), C.byref(cus), C.byref(mem), name, 64))
        return dict(device=dev.value, cu_count=cus.value, hbm_bytes=mem.value, arch=name.value.decode())

    # -- memory
    def malloc(self, nbytes):
        p = _vp()
        self.check(self.lib.phip_malloc(self.h, nbytes, C.byref(p)))
        return p

    def free(self, p):
        self.check(self.lib.phip_free(self.h, p))

    def to_device(self, arr):
        arr = np.ascontiguousarray(arr)
        p = self.malloc(arr.nbytes)
        self.check(self.lib.phip_memcpy_h2d(self.h, p, arr.ctypes.data_as(_vp), arr.nbytes))
        return p

    def upload(self, p, arr):
        arr = np.ascontiguousarray(arr)
        self.check(self.lib.phip_memcpy_h2d(self.h, p, arr.ctypes.data_as(_vp), arr.nbytes))

    def to_host(self, p, n, dtype=np.float64):
        out = np.empty(n, dtype=dtype)
        self.check(self.lib.phip_memcpy_d2h(self.h, out.ctypes.data_as(_vp), p, out.nbytes))
        return out

    def sync(self):
        self.check(self.lib.phip_sync(self.h))

    # -- timing
    def time_region(self, fn, reps):
        """Device time (ms per call) of `reps` back-to-back calls of fn(), by hipEvents on the context's stream."""
        e0, e1 = _vp(), _vp()
        self.check(self.lib.phip_event_create(self.h, C.byref(e0)))
        self.check(self.lib.phip_event_create(self.h, C.byref(e1)))
        self.check(self.lib.phip_event_record(self.h, e0))
        for _ in range(reps):
            fn()
        self.check(self.lib.phip_event_record(self.h, e1))
        ms = C.c_float()
        self.check(self.lib.phip_event_elapsed_ms(self.h, e0, e1, C.byref(ms)))
        self.lib.phip_event_destroy(self.h, e0)
        self.lib.phip_event_destroy(self.h, e1)
        return ms.value / reps

    # -- matrix
    def mat_create(self, m, n, ai, aj, aa, cprow=None):
        ai = np.ascontiguousarray(ai, np.int32); aj = np.ascontiguousarray(aj, np.int32); aa = np.ascontiguousarray(aa, np.float64)
        A = _vp()
        if cprow is None:
            rc = self.lib.phip_mat_create(self.h, m, n, _ip(ai), _ip(aj), _dp(aa), 0, 0, None, None, C.byref(A))
        else:
            ci = np.ascontiguousarray(cprow[0], np.int32); cr = np.ascontiguousarray(cprow[1], np.int32)
            rc = self.lib.phip_mat_create(self.h, m, n, _ip(ai), _ip(aj), _dp(aa), 1, len(cr), _ip(ci), _ip(cr), C.byref(A))
        self.check(rc)
        return A

    def mat_plan(self, A):
        nb, cap, tpr, nnz = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64()
        self.check(self.lib.phip_mat_plan_info(A, C.byref(nb), C.byref(cap), C.byref(tpr), C.byref(nnz)))
        br = np.zeros(nb.value + 1, np.int32)
        self.check(self.lib.phip_mat_plan_get(A, _ip(br)))
        return dict(nblocks=nb.value, cap=cap.value, tpr=tpr.value, nnz=nnz.value, block_row=br)

    # -- reductions
    def dot(self, n, x, y):
        z = C.c_double()
        self.check(self.lib.phip_vec_dot(self.h, n, x, y, C.byref(z)))
        return z.value

    def norm(self, n, x, ntype=1):
        z = (C.c_double * 2)()
        self.check(self.lib.phip_vec_norm(self.h, n, x, ntype, z))
        return (z[0], z[1]) if ntype == 4 else z[0]

    def mdot(self, n, x, ys):
        nv = len(ys)
        arr = (_vp * max(nv, 1))(*ys)
        z = np.zeros(nv)
        self.check(self.lib.phip_vec_mdot(self.h, n, x, nv, arr, _dp(z)))
        return z

    def maxpy(self, n, x, alpha, ys):
        nv = len(ys)
        arr = (_vp * max(nv, 1))(*ys)
        al = np.ascontiguousarray(alpha, np.float64)
        self.check(self.lib.phip_vec_maxpy(self.h, n, x, nv, _dp(al), arr))

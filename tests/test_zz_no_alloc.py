"""GPU test of phip_vec_mdot_static, the entry point that allocates nothing (include/petschip.h): through the C ABI, against the existing
entry point phip_vec_reduce_local (kind 0) it is a variant of -- bit for bit (same kernel, grid and summation order) -- and against the
oracle's VecMDot_Seq restatement, on two calls in a row, the second with a smaller input."""
import ctypes as C

import numpy as np
import pytest

from petsc_amd import capi

pytestmark = pytest.mark.gpu
_vp = C.c_void_p


def _both(ctx, n, nv, x, ys, dz_static, dz_ref):
    """one call of each entry point on the first n entries; returns (static, existing) results"""
    L = ctx.lib
    arr = (_vp * nv)(*ys[:nv])
    ctx.check(L.phip_vec_mdot_static(L.phip_stream(ctx.h), n, x, nv, arr, dz_static))
    ctx.check(L.phip_vec_reduce_local(ctx.h, 0, n, x, nv, arr, dz_ref))
    return ctx.to_host(dz_static, nv), ctx.to_host(dz_ref, nv)


def test_mdot_static_matches_reduce_local(orc):
    ctx = capi.Context(0)
    L = ctx.lib
    rng = np.random.default_rng(77)
    n1, n2 = 1_300_001, 4_097          # odd lengths: the scalar tail runs; the second call is smaller and uses fewer workgroups
    hx = rng.uniform(-1, 1, n1)
    hys = [rng.uniform(-1, 1, n1) for _ in range(16)]
    x = ctx.to_device(hx)
    ys = [ctx.to_device(h) for h in hys]
    dz_s, dz_r = ctx.malloc(8 * 16), ctx.malloc(8 * 16)
    try:
        for n, nv in ((n1, 16), (n2, 5), (n1, 1), (0, 3), (n2, 16)):
            got, ref = _both(ctx, n, nv, x, ys, dz_s, dz_r)
            assert np.array_equal(got, ref), (n, nv, got, ref)                      # the existing entry point's bits
            want = orc.mdot(hx[:n], [h[:n] for h in hys[:nv]])                       # VecMDot_Seq, dvec2.c:98-294
            scale = np.array([np.abs(hx[:n] * h[:n]).sum() for h in hys[:nv]])
            assert np.all(np.abs(got - want) <= 1e-13 * scale + 1e-300), (n, nv)
        # sizes beyond the fixed array are rejected with a clear error, and nothing was launched
        arr = (_vp * 17)(*([ys[0]] * 17))
        rc = L.phip_vec_mdot_static(L.phip_stream(ctx.h), n2, x, 17, arr, dz_s)
        assert rc == 63 and b"1..16" in L.phip_last_error()
        assert L.phip_vec_mdot_static(L.phip_stream(ctx.h), n2, x, 0, arr, dz_s) == 63
        # and the array is still clean afterwards
        got, ref = _both(ctx, n2, 2, x, ys, dz_s, dz_r)
        assert np.array_equal(got, ref)
    finally:
        for p in [x, dz_s, dz_r] + ys:
            ctx.free(p)
        ctx.close()

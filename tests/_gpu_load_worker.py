"""Worker of tests/test_kernels_gpu.py::test_reductions_under_foreign_load: a second PROCESS that keeps GPU 0 busy for argv[1] seconds with
bandwidth-bound launches of its own (large AXPYs and many-vector dot products), so that the reduction kernels of the test process share
the device with foreign work while their collector workgroup waits for the partial results of the others (phip_reduce.h)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    seconds = float(sys.argv[1])
    from petsc_amd import capi
    ctx = capi.Context(0)
    n = 16 << 20
    xs = [ctx.to_device(np.full(n, 1.0 + k)) for k in range(9)]
    print("load running", flush=True)
    t0 = time.time()
    launches = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            ctx.lib.phip_vec_axpy(ctx.h, n, xs[0], 1e-9, xs[1])
            ctx.lib.phip_stream(ctx.h)
        z = ctx.mdot(n, xs[0], xs[1:])
        assert np.all(np.isfinite(z))
        launches += 21
    for p in xs:
        ctx.free(p)
    ctx.close()
    print("load done after %d launches" % launches, flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Do the streams of an elementwise kernel collide in the memory system when the vectors lie a power of two apart?  z = a x + b y + c z
(3 reads, 1 write, 32 B per entry) on vectors of n = 2^24 doubles placed at base + k (n + skew) doubles inside one allocation.
   python scripts/exp_vec_skew.py"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from petsc_amd import capi  # noqa: E402

ctx = capi.Context(0)
L = ctx.lib
n = 1 << 24
maxskew = 1 << 20
base = ctx.to_device(np.random.default_rng(1).uniform(-1.0, 1.0, 3 * (n + maxskew) + 16))  # every page written: reads of untouched memory are not HBM reads
for skew in (0, 32, 512, 2048 + 32, 8192 + 512, 65536 + 2048, (1 << 19) + 8192 + 32):
    p = [C.c_void_p(base.value + 8 * k * (n + skew)) for k in range(3)]
    f = lambda: L.phip_vec_axpbypcz(ctx.h, n, p[2], 0.5, 0.25, 0.125, p[0], p[1])
    for _ in range(5):
        f()
    ctx.sync()
    ms = min(ctx.time_region(f, 50) for _ in range(3))
    print(json.dumps({"skew_doubles": skew, "us": round(ms * 1e3, 2), "GBps": round(32 * n / ms / 1e6, 1)}), flush=True)
ctx.free(base)

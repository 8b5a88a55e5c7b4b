#!/usr/bin/env python3
"""bench.py -- the judged benchmark of the sparse-solve hot path (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload box27_256|lap7_128|lap7_<n>|box27_<n>] [--scaling strong|weak]
                    [--legs auto|none|name,name] [--no-cpu-baseline]
    python bench.py --gpus N ...           starts its N ranks itself (one process per GPU); under torch.distributed.run it takes the ranks given
    --workload box27_256_var               the same system with a value of its own in every stored entry (no dictionary: SELL-dict streamed / CSR)

A "step" is one KSPCG + PCJACOBI iteration through MATAIJHIP / VECHIP (cg.c:180-275: MatMult, 2 VecTDot, VecNorm, 2 VecAXPY,
VecAYPX, PCApply), i.e. one pass of the whole hot path.  The K timed steps are ONE KSPSolve capped at K iterations (rtol 0), so
the timed region also contains KSPSolve's set-up (VecSet, VecCopy, one PCApply, VecNorm, VecTDot): extra work, never less.

Default workload, at every N: BASELINE.json configs[2], the north star -- KSPCG+PCJACOBI on the 3-D 27-point 256^3 problem, STRONG
scaling: the same global matrix cut into MPIAIJ row blocks (z-slabs), ghost exchange and reductions between the ranks.  It fits one
GPU, so N = 1 runs it too and value(N) / value(1) is the strong-scaling speed-up the >= 6x target is quoted on.

value    = whole-job bytes the K iterations REALLY move through HBM / wall time (GB/s): per rank and iteration the SpMV's bytes in its device
           layout (x and y once each; + the off-diagonal block's MultAdd 12 nzB + 28 mB on N > 1) + 80 m of the two fused vector launches.
           Never above N x 8 TB/s.  The algorithmic bytes of SURVEY.md 8d (12 nz + 20 m + 4 + 136 m: what the reference's passes move)
           ride along as algorithmic_bytes_per_iteration, KSPCG iterations/s as cg_iters_per_s.
roofline = the kernel with the largest share of the timed iteration, priced on the bytes it REALLY moves in its device layout (never
           more than the algorithmic CSR bytes) over its average launch time: frac <= 1 by construction.  Every other kernel of the
           iteration and every leg carries an object of the same shape under "kernels" / "legs".
legs     (N = 1, rank 0): the other single-GPU measurements the metric names --
           lap7_128      BASELINE configs[1]: 7-point 128^3 SpMV (the >= 70 % target) + KSPCG it/s; the working set fits the 256 MiB
                         Infinity Cache, which the leg says
           lap7_256_var  7-point 256^3 with per-entry perturbed values (what a variable-coefficient / FEM assembly gives): no value
                         dictionary applies, the matrix streams 9 B/nz (SELL-dict) from HBM, 1.5 GB per launch, far outside the caches
                         ("roofline"), and the same matrix through the general CSR row-block kernel ("roofline_csr", 12 B/nz, any matrix);
                         KSPCG+PCJACOBI iterations/s through both
           box27_256_var the north-star system (27-point 256^3) the same way: what a MatSetValues-assembled operator gets
           rmat_22       BASELINE config 5's matrix family (R-MAT) through the CSR row-block kernel
cpu_baseline = the oracle (C restatement of the reference loops, -O3 -march=native) on the same system: 16 row blocks / threads as
           MATMPIAIJ would run it on the host's CPU share, and one core (MATSEQAIJ); rank 0 at N = 1 only, bounded sample.
Timing: hipEvents on the library's compute stream.  avg = K launches back to back between two events (best of 3); p10/p50/p90 = the
same launches bracketed one by one (each figure then carries the event pair's own ~2-3 us).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec; ~6.3 TB/s is what a streaming copy achieves
READ_STREAM_CEILING_GBPS = 7100.0  # measured, profiles/r03_micro_read_stream.jsonl (7.0 - 7.17 TB/s); context for the fractions, never the peak
REPS = 200


def parse_workload(name):
    """lap7_<n> | box27_<n> [+ _var: every stored entry gets a value of its own] -> (kind, n, perturb)"""
    parts = name.split("_")
    if len(parts) not in (2, 3) or parts[0] not in ("lap7", "box27") or not parts[1].isdigit() or (len(parts) == 3 and parts[2] != "var"):
        raise SystemExit("unknown workload %s" % name)
    return parts[0], int(parts[1]), len(parts) == 3


def perturb_values(ai, aj, aa, rs):
    """every stored entry gets a value of its own (what a variable-coefficient / FEM assembly gives: nothing for a value dictionary to
    find), symmetric and diagonally dominant like the stencil it starts from: a(i,j) and a(j,i) get the same factor from a hash of
    (min, max) of the GLOBAL indices -- independent of the partition -- off-diagonal entries shrink by up to 1e-3, diagonal entries grow."""
    m = len(ai) - 1
    row = np.repeat(np.arange(rs, rs + m, dtype=np.int64), np.diff(ai).astype(np.int64))
    lo = np.minimum(row, aj).astype(np.uint64)
    hi = np.maximum(row, aj).astype(np.uint64)
    diag = row == aj
    del row
    k = (lo * np.uint64(0x9E3779B97F4A7C15) + hi * np.uint64(0xC2B2AE3D27D4EB4F))
    del lo, hi
    k ^= k >> np.uint64(29)
    k *= np.uint64(0xBF58476D1CE4E5B9)
    delta = (k >> np.uint64(40)).astype(np.float64) * (1e-3 / float(1 << 24))
    del k
    return aa * np.where(diag, 1.0 + delta, 1.0 - delta)


def local_rows(kind, n, size, rank, scaling, perturb=False):
    """CSR of this rank's rows (global sorted columns).  strong: n^3 grid; weak: n x n x (n*size) grid, one n^3 slab per rank."""
    from petsc_amd import generators as G
    nzdim = n * size if scaling == "weak" else n
    N = n * n * nzdim
    m = N // size + (1 if (N % size) > rank else 0)
    rs = sum(N // size + (1 if (N % size) > r else 0) for r in range(rank))
    # = G.laplace3d_7pt(..., dirichlet_rows=False) / G.box3d_27pt(...), built by the C loop of the host library (tests/test_host_cpu.py)
    ai, aj, aa = G.stencil_rows_native(7 if kind == "lap7" else 27, n, n, nzdim, rs, rs + m)
    if perturb:
        aa = perturb_values(ai, aj, aa, rs)
    return N, m, rs, ai, aj, aa


def pct(v, q):
    return float(np.percentile(np.asarray(v, dtype=np.float64), q))


class Timer:
    """hipEvent timing on the library's compute stream (phip_event_*)."""

    def __init__(self, L, ctx):
        self.L, self.ctx = L, ctx

    def _ev(self):
        e = C.c_void_p()
        rc = self.L.phip_event_create(self.ctx, C.byref(e))
        assert rc == 0, self.L.phip_last_error()
        return e

    def region(self, fn, reps):
        e0, e1 = self._ev(), self._ev()
        self.L.phip_event_record(self.ctx, e0)
        for _ in range(reps):
            fn()
        self.L.phip_event_record(self.ctx, e1)
        ms = C.c_float()
        rc = self.L.phip_event_elapsed_ms(self.ctx, e0, e1, C.byref(ms))
        assert rc == 0, self.L.phip_last_error()
        self.L.phip_event_destroy(self.ctx, e0); self.L.phip_event_destroy(self.ctx, e1)
        return ms.value / reps

    def each(self, fn, reps):
        ev = [self._ev() for _ in range(reps + 1)]
        for i in range(reps):
            self.L.phip_event_record(self.ctx, ev[i])
            fn()
        self.L.phip_event_record(self.ctx, ev[reps])
        out, ms = [], C.c_float()
        self.L.phip_event_elapsed_ms(self.ctx, ev[0], ev[reps], C.byref(ms))  # synchronises on the last event
        for i in range(reps):
            self.L.phip_event_elapsed_ms(self.ctx, ev[i], ev[i + 1], C.byref(ms))
            out.append(ms.value)
        for e in ev:
            self.L.phip_event_destroy(self.ctx, e)
        return out

    def cold(self, fn, reps=24):
        """average us of fn()'s launch when nothing it reads is left in L2 / Infinity Cache: before every bracketed launch a VecDot over two
        512 MiB vectors reads 1 GiB through the caches (read-only, so the launch does not pay for the sweep's write-backs; outside the event pair)"""
        from petsc_amd.capi import Context
        n = 1 << 26
        if getattr(self, "_sweep", None) is None:
            dev = Context.__new__(Context)
            dev.lib, dev.h = self.L, self.ctx
            self._sweep = (dev, dev.malloc(8 * n), dev.malloc(8 * n))
            self.L.phip_vec_set(self.ctx, n, self._sweep[1], 1.0)
            self.L.phip_vec_set(self.ctx, n, self._sweep[2], 0.0)
        _, sx, sy = self._sweep
        ev = [(self._ev(), self._ev()) for _ in range(reps)]
        for e0, e1 in ev:
            z = C.c_double()
            rc = self.L.phip_vec_dot(self.ctx, n, sx, sy, C.byref(z))
            assert rc == 0, self.L.phip_last_error()
            self.L.phip_event_record(self.ctx, e0)
            fn()
            self.L.phip_event_record(self.ctx, e1)
        out, ms = [], C.c_float()
        for e0, e1 in ev:
            rc = self.L.phip_event_elapsed_ms(self.ctx, e0, e1, C.byref(ms))
            assert rc == 0, self.L.phip_last_error()
            out.append(ms.value * 1e3)
            self.L.phip_event_destroy(self.ctx, e0); self.L.phip_event_destroy(self.ctx, e1)
        return float(np.mean(out[2:])), [round(pct(out[2:], q), 2) for q in (10, 50, 90)]

    def close(self):
        if getattr(self, "_sweep", None) is not None:
            dev, sx, sy = self._sweep
            dev.free(sx); dev.free(sy)
            self._sweep = None

    def kernel(self, fn, reps=REPS, warm=20):
        """-> (avg_us back to back, [p10, p50, p90] us of individually bracketed launches)"""
        for _ in range(warm):
            fn()
        self.L.phip_sync(self.ctx)
        avg = min(self.region(fn, reps) for _ in range(3)) * 1e3
        d = self.each(fn, reps)
        return avg, [round(pct(d, q) * 1e3, 2) for q in (10, 50, 90)]


def layout_of(L, hA, m, n, nz):
    """(kernel name, layout text, bytes one launch really moves, algorithmic CSR bytes) of the device layout the library chose."""
    f_, nd_, pad_ = C.c_int(), C.c_int32(), C.c_int64()
    L.phip_mat_format_info(hA, C.byref(f_), C.byref(nd_), C.byref(pad_))
    vdb_, vdu_, vdn_ = C.c_int(), C.c_int(), C.c_int32()
    L.phip_mat_value_dict_info(hA, C.byref(vdb_), C.byref(vdu_), C.byref(vdn_))
    dmb_, dmu_, dmn_ = C.c_int(), C.c_int(), C.c_int32()
    L.phip_mat_diag_mask_info(hA, C.byref(dmb_), C.byref(dmu_), C.byref(dmn_))
    hb_, hu_ = C.c_int(), C.c_int()
    L.phip_mat_hot_columns_info(hA, C.byref(hb_), C.byref(hu_), None, None)
    alg = 12 * nz + 20 * m + 4
    nsl = (m + 511) // 512
    if f_.value != 2:
        nb = C.c_int32()
        L.phip_mat_plan_info(hA, C.byref(nb), None, None, None)
        real = 12 * nz + 4 * (m + 1) + 8 * n + 8 * m + 5 * nb.value + (20 * n if hu_.value else 0)
        return "spmv_rowblock_kernel", "CSR row blocks (12 B/nz)" + (", x gathered in hot-column order (+20 B/column)" if hu_.value else ""), real, alg
    if dmu_.value:
        if m >= (4 << 20) or dmn_.value >= 12:  # the library's rule (phip_set_dmask_kernel)
            ns_, nw_ = C.c_int64(), C.c_int64()
            L.phip_mat_mask_sets_info(hA, C.byref(ns_), C.byref(nw_))
            return ("spmv_dmaskg_kernel", "diagonal masks: %d (diagonal, value) pairs in 64-byte group descriptors; per wave of 256 rows and pair four 64-bit lane masks, equal sets stored once (%d sets of 33 x 32 B for %d waves + 4 B per wave), four rows a lane" % (dmn_.value, ns_.value, nw_.value),
                    33 * 32 * ns_.value + 4 * nw_.value + 8 * n + 8 * m, alg)
        return ("spmv_dmask4_kernel", "diagonal masks: %d (diagonal, value) pairs, a 32-bit set of them per row (4 B/row), four rows a lane" % dmn_.value,
                4 * 512 * nsl + 8 * n + 8 * m, alg)
    if vdu_.value:
        return ("spmv_pair_kernel", "SELL-dict with a pair dictionary of %d (diagonal, value) pairs (1 B/nz)" % vdn_.value,
                ((pad_.value + 3) // 4) * 4 + m + 8 * nsl + 8 * n + 8 * m, alg)
    return "spmv_sell_kernel", "SELL-dict, streamed values (8 B value + 1 B diagonal code per stored entry)", 9 * pad_.value + m + 8 * nsl + 8 * n + 8 * m, alg


def source_sha16():
    """sha256 (first 16 hex digits) over the kernel sources the running library was built from: committed counter files carry the same
    figure (scripts/summarize_profiles.py), so a figure measured on other kernels of the same name is flagged, not passed on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "petsc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


TRAFFIC_FILE = "profiles/r03_hbm_traffic.json"
_traffic = None


def traffic_for(workload, kernel):
    """HBM bytes per launch from a committed rocprofv3 --pmc measurement of THIS workload and kernel (profiles/r03_hbm_traffic.json,
    written by scripts/collect_profiles.sh on the GPU box), or None.  Offline by nature: a process cannot count its own HBM traffic.
    -> (bytes, source text, stale?) ; stale = the file was measured on other kernel sources than the ones this library was built from."""
    global _traffic
    if _traffic is None:
        try:
            _traffic = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
            _traffic["_stale"] = _traffic.get("_source_sha16") != source_sha16()
        except (OSError, ValueError):
            _traffic = {}
    e = _traffic.get("%s:%s" % (workload, kernel))
    if not e:
        return None, None, None
    return (e.get("bytes_per_launch"), "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, 2 x FETCH + WRITE; %s; profiled launch %.1f us)"
            % (TRAFFIC_FILE, e.get("source", ""), e.get("kernel_stats_avg_us") or float("nan")), bool(_traffic.get("_stale")))


def roofline_obj(workload, kernel, layout, real_bytes, alg_bytes, avg_us, pcts, resident_bytes=None, extra=None):
    """achieved = bytes the kernel REALLY moves in its device layout (x and y once each; never more than the algorithmic CSR bytes
    12 nz + 20 m + 4 of SURVEY.md 8d) / average launch time, so no figure exceeds the HBM peak.  Where the layout streams fewer bytes than
    CSR the ratio is reported as algorithmic_over_layout_bytes (no GB/s is quoted on bytes that never moved)."""
    ach = real_bytes / (avg_us * 1e-6) / 1e9
    traffic, src, stale = traffic_for(workload, kernel)
    o = {"bound": "hbm", "kernel": kernel, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
         "traffic": traffic, "traffic_source": src, "traffic_stale": stale, "bytes_per_launch": int(real_bytes), "avg_launch_us": round(avg_us, 2), "launch_us_p10_p50_p90": pcts,
         "launches_timed": REPS, "layout": layout,
         "algorithmic_bytes_per_launch": int(alg_bytes), "algorithmic_over_layout_bytes": round(alg_bytes / float(real_bytes), 3),
         "achieved_is": "bytes the kernel moves in its device layout (x and y once each) / average launch time",
         # context, not a peak: what a bare read-only stream of 4 GiB reaches on this part in the shape these kernels read their
         # matrices in (scripts/micro/read_stream.hip, committed result profiles/r03_micro_read_stream.jsonl); frac stays against 8 TB/s
         "read_stream_ceiling_GBps": READ_STREAM_CEILING_GBPS}
    if alg_bytes <= 1.001 * real_bytes:
        o["streams_the_csr"] = True  # the SURVEY.md 8d figure and the real figure coincide: frac IS the north star's "% of HBM peak on 12 nz + 20 m + 4"
    if resident_bytes is not None and resident_bytes < (240 << 20):
        o["cache_note"] = "working set %.0f MB fits the 256 MiB Infinity Cache: back-to-back launches are served on-die, achieved is cache bandwidth, not HBM (cold_launch_us: the same launch behind a 1 GiB read sweep)" % (resident_bytes / 1e6)
    if extra:
        o.update(extra)
    return o


def vec_kernels(L, ctx, tm, m, workload="", incg_batch=None, incg_xp=None):
    """the two vector launches of a KSPCG iteration: cg_update_kernel = VecAXPY(R,-a,W) + VecPointwiseMult(Z,D,R) with its three fused sums
    (cg.c:223-229), and xp_update_kernel = VecAXPY(X,a,P) (cg.c:222, deferred) + VecAYPX(P,b,Z) (cg.c:202).  Back to back: the pair, and
    the second kernel alone (the first cannot be launched alone through the API: it exists only as what is left of KSPCG's batch);
    incg_*: their per-launch durations inside the timed solve (phip_profile_durations kinds 2 and 3), which is what avg_launch_us
    reports when present."""
    from petsc_amd.capi import Context, Ticket
    dev = Context.__new__(Context)
    dev.lib, dev.h = L, ctx
    vecs = [dev.to_device(np.linspace(0.1, 1.0, m) * (k + 1)) for k in range(5)]
    vx, vp, vr, vw, vd = vecs
    vz = vw  # KSPCG keeps W and Z in one work vector (cg.c:106)
    tk = Ticket()

    def pair():
        L.phip_vec_axpy(ctx, m, vx, 1e-3, vp)
        L.phip_vec_axpy(ctx, m, vr, -1e-3, vw)
        L.phip_vec_pointwise_mult_reduce(ctx, m, vz, vd, vr, 0, C.byref(tk))
        L.phip_vec_aypx(ctx, m, vp, 0.5, vz)
        L.phip_stream(ctx)  # launches what is pending, as the next MatMult does

    def xp():
        L.phip_vec_axpy(ctx, m, vx, 1e-3, vp)
        L.phip_vec_aypx(ctx, m, vp, 0.5, vz)
        L.phip_stream(ctx)

    pair_us, _ = tm.kernel(pair, REPS, 10)
    xp_us, xp_pcts = tm.kernel(xp, REPS, 10)
    out = {}
    for name, kern, nbytes, what, b2b, pcts, incg in (
            ("cg_batch", "cg_update_kernel", 40 * m, "VecAXPY(R,-a,W) + VecPointwiseMult(Z,D,R) with 3 fused sums, one launch: 3 vectors read, 2 written", pair_us - xp_us, None, incg_batch),
            ("xp_update", "xp_update_kernel", 40 * m, "VecAXPY(X,a,P) (deferred) + VecAYPX(P,b,Z), one launch: 3 vectors read, 2 written", xp_us, xp_pcts, incg_xp)):
        if incg:
            avg = float(np.mean(incg)) * 1e3
            p3 = [round(pct(incg, q) * 1e3, 2) for q in (10, 50, 90)]
            src = "inside the timed KSPCG solve (hipEvent pair per launch, %d launches)" % len(incg)
        else:
            avg, p3, src = b2b, pcts, "back to back"
        out[name] = roofline_obj(workload, kern, "contiguous double[m]", nbytes, nbytes, avg, p3, resident_bytes=nbytes,
                                 extra={"what": what, "timed": src, "back_to_back_us": round(b2b, 2),
                                        "back_to_back_is": "pair of both vector kernels minus the second alone" if name == "cg_batch" else "200 launches between two events"})
    for v in vecs:
        dev.free(v)
    return out


def rmat_leg(P, L, comm, tm, name, scale, ranks=0):
    """single-GPU leg: SpMV of BASELINE config 5's matrix family (R-MAT, a,b,c,d = 0.57,0.19,0.19,0.05, 16 edges per row, duplicates summed) at a
    scale that generates in seconds; the product is the CSR row-block kernel's (irregular matrices have no stencil layout)"""
    from petsc_amd import generators as G
    t0 = time.time()
    ai, aj, aa = G.rmat(scale)
    gen_s = time.time() - t0
    m, nz = len(ai) - 1, int(ai[-1])
    lens = np.diff(ai)
    t0 = time.time()
    A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
    setup_s = time.time() - t0
    hA = A.device_mat(0)
    x, y = A.create_vecs()
    x.set_array(np.linspace(-1.0, 1.0, m))
    dx, dy = x.device_ptr(), y.device_ptr()
    per = [int(lens[(m * r) // 8:(m * (r + 1)) // 8].sum()) for r in range(8)]
    out = {"matrix": "R-MAT scale %d (BASELINE config 5's generator; config 5 itself is scale 24)" % scale, "rows": m, "nnz": nz, "generate_s": round(gen_s, 1),
           "setup_s": round(setup_s, 2), "max_row": int(lens.max()), "empty_rows": int((lens == 0).sum()),
           "nnz_imbalance_8_row_blocks": round(max(per) / (nz / 8.0), 2)}
    kernel, layout, real, alg = layout_of(L, hA, m, m, nz)
    avg, pcts = tm.kernel(lambda: L.phip_mat_mult(hA, dx, dy))
    out["roofline"] = roofline_obj(name, kernel, layout, real, alg, avg, pcts, resident_bytes=real)
    out["gathers_per_s"] = round(nz / (avg * 1e-6))
    out["bound_note"] = ("one divergent 8-byte gather per nonzero: a CU serves an L1-missing gather lane in about 3 clocks (207-244 G/s chip-wide on an L2-resident "
                         "table, scripts/micro/gather_policy.hip; 54 G/s on a 128 MiB table whatever the load's cache policy), so frac prices the kernel against a "
                         "roofline that does not bind it: with every gather an L2 hit and the fabric traffic at the algorithmic bytes (round 3's routing experiment, "
                         "profiles/r03_pmc_rmat22_route.json) the product of this matrix still takes 401 us (DESIGN.md section 3)")
    for o in (x, y, A):
        o.destroy()
    if ranks:
        out["mpiaij_%d_row_blocks" % ranks] = rank_report(P, L, ai, aj, aa, ranks)
    return out


def rank_report(P, L, ai, aj, aa, nranks=8):
    """What each of `nranks` MPIAIJ row blocks of this matrix holds and costs -- the part of the 8-GPU run a one-GPU box can bound: the
    ranks run as host threads sharing the GPU (the "threads" communicator: the real partition, assembly products, ghost plan and device
    blocks; not a performance path), and each rank in turn, alone on the GPU, times the kernel of its diagonal block (A x) and of its
    off-diagonal block (y += B lvec).  mpiaij.c:984-998, mmaij.c:69-101."""
    import struct
    import threading
    M = len(ai) - 1
    key = struct.pack("<Q", 0x72616e6b00000000 | (os.getpid() & 0xffffffff)) + bytes(120)
    out, err = [None] * nranks, [None] * nranks
    turn = threading.Lock()
    bar = threading.Barrier(nranks)

    def work(r):
        comm = None
        try:
            comm = P.Comm(device=0, backend="threads", id128=key, nranks=nranks, rank=r)
            A = P.Mat.from_global_csr(comm, ai, aj, aa)
            tmr = Timer(L, comm.ctx)
            hA, hB = A.device_mat(0), A.device_mat(1)
            blkA, blkB = A.block(0), A.block(1)
            gh = A.ghosts()
            m, ec = blkA["m"], len(gh["garray"])
            dev = capi.Context.__new__(capi.Context)
            dev.lib, dev.h = L, comm.ctx
            dx, dy, dl = dev.to_device(np.linspace(-1.0, 1.0, max(m, 1))), dev.to_device(np.zeros(max(m, 1))), dev.to_device(np.linspace(-1.0, 1.0, max(ec, 1)))
            hb, hu = C.c_int(), C.c_int()
            L.phip_mat_hot_columns_info(hB, C.byref(hb), C.byref(hu), None, None)
            bar.wait()
            with turn:   # one rank at a time: the others leave the GPU alone
                ta, _ = tmr.kernel(lambda: L.phip_mat_mult(hA, dx, dy), reps=30, warm=5)
                tb, _ = tmr.kernel(lambda: L.phip_mat_multadd(hB, dl, dy, dy), reps=30, warm=5) if blkB["nz"] else (0.0, None)
            out[r] = {"rank": r, "rows": m, "nnz_diag_block": blkA["nz"], "nnz_offdiag_block": blkB["nz"], "ghost_entries": ec, "entries_sent": int(len(gh["to_indices"])),
                      "neighbours": int(len(gh["from_procs"])), "offdiag_compressed_rows": bool(blkB["cprow_use"]), "offdiag_hot_column_order": bool(hu.value),
                      "diag_block_kernel_us": round(ta, 1), "offdiag_block_kernel_us": round(tb, 1), "exchange_bytes": 8 * (ec + int(len(gh["to_indices"])))}
            bar.wait()
            for p_ in (dx, dy, dl):
                dev.free(p_)
            A.destroy()
        except BaseException as e:  # noqa: BLE001
            err[r] = e
            bar.abort()
        finally:
            if comm is not None and err[r] is None:
                comm.finalize()

    from petsc_amd import capi
    ts = [threading.Thread(target=work, args=(r,)) for r in range(nranks)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(600)
    for e in err:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    if any(o is None for o in out):
        raise RuntimeError("rank report: a rank did not finish")
    tot = [o["diag_block_kernel_us"] + o["offdiag_block_kernel_us"] for o in out]
    nn = [o["nnz_diag_block"] + o["nnz_offdiag_block"] for o in out]
    return {"ranks": nranks, "per_rank": out,
            "kernel_us_slowest_over_mean": round(max(tot) / (sum(tot) / nranks), 2), "kernel_us_slowest": round(max(tot), 1), "kernel_us_mean": round(sum(tot) / nranks, 1),
            "nnz_slowest_over_mean": round(max(nn) / (sum(nn) / nranks), 2),
            "offdiag_share_of_nnz": round(sum(o["nnz_offdiag_block"] for o in out) / float(sum(nn)), 3),
            "offdiag_share_of_kernel_time": round(sum(o["offdiag_block_kernel_us"] for o in out) / sum(tot), 3),
            "exchange_bytes_largest_rank": max(o["exchange_bytes"] for o in out),
            "is": "the MPIAIJ partition of this matrix over %d row blocks (the %d-GPU run's), each rank's two kernels timed alone on one GPU; an upper bound of the "
                  "%d-GPU MatMult's kernel time is kernel_us_slowest, its ghost exchange moves exchange_bytes per rank and product" % (nranks, nranks, nranks)}


def spmv_leg(P, L, comm, tm, name, kind, n, perturb, with_cg, steps, fmt_csr_too=False, cold=False):
    """single-GPU leg: SpMV roofline of the layout the library picks (and, fmt_csr_too, of the CSR row-block kernel on the same matrix);
    with_cg: KSPCG+PCJACOBI iterations/s through each of them; cold: the launch also timed behind a cache sweep"""
    t0 = time.time()
    N, m, rs, ai, aj, aa = local_rows(kind, n, 1, 0, "strong", perturb)
    nz = int(ai[-1])
    A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
    del ai, aj, aa
    setup_s = time.time() - t0
    hA = A.device_mat(0)
    x, y = A.create_vecs()
    x.set_array(np.linspace(-1.0, 1.0, m))
    dx, dy = x.device_ptr(), y.device_ptr()
    out = {"matrix": "3-D %s-point %s %d^3%s" % ("7" if kind == "lap7" else "27", "Laplacian" if kind == "lap7" else "box stencil", n,
                                                  ", every stored entry with a value of its own (symmetric, <= 1e-3 relative; no value dictionary applies)" if perturb else ""),
           "rows": m, "nnz": nz, "setup_s": round(setup_s, 2)}
    b = None
    if with_cg:
        b = y.duplicate()

    def cg_rate(alg):
        x.set(1.0)
        A.mult(x, b)
        ksp = P.KSP(comm, "cg", "jacobi")
        ksp.set_operators(A)
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=max(steps // 4, 5))
        ksp.solve(b, x)
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=steps)
        L.phip_device_sync(comm.ctx)
        t1 = time.perf_counter()
        res = ksp.solve(b, x)
        L.phip_device_sync(comm.ctx)
        dt = time.perf_counter() - t1
        assert res["its"] == steps, res
        ksp.destroy()
        x.set_array(np.linspace(-1.0, 1.0, m))
        return round(steps / dt, 1)

    def one(key, suffix):
        kernel, layout, real, alg = layout_of(L, hA, m, m, nz)
        avg, pcts = tm.kernel(lambda: L.phip_mat_mult(hA, dx, dy))
        extra = None
        if cold:
            cavg, cp = tm.cold(lambda: L.phip_mat_mult(hA, dx, dy))
            extra = {"cold_launch_us": round(cavg, 2), "cold_launch_us_p10_p50_p90": cp, "cold_frac": round(real / (cavg * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4),
                     "cold_is": "the same launch with matrix, x and y evicted (1 GiB read through the caches before every bracketed launch): every byte comes from HBM"}
        out[key] = roofline_obj(name, kernel, layout, real, alg, avg, pcts, resident_bytes=real, extra=extra)
        if with_cg:
            out["cg_iters_per_s" + suffix] = cg_rate(alg)
            out["cg_real_GBps" + suffix] = round((real + 80 * m) * out["cg_iters_per_s" + suffix] / 1e9, 1)

    one("roofline", "")
    if fmt_csr_too:
        assert L.phip_mat_set_format(hA, 1) == 0
        one("roofline_csr", "_csr")
        assert L.phip_mat_set_format(hA, 0) == 0
    if with_cg:
        out["cg_real_GBps_is"] = "(bytes the SpMV moves in its layout + 80 m of the two fused vector launches) x iterations/s"
        b.destroy()
    for o in (x, y, A):
        o.destroy()
    return out


def run_leg(P, L, comm, tm, leg, steps):
    """one single-GPU leg by name: lap7_128 | lap7_<n>[_var] | box27_<n>[_var] | rmat_<scale>"""
    if leg == "lap7_128":
        o = spmv_leg(P, L, comm, tm, leg, "lap7", 128, False, True, max(steps, 20), fmt_csr_too=True, cold=True)
        o["is"] = ("BASELINE configs[1]: SeqAIJ SpMV + CG on the 3-D 7-point Laplacian 128^3 (target: SpMV >= 70 % of HBM peak on 12 nz + 20 m + 4); "
                   "roofline = the layout the library picks for this matrix, roofline_csr = the general CSR row-block kernel on it, which streams exactly those bytes "
                   "(its frac is the target's figure; cold_frac the same with nothing left in the caches)")
    elif leg.startswith("rmat_"):
        o = rmat_leg(P, L, comm, tm, leg, int(leg.split("_")[1]), ranks=8)
        o["is"] = "BASELINE config 5's matrix family: irregular (power-law) SpMV through the CSR row-block kernel; mpiaij_8_row_blocks = what each rank of the 8-GPU partition holds and costs"
    else:
        k2, n2, pv = parse_workload(leg)
        o = spmv_leg(P, L, comm, tm, leg, k2, n2, pv, True, max(min(steps, 100), 20), fmt_csr_too=pv)
        if pv:
            o["is"] = ("out-of-cache, non-dictionary path -- what a MatSetValues-assembled variable-coefficient operator gets (aij.c:430-501, 1313-1320): SpMV + "
                       "KSPCG+PCJACOBI through SELL-dict with streamed values (roofline, cg_iters_per_s) and through the general CSR row-block kernel "
                       "(roofline_csr, cg_iters_per_s_csr) on the same matrix")
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="box27_256")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"])
    ap.add_argument("--legs", default="auto", help="auto (N=1: lap7_128,lap7_256_var,box27_256_var,rmat_22) | none | comma-separated names (lap7_<n>, box27_<n>, rmat_<scale>)")
    ap.add_argument("--ksp", default="cg", choices=["cg", "pipecg"], help="developer switch; the judged metric is KSPCG")
    ap.add_argument("--leg-only", default="", help="developer switch (scripts/collect_profiles.sh): run just this leg and print {\"legs\": {...}}, so that a profile holds one matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    a = ap.parse_args()

    if a.leg_only:
        from petsc_amd import capi, petsc as P
        comm = P.Comm(device=0, backend="self")
        L = capi.load()
        tm = Timer(L, comm.ctx)
        print(json.dumps({"legs": {a.leg_only: run_leg(P, L, comm, tm, a.leg_only, a.steps)}}), flush=True)
        tm.close()
        comm.finalize()
        return
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # a plain `python bench.py --gpus N`: this process becomes the launcher -- before it touches HIP, RCCL or torch -- of N fresh
        # ranks, one GPU each (the reference's parallel tests are launched the same way: cmake/TestRunner.cmake:5-18, `${MPIEXEC} -n N exe`)
        raise SystemExit(self_launch(a.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        a.gpus = world
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane (unique id, barrier, max over ranks) on gloo; the data path is RCCL inside libpetschip
        dist.init_process_group("gloo", rank=rank, world_size=world)

    from petsc_amd import capi, petsc as P
    kind, n, perturb = parse_workload(a.workload)
    backend = "self"
    if world > 1:
        # data path: RCCL (one GPU per rank).  PHIP_BENCH_BACKEND=shm stages the ghost exchange through host shared memory instead, so
        # that the ranks may share a GPU: a rehearsal of this script's distributed flow on a one-GPU box, not a measurement.
        backend = os.environ.get("PHIP_BENCH_BACKEND", "rccl")
        ndev = max(torch.cuda.device_count(), 1)

        def make_comm(be):
            """the communicator of all ranks, or (None, reason) on every rank when it cannot be brought up on one of them"""
            why = ""
            try:
                ids = [(P.Comm.unique_id() if be == "rccl" else os.urandom(128)) if rank == 0 else None]
            except Exception as e:  # ncclGetUniqueId failed on rank 0: the others must not wait for it in ncclCommInitRank
                ids, why = [b""], "unique id: %s" % e
            dist.broadcast_object_list(ids, src=0)
            if not ids[0]:
                return None, why or "rank 0 could not make the RCCL unique id"
            cm = None
            try:
                cm = P.Comm(device=local_rank % ndev, backend=be, id128=ids[0], nranks=world, rank=rank)
                one = np.ones(1)
                if L0.phip_comm_allreduce_host(cm.hip_comm(), one.ctypes.data_as(C.POINTER(C.c_double)), 1, 0) or one[0] != world:
                    raise RuntimeError("trial sum over the ranks gave %r: %s" % (one[0], L0.phip_last_error().decode()))
            except Exception as e:
                why = "rank %d: %s" % (rank, e)
            whys = [None] * world
            dist.all_gather_object(whys, why)
            bad = [w for w in whys if w]
            if bad:
                if cm is not None and not why:
                    try:
                        cm.finalize()
                    except Exception:
                        pass
                return None, bad[0]
            return cm, ""

        L0 = capi.load()
        comm, fell_back = make_comm(backend)
        if comm is None and backend == "rccl":
            # RCCL did not come up (reported in multi_gpu.backend_fallback): the ghost exchange and the sums over the ranks staged
            # through host shared memory still give a measurement of the same job, one rank per GPU
            if rank == 0:
                sys.stderr.write("bench: RCCL backend failed (%s); falling back to the host-staged backend\n" % fell_back)
            backend = "shm"
            comm, why2 = make_comm(backend)
            if comm is None:
                raise SystemExit("no communicator: rccl: %s; shm: %s" % (fell_back, why2))
        elif comm is None:
            raise SystemExit("no communicator: %s: %s" % (backend, fell_back))
        else:
            fell_back = ""
    else:
        comm = P.Comm(device=0, backend="self")
    L = capi.load()
    tm = Timer(L, comm.ctx)

    def dsync():
        rc = L.phip_device_sync(comm.ctx)
        if rc:
            raise RuntimeError(L.phip_last_error().decode())

    def barrier():
        dsync()
        if dist is not None:
            dist.barrier()
        dsync()

    t0 = time.time()
    N, m, rs, ai, aj, aa = local_rows(kind, n, world, rank, a.scaling, perturb)
    nz_local = int(ai[-1])
    halo = None
    if world > 1:
        A = P.Mat.mpiaij_with_arrays(comm, m, m, N, N, ai, aj, aa)
        blkA, blkB = A.block(0), A.block(1)
        nzA, nzB, mB = blkA["nz"], blkB["nz"], (len(blkB["cprow_rindex"]) if blkB["cprow_use"] else m)
        del blkA, blkB
        gh = A.ghosts()
        halo = {"ghost_entries_received": int(len(gh["garray"])), "entries_sent": int(len(gh["to_indices"])), "neighbours_to": [int(p) for p in gh["to_procs"]],
                "neighbours_from": [int(p) for p in gh["from_procs"]], "bytes_per_matmult": int(8 * (len(gh["garray"]) + len(gh["to_indices"])))}
        del gh
    else:
        A = P.Mat.seqaij_with_arrays(comm, m, m, ai, aj, aa)
        nzA, nzB, mB = nz_local, 0, 0
    setup_s = time.time() - t0
    keep_host_csr = (world == 1 and rank == 0 and not a.no_cpu_baseline) or world > 1  # N > 1: for the check of the distributed MatMult below
    if not keep_host_csr:
        del ai, aj, aa
    x, b = A.create_vecs()
    x.set(1.0)
    A.mult(x, b)  # b = A * 1
    ksp = P.KSP(comm, a.ksp, "jacobi")
    ksp.set_operators(A)

    def run(its):
        ksp.set_tolerances(rtol=0.0, abstol=1e-300, dtol=1e300, max_it=its)
        return ksp.solve(b, x)

    hA = A.device_mat(0)
    kernel_name, layout_name, real_bytes, spmv_bytes = layout_of(L, hA, m, m, nzA)
    spmv_add_bytes = (12 * nzB + 28 * mB) if nzB else 0
    iter_bytes = spmv_bytes + spmv_add_bytes + 136 * m          # algorithmic: SURVEY.md 8d (the reference's passes)
    iter_real_bytes = real_bytes + spmv_add_bytes + 80 * m      # really moved: the SpMV's layout bytes + the two fused vector launches (40 m each)

    run(max(a.warmup, 1))
    barrier()
    t1 = time.perf_counter()
    res = run(a.steps)
    barrier()
    dt = time.perf_counter() - t1
    # the same solve once more, untimed, with a hipEvent pair around every launch: per-launch durations of the three kernels inside KSPCG
    x.set(1.0)
    L.phip_profile_enable(comm.ctx, 1)
    run(a.steps)
    barrier()
    cap = a.steps + 8
    dur = (C.c_float * cap)()
    cnt = C.c_int32()
    L.phip_profile_durations(comm.ctx, 0, cap, dur, C.byref(cnt))
    assert res["its"] == a.steps, "CG stopped after %d of %d iterations (reason %d)" % (res["its"], a.steps, res["reason"])
    incg = [dur[i] for i in range(min(cnt.value, cap))]
    spmv_ms = float(np.mean(incg)) if incg else 0.0
    incg_vec = {}
    for kind_ in (2, 3):  # the reducing batch / the pending elementwise operations launched by the next MatMult
        L.phip_profile_durations(comm.ctx, kind_, cap, dur, C.byref(cnt))
        incg_vec[kind_] = [dur[i] for i in range(min(cnt.value, cap))][2:]  # the first launches belong to KSPSolve's set-up (other shapes)
    L.phip_profile_enable(comm.ctx, 0)

    # the diagonal-block SpMV kernel alone (kernel-level handle of the matrix the solve used), then the whole MatMult (ghost exchange
    # + diagonal block + off-diagonal block on N > 1)
    dxr, dyr = x.device_ptr(), b.device_ptr()
    xsave = x.array()
    b2b_us, b2b_pcts = tm.kernel(lambda: L.phip_mat_mult(hA, dxr, dyr))
    if dist is not None:
        dist.barrier()
    y2 = b.duplicate()
    mm_us, mm_pcts = tm.kernel(lambda: A.mult(x, y2))
    offd_us = None
    if world > 1:
        hB = A.device_mat(1)
        lv = capi.Context.__new__(capi.Context)
        lv.lib, lv.h = L, comm.ctx
        dl = lv.to_device(np.zeros(max(halo["ghost_entries_received"], 1)))
        dy2 = y2.device_ptr()
        offd_us, _ = tm.kernel(lambda: L.phip_mat_multadd(hB, dl, dy2, dy2))
        lv.free(dl)
    mm_check = None
    if world > 1:
        # the distributed MatMult (ghost exchange + both blocks) against numpy on this rank's rows, three different vectors in a row
        # (a ghost value left over from the exchange before would show): the one place where this script sees the exchange between GPUs
        errs = []
        for k in range(3):
            xg = np.cos(0.37 * (np.arange(N, dtype=np.float64) % 1013.0) + k)
            x.set_array(xg[rs:rs + m])
            A.mult(x, y2)
            ref = np.add.reduceat(aa * xg[aj], ai[:-1].astype(np.int64))
            ref[np.diff(ai) == 0] = 0.0
            errs.append(float(np.abs(y2.array() - ref).max() / max(np.abs(ref).max(), 1e-300)))
            del xg, ref
        te = torch.tensor(errs, dtype=torch.float64)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        mm_check = {"max_rel_err_vs_numpy": [float(v) for v in te], "tolerance": 1e-12, "ok": bool(float(te.max()) <= 1e-12),
                    "what": "MatMult_MPIAIJ of 3 different vectors in a row against numpy on every rank's rows, max over ranks"}
        del ai, aj, aa
    y2.destroy()
    x.set_array(xsave)
    vk = vec_kernels(L, comm.ctx, tm, m, a.workload if world == 1 else "", incg_vec.get(2), incg_vec.get(3))

    tot = np.array([float(iter_bytes), float(spmv_bytes + spmv_add_bytes), float(dt), float(spmv_ms), float(b2b_us), float(mm_us), float(m), float(nz_local),
                    float(iter_real_bytes), float(real_bytes + spmv_add_bytes)])
    if dist is not None:
        import torch
        tb = torch.tensor([tot[0], tot[1], tot[8], tot[9]], dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        tmx = torch.tensor([tot[2], tot[3], tot[4], tot[5], tot[6], tot[7]], dtype=torch.float64)
        dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
        tmn = torch.tensor([tot[6], tot[7]], dtype=torch.float64)
        dist.all_reduce(tmn, op=dist.ReduceOp.MIN)
        job_iter_bytes, job_spmv_bytes, job_iter_real, job_spmv_real = float(tb[0]), float(tb[1]), float(tb[2]), float(tb[3])
        dt, spmv_ms_max, b2b_us_max, mm_us_max = float(tmx[0]), float(tmx[1]), float(tmx[2]), float(tmx[3])
        rows_minmax, nnz_minmax = [int(tmn[0]), int(tmx[4])], [int(tmn[1]), int(tmx[5])]
    else:
        job_iter_bytes, job_spmv_bytes, spmv_ms_max, b2b_us_max, mm_us_max = float(iter_bytes), float(spmv_bytes), spmv_ms, b2b_us, mm_us
        job_iter_real, job_spmv_real = float(iter_real_bytes), float(real_bytes)
        rows_minmax, nnz_minmax = [m, m], [nz_local, nz_local]

    out = None
    if rank == 0:
        ms_per_step = dt * 1e3 / a.steps
        value = job_iter_real * a.steps / dt / 1e9
        resident = real_bytes + 8 * m * 6
        spmv_roof = roofline_obj(a.workload if world == 1 else "%s/%d" % (a.workload, world), kernel_name, layout_name, real_bytes, spmv_bytes, b2b_us, b2b_pcts,
                                 resident_bytes=resident)
        spmv_roof["avg_launch_us_inside_cg"] = round(spmv_ms * 1e3, 2)
        spmv_roof["launch_us_inside_cg_p10_p50_p90"] = [round(pct(incg, q) * 1e3, 2) for q in (10, 50, 90)] if incg else None
        spmv_roof["inside_cg_note"] = "inside KSPCG the product also leaves sum_i x_i y_i (the VecTDot(P,W) of cg.c:207) and is bracketed by an event pair per launch"
        # which kernel owns the largest share of the timed iteration?
        shares = {"spmv": spmv_ms * 1e3 if incg else b2b_us, "cg_batch": vk["cg_batch"]["avg_launch_us"], "xp_update": vk["xp_update"]["avg_launch_us"]}
        dom = max(shares, key=shares.get)
        roof = dict(spmv_roof if dom == "spmv" else vk[dom])
        roof["share_of_iteration_us"] = {k: round(v, 2) for k, v in shares.items()}
        roof["why_this_kernel"] = "largest share of the timed KSPCG iteration (kernel times above; the rest of %.1f us per step is two host round trips for the reductions)" % (ms_per_step * 1e3)
        out = {
            "metric": "SpMV effective HBM GB/s (% of peak) + KSPCG iters/sec at 1/2/4/8 GPUs",
            "value": round(value, 1), "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "KSPCG+PCJACOBI step (MatMult + Krylov vector kernels) on the 3-D %s-point %s %d^3%s%s, MAT%sAIJHIP/VEC%sHIP, int32/f64, %s scaling over MPIAIJ row blocks"
                       % ("7" if kind == "lap7" else "27", "Laplacian" if kind == "lap7" else "box stencil", n, " per GPU" if a.scaling == "weak" else " (global)",
                          ", every stored entry with a value of its own" if perturb else "",
                          "SEQ" if world == 1 else "MPI", "SEQ" if world == 1 else "MPI", a.scaling),
                       "name": a.workload, "global_rows": N, "rows_per_gpu_min_max": rows_minmax, "nnz_per_gpu_min_max": nnz_minmax,
                       "partition": "MPIAIJ row blocks (z-slabs)",
                       "value_is": "bytes one KSPCG iteration REALLY moves through HBM (the SpMV in its device layout, x and y once each, + 80 B/row of the two fused vector launches "
                                   "+ the off-diagonal block on N > 1) x K / wall time, summed over the ranks (time = max over ranks): never above N x 8 TB/s.  The algorithmic bytes of "
                                   "SURVEY.md 8d (12 nz + 20 m + 4 + 136 m, what the reference's passes would move) are under algorithmic_bytes_per_iteration"},
            "cg_iters_per_s": round(a.steps / dt, 1),
            "pct_of_hbm_peak": round(value / (HBM_PEAK_GBPS * world) * 100, 1),
            "real_bytes_per_iteration": int(job_iter_real), "algorithmic_bytes_per_iteration": int(job_iter_bytes),
            "algorithmic_over_real_bytes": round(job_iter_bytes / job_iter_real, 3),
            "spmv": {"real_GBps_back_to_back": round(job_spmv_real / (mm_us_max * 1e-6) / 1e9, 1),
                     "pct_of_8TBps_per_gpu": round(job_spmv_real / world / (mm_us_max * 1e-6) / 1e9 / HBM_PEAK_GBPS * 100, 1),
                     "real_GBps_in_cg": round(job_spmv_real / (spmv_ms_max * 1e-3) / 1e9, 1) if spmv_ms_max else None,
                     "real_bytes_all_ranks": int(job_spmv_real),
                     "matmult_us_back_to_back": round(mm_us_max, 2), "matmult_us_p10_p50_p90_rank0": mm_pcts,
                     "diag_block_kernel_us": round(b2b_us_max, 2), "offdiag_block_kernel_us_rank0": round(offd_us, 2) if offd_us is not None else None,
                     "algorithmic_bytes_all_ranks": int(job_spmv_bytes), "launches_timed": REPS,
                     "note": "matmult = the whole MatMult (N > 1: pack + ghost exchange overlapped with the diagonal block, then the off-diagonal block); exchange cost not hidden = matmult - diag - offdiag"},
            "roofline": roof,
            "kernels": {"spmv": spmv_roof, "cg_batch": vk["cg_batch"], "xp_update": vk["xp_update"]},
            "final_rnorm": res["rnorm"], "setup_s": round(setup_s, 2),
        }
        if world > 1:
            hc = comm.hip_comm()
            rr, ss = C.c_int(), C.c_int()
            L.phip_comm_rank_size(hc, C.byref(rr), C.byref(ss))
            ipcn = C.c_ulonglong()
            L.phip_halo_ipc_count(C.byref(ipcn))
            out["multi_gpu"] = {"backend": backend, "backend_fallback": ("rccl failed: " + fell_back) if fell_back else None, "matmult_check": mm_check, "rccl_ranks": ss.value if backend == "rccl" else 0, "comm_ranks": ss.value, "halo_rank0": halo,
                                "ghost_exchange": "peer writes into hipIpc-mapped receive buffers (PHIP_HALO_IPC=1)" if ipcn.value else
                                ("grouped ncclSend/ncclRecv on a second stream" if backend == "rccl" else "staged through host shared memory (rehearsal backend)"),
                                "reductions": "host shared-memory ring" if not os.environ.get("PETSCHIP_DEVICE_ALLREDUCE") else "ncclAllReduce"}
    for o in (ksp, x, b, A):
        o.destroy()

    if rank == 0 and world == 1:
        legs = a.legs
        if legs == "auto":
            legs = "lap7_128,lap7_256_var,box27_256_var,rmat_22"
        out["legs"] = {}
        for leg in [s for s in legs.split(",") if s and s != "none"]:
            out["legs"][leg] = run_leg(P, L, comm, tm, leg, a.steps)
        if keep_host_csr:
            out["cpu_baseline"] = cpu_baseline(ai, aj, aa, m, iter_bytes, spmv_bytes, a.cpu_seconds)
    tm.close()
    comm.finalize()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


def self_launch(n):
    """`python bench.py --gpus N` from a plain shell: start N ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set, rank r on GPU r modulo the visible devices), wait for them under a time limit, forward rank 0's JSON line, return non-zero when
    any rank failed.  The launcher itself never initialises HIP.  PHIP_BENCH_TIMEOUT (seconds, default 900)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    limit = float(os.environ.get("PHIP_BENCH_TIMEOUT", "900"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=(r == 0)))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)  # rank 0's line; never blocks the ranks
    reader.start()
    t0 = time.time()
    rc = 0
    while True:
        codes = [p_.poll() for p_ in procs]
        if all(c is not None for c in codes):
            break
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:  # a rank died: the others would wait for it in a barrier until their own time limits
            sys.stderr.write("bench: rank %d exited with %s, stopping the other ranks\n" % bad[0])
            rc = bad[0][1] if bad[0][1] > 0 else 1
            break
        if time.time() - t0 > limit:
            sys.stderr.write("bench: ranks still running after %.0f s, stopping them\n" % limit)
            rc = 124
            break
        time.sleep(0.2)
    for r, p_ in enumerate(procs):
        if p_.poll() is None:
            p_.kill()  # exactly the processes started above
            p_.wait()
        elif p_.returncode and not rc:
            sys.stderr.write("bench: rank %d exited with %s\n" % (r, p_.returncode))
            rc = p_.returncode if p_.returncode > 0 else 1
    reader.join(10)
    out0 = "".join(c for c in chunks if c)
    lines = [l for l in (out0 or "").splitlines() if l.startswith("{")]
    if lines and not rc:
        print(lines[-1], flush=True)
    elif not rc:
        sys.stderr.write("bench: rank 0 printed no result line\n")
        rc = 1
    return rc


def host_cores():
    """(cpus this job may use, physical cores of the machine, logical cpus of the machine)"""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    phys = None
    try:
        txt = subprocess.run(["lscpu", "-p=core,socket"], capture_output=True, text=True, timeout=10).stdout
        phys = len({l for l in txt.splitlines() if l and not l.startswith("#")})
    except Exception:
        pass
    return avail, phys, os.cpu_count()


def cpu_baseline(ai, aj, aa, m, iter_bytes, spmv_bytes, seconds):
    """The oracle (oracle/petsc_oracle.c, gcc -O3 -march=native -fopenmp) on the host: KSPCG+PCJACOBI iterations of the same system,
    (i) as MATMPIAIJ would run it on P cores -- P row blocks, one thread each (P = this job's CPU share, at most 16), (ii) on 8 row
    blocks (the 8-rank partition of the GPU runs), (iii) on one core (MATSEQAIJ).  SURVEY.md 8d."""
    from oracle.oracle import Oracle
    orc = Oracle(fast=True)
    avail, phys, logical = host_cores()
    Pn = max(1, min(int(os.environ.get("OMP_NUM_THREADS", "16")), avail, 16))

    def leg(size, threads, budget):
        op = orc.operator(ai, aj, aa, size, threads)
        bvec = op.mult(np.ones(m))
        cal = 2 if threads == 1 else 10   # the threaded legs need a run-in (first touch, thread start) for a fair estimate
        op.solve(bvec, "cg", "jacobi", rtol=1e-300, abstol=1e-300, divtol=1e300, max_it=2)
        t = time.perf_counter()
        op.solve(bvec, "cg", "jacobi", rtol=1e-300, abstol=1e-300, divtol=1e300, max_it=cal)
        per = (time.perf_counter() - t) / (cal + 1.0)
        its = int(max(3, min(5000, budget / max(per, 1e-6))))
        t = time.perf_counter()
        r = op.solve(bvec, "cg", "jacobi", rtol=1e-300, abstol=1e-300, divtol=1e300, max_it=its)
        dt = time.perf_counter() - t
        t = time.perf_counter()
        nmult = 3
        for _ in range(nmult):
            op.mult(bvec)
        dts = (time.perf_counter() - t) / nmult
        return {"value": round(iter_bytes * r["its"] / dt / 1e9, 2), "unit": "GB/s", "cores": threads, "kind": "port",
                "cg_iters_per_s": round(r["its"] / dt, 2), "spmv_GBps": round(spmv_bytes / dts / 1e9, 2), "iterations": r["its"], "seconds": round(dt, 1)}

    one = leg(1, 1, seconds / 4)
    out = one
    if Pn > 1:
        out = leg(Pn, Pn, seconds / 2)
        out["one_core"] = one
        if Pn >= 8:
            out["eight_row_blocks"] = leg(8, 8, seconds / 4)
    out["sample"] = ("%d KSPCG+PCJACOBI iterations (%.1f s) of the same system by the oracle's C restatement of MatMult_MPIAIJ / Vec*_MPI on %d row blocks (MPIAIJ's partition), "
                     "one OpenMP thread each (gcc -O3 -march=native -fopenmp); one_core = MatMult_SeqAIJ / Vec*_Seq on a single thread; eight_row_blocks = 8 threads, the partition of the 8-GPU run"
                     % (out["iterations"], out["seconds"], out["cores"]))
    out["host"] = {"cpus_available_to_this_job": avail, "physical_cores_of_the_machine": phys, "logical_cpus_of_the_machine": logical}
    return out


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn the rocprofv3 output of scripts/collect_profiles.sh (gpurun_out/prof_<workload>_{kt,fetch,write}) into the summaries committed
under profiles/: <tag>_bench_<workload>_kernel_stats.csv, <tag>_bench_<workload>_pmc_hbm.json and <tag>_hbm_traffic.json (the
file bench.py reads `roofline.traffic` from, keyed "<workload>:<kernel>").

    python scripts/summarize_profiles.py r02

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes): gfx950 reports half of a wide coalesced read (MI355X_MICROARCH.md,
HBM / rocprofv3 section; confirmed on the axpy kernel in round 1).  The two counters come from separate --pmc passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
# bench.py kernel label -> prefix of the demangled kernel name the counters are taken from (the variant the back-to-back timing launches)
WANT = {
    "box27_256": {"spmv_dmaskg_kernel": "void spmv_dmaskg_kernel<false", "cg_update_kernel": "void cg_update_kernel", "xp_update_kernel": "void xp_update_kernel"},
    "lap7_128": {"spmv_dmask4_kernel": "void spmv_dmask4_kernel<false", "spmv_rowblock_kernel": "void spmv_rowblock_kernel<false"},
    "lap7_256_var": {"spmv_sell_kernel": "void spmv_sell_kernel<true, false", "spmv_rowblock_kernel": "void spmv_rowblock_kernel<false"},
    "box27_256_var": {"spmv_sell_kernel": "void spmv_sell_kernel<true, false", "spmv_rowblock_kernel": "void spmv_rowblock_kernel<false"},
    "rmat_22": {"spmv_rowblock_kernel": "void spmv_rowblock_kernel<false"},
}


def newest(pat):
    c = glob.glob(pat)
    return max(c, key=os.path.getmtime) if c else None  # gpurun merges new files beside older ones


def source_sha16():
    """the figure bench.py compares: sha256 over the kernel sources of this tree (the tree the profiled library was built from)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join("petsc_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def stats_avg_us(path, prefix):
    """average duration (us) and calls of the kernel whose demangled name starts with prefix in a rocprofv3 kernel_stats.csv (the variant with the most total time)"""
    best = None
    for r in csv.DictReader(open(path)):
        if r["Name"].startswith(prefix):
            if best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"]):
                best = r
    return (round(float(best["AverageNs"]) / 1e3, 2), int(best["Calls"]), best["Name"].split("(")[0]) if best else (None, None, None)


traffic = {"_source_sha16": source_sha16(),
           "_note": "HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), mean over the dispatches of the named kernel in "
                    "`python bench.py --no-cpu-baseline --steps 20 --warmup 2 <--workload W --legs none | --leg-only L>` (scripts/collect_profiles.sh) under rocprofv3 --pmc "
                    "(one counter per pass).  FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950."}
for wl, want in WANT.items():
    kt = newest("gpurun_out/prof_%s_kt/*/*_kernel_stats.csv" % wl)
    if kt:
        shutil.copyfile(kt, "profiles/%s_bench_%s_kernel_stats.csv" % (tag, wl))
        for f in glob.glob("gpurun_out/prof_%s_kt.log" % wl):   # the bench line of the profiled run itself
            lines = [l for l in open(f) if l.startswith("{")]
            if lines:
                open("profiles/%s_bench_%s_line.json" % (tag, wl), "w").write(lines[-1])
    out = {}
    for nm in ("FETCH_SIZE", "WRITE_SIZE"):
        f = newest("gpurun_out/prof_%s_%s/*/*_counter_collection.csv" % (wl, nm.split("_")[0].lower()))
        if not f:
            continue
        d = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == nm:
                d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in d.items():
            out.setdefault(k, {})[nm + "_KB_avg"] = round(sum(v) / len(v), 1)
            out[k][nm + "_KB_max"] = round(max(v), 1)
            out[k][nm + "_launches"] = len(v)
    if not out:
        continue
    json.dump(out, open("profiles/%s_bench_%s_pmc_hbm.json" % (tag, wl), "w"), indent=1, sort_keys=True)
    for label, prefix in want.items():
        ks = [k for k in out if k.startswith(prefix) and "FETCH_SIZE_KB_avg" in out[k] and "WRITE_SIZE_KB_avg" in out[k]]
        if not ks:
            print("no counters for", wl, label)
            continue
        # with several variants under one prefix take the one with the most traffic (the main workload's size, not a smaller leg's)
        k = max(ks, key=lambda q: out[q]["FETCH_SIZE_KB_avg"])
        b = int((2 * out[k]["FETCH_SIZE_KB_avg"] + out[k]["WRITE_SIZE_KB_avg"]) * 1024)
        avg_us, calls, kname = stats_avg_us(kt, prefix) if kt else (None, None, None)
        traffic["%s:%s" % (wl, label)] = {"bytes_per_launch": b, "kernel": k, "fetch_KB_reported": out[k]["FETCH_SIZE_KB_avg"], "write_KB": out[k]["WRITE_SIZE_KB_avg"],
                                          "launches": out[k]["FETCH_SIZE_launches"], "source": "profiles/%s_bench_%s_pmc_hbm.json" % (tag, wl),
                                          "kernel_stats_avg_us": avg_us, "kernel_stats_calls": calls, "kernel_stats_row": kname,
                                          "kernel_stats_file": "profiles/%s_bench_%s_kernel_stats.csv" % (tag, wl)}
        print(wl, label, k[:70], b)
json.dump(traffic, open("profiles/%s_hbm_traffic.json" % tag, "w"), indent=1, sort_keys=True)

#!/usr/bin/env python3
"""Turn the rocprofv3 output under gpurun_out/prof_{kt,fetch,write} into the summaries committed under profiles/.

    python scripts/summarize_profiles.py <tag> <workload>      e.g.  r01 lap7_128
"""
import collections
import os
import csv
import glob
import json
import shutil
import statistics
import sys

tag, workload = sys.argv[1], sys.argv[2]
def newest(pat):
    return max(glob.glob(pat), key=os.path.getmtime)  # gpurun merges new files beside older ones


kt = newest("gpurun_out/prof_kt/*/*_kernel_stats.csv")
shutil.copyfile(kt, "profiles/%s_bench_%s_kernel_stats.csv" % (tag, workload))
out = {}
for nm, pat in (("FETCH_SIZE", "gpurun_out/prof_fetch/*/*_counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/prof_write/*/*_counter_collection.csv")):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(pat))):
        d[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in d.items():
        out.setdefault(k, {})[nm + "_KB_avg"] = sum(v) / len(v)
        out[k][nm + "_launches"] = len(v)
json.dump(out, open("profiles/%s_bench_%s_pmc_hbm.json" % (tag, workload), "w"), indent=1, sort_keys=True)
spmv = [k for k in out if "spmv_sell" in k or "spmv_rowblock" in k or "spmv_pair" in k or "spmv_dmask" in k]
spmv.sort(key=lambda k: -out[k].get("FETCH_SIZE_launches", 0))
k = spmv[0]
traffic = int(2 * out[k]["FETCH_SIZE_KB_avg"] * 1024 + out[k]["WRITE_SIZE_KB_avg"] * 1024)
try:
    t = json.load(open("profiles/hbm_traffic.json"))
except Exception:
    t = {}
t[workload] = traffic
t["_note"] = ("HBM bytes per SpMV launch = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half "
              "of a wide coalesced read; confirmed on the axpy kernel: 16396 KB reported for 2 x 16384 KB read). Separate --pmc passes.")
t["_kernel_" + workload] = k
json.dump(t, open("profiles/hbm_traffic.json", "w"), indent=1)
print(k, out[k], traffic)
# in-CG vs back-to-back durations of the SpMV kernel, and one iteration's timeline
rows = list(csv.DictReader(open(newest("gpurun_out/prof_kt/*/*_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
incg, b2b = [], []
for i, r in enumerate(rows):
    if "spmv" in r["Kernel_Name"].split("(")[0]:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        nxt = rows[i + 1]["Kernel_Name"].split("(")[0] if i + 1 < len(rows) else ""
        (b2b if "spmv" in nxt else incg).append(dur)
print("spmv in CG: n=%d mean %.2f us; back to back: n=%d mean %.2f us" % (len(incg), statistics.mean(incg) / 1e3, len(b2b), statistics.mean(b2b) / 1e3))
cnt, start = 0, None
for r in rows:
    if "spmv" in r["Kernel_Name"].split("(")[0]:
        cnt += 1
    if cnt == 150:
        start = start or int(r["Start_Timestamp"])
        print("%8.2f %7.2f %s" % ((int(r["Start_Timestamp"]) - start) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"][:60]))
    if cnt == 151:
        print("%8.2f next spmv" % ((int(r["Start_Timestamp"]) - start) / 1e3))
        break

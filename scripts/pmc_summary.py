#!/usr/bin/env python3
"""Per-kernel means of the counters collected by scripts/pmc_spmv.sh.   python scripts/pmc_summary.py <tag> [kernel-substring]"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "spmv"
out = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmc_%s_*/" % tag)):
    fs = glob.glob(d + "*/*_counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0]
        if sub in k:
            acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        out[k][c] = round(sum(v) / len(v), 1)
        out[k]["_launches"] = len(v)
print(json.dumps(out, indent=1, sort_keys=True))

#!/usr/bin/env python3
"""Average counter value per launch and kernel from gpurun_out/pmc_pair_*/**/*_counter_collection.csv.  python scripts/pmc_summary.py [substring]"""
import collections
import csv
import glob
import json
import sys

want = sys.argv[1] if len(sys.argv) > 1 else "spmv"
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_pair_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if want in k:
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: {c: round(sum(v) / len(v), 1) for c, v in d.items()} | {"_launches": max(len(v) for v in d.values())} for k, d in out.items()}
print(json.dumps(res, indent=1, sort_keys=True))

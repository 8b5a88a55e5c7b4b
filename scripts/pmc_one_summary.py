#!/usr/bin/env python3
"""mean per-launch counter values of the SpMV kernels in gpurun_out/r03/pmc_<tag>_<pass>/ -> one JSON line (stdout) and
gpurun_out/r03/pmc_<tag>.json"""
import collections
import csv
import glob
import json
import sys

root, tag = sys.argv[1], sys.argv[2]
out = collections.defaultdict(dict)
for f in glob.glob("%s/pmc_%s_*/**/*_counter_collection.csv" % (root, tag), recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "spmv" in k or "hot_gather" in k or "route" in k:
            d[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in d.items():
        out[k][c] = round(sum(v) / len(v), 1)
        out[k]["launches"] = len(v)
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        v["hbm_bytes_2xFETCH_plus_WRITE"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
for f in glob.glob("%s/pmc_%s_1.log" % (root, tag)):
    for line in open(f):
        if line.startswith("{"):
            out["_run"] = json.loads(line)
json.dump(out, open("%s/pmc_%s.json" % (root, tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, sort_keys=True))

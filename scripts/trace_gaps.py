#!/usr/bin/env python3
"""Timeline of one KSPCG iteration from a rocprofv3 --kernel-trace CSV: for every kernel of the steady state the mean duration and the
mean idle time before it (its start minus the previous kernel's end).   python scripts/trace_gaps.py <kernel_trace.csv> [period [substring a kernel of the stretch must contain]]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].replace("void ", "") for r in rows]
# steady state: the longest stretch in which the sequence of names repeats with the given period (default: found by search)
period = int(sys.argv[2]) if len(sys.argv) > 2 else 0
best = (0, 0, 0)
must = sys.argv[3] if len(sys.argv) > 3 else ""
for p in ([period] if period else range(2, 9)):
    i = 0
    while i + p < len(names):
        j = i
        while j + p < len(names) and names[j] == names[j + p]:
            j += 1
        if j - i > best[0] and (not must or any(must in nm for nm in names[i:i + p])):
            best = (j - i, i, p)
        i = max(j, i + 1)
n, i0, p = best
print("steady stretch: %d launches from %d, period %d" % (n, i0, p))
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
for k in range(i0 + p, i0 + n):
    key = ((k - i0) % p, names[k][:60])
    dur[key].append((int(rows[k]["End_Timestamp"]) - int(rows[k]["Start_Timestamp"])) / 1e3)
    gap[key].append((int(rows[k]["Start_Timestamp"]) - int(rows[k - 1]["End_Timestamp"])) / 1e3)
tot = 0.0
for key in sorted(dur):
    d, g = sum(dur[key]) / len(dur[key]), sum(gap[key]) / len(gap[key])
    tot += d + g
    print("%d %-60s idle before %6.2f us   runs %7.2f us   (%d)" % (key[0], key[1], g, d, len(dur[key])))
print("period total %.2f us" % tot)

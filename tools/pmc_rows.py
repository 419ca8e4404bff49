#!/usr/bin/env python3
"""Dev tool: every counter of every rocprofv3 --pmc pass under DIR, averaged per launch of the kernels whose name contains
FILTER.  Usage: pmc_rows.py DIR [FILTER]"""
import collections
import csv
import glob
import sys

flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:80]
        if flt and flt not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
for k, v in acc.items():
    print(k)
    for c in sorted(v):
        n = max(len(launches[k][c]), 1)
        print(f"    {c:34s} {v[c] / n:18.0f}   per launch ({n} launches)")

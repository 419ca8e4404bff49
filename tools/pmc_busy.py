#!/usr/bin/env python3
"""Dev tool: MFMA-busy fraction per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
csv directory: busy = SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / (GRBM_GUI_ACTIVE (summed over 8 XCDs) / 8 x 1024), the same
formula tools/pmc_summary.py uses).  Usage: pmc_busy.py DIR [name filter]"""
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70] + " grid " + r.get("Grid_Size", "?")
    if flt and flt not in k:
        continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
for k, v in acc.items():
    if v.get("GRBM_GUI_ACTIVE"):
        print(f"{k:90s} launches {cnt[k]:3d}  MFMA busy {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 1024):5.1f} %")

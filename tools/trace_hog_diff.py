#!/usr/bin/env python3
"""Which kernels get slower while a collective's stand-in holds a CU?  Input: the rocprofv3 --kernel-trace csv of
`tools/rccl_contention.py --rounds 1 --steps 2 --only "R=1 held"`.  Steps are delimited by the CFG-input kernel (one per step);
a step is a "hog step" when occupancy-hog kernels fall inside it.  Prints per kernel name: launches, time in a plain step,
time in a hog step, difference — and the steps' wall times, compute-stream busy times and idle gaps."""
import csv
import sys
from collections import defaultdict

rows = [r for r in csv.DictReader(open(sys.argv[1]))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [r["s"] for r in rows if "cfg_input" in r["Kernel_Name"]]
hogs = [r for r in rows if "occupancy_hog" in r["Kernel_Name"]]
steps = []
for a, b in zip(marks[:-1], marks[1:]):
    ks = [r for r in rows if a <= r["s"] < b and "occupancy_hog" not in r["Kernel_Name"]]
    nh = sum(1 for h in hogs if a <= h["s"] < b)
    steps.append((a, b, ks, nh))
plain = [s for s in steps if s[3] == 0]
hog = [s for s in steps if s[3] >= 40]
print(f"{len(steps)} steps: {len(plain)} plain, {len(hog)} with hogs")


def summarize(st):
    a, b, ks, nh = st
    by = defaultdict(lambda: [0, 0.0])
    for r in ks:
        by[r["Kernel_Name"][:60]][0] += 1
        by[r["Kernel_Name"][:60]][1] += (r["e"] - r["s"]) / 1e3
    busy = sum(r["e"] - r["s"] for r in ks) / 1e6
    return (b - a) / 1e6, busy, by


pw, pb, pby = summarize(plain[-1])       # the last plain step before the hog steps ... (any steady-state one)
hw, hb, hby = summarize(hog[-1])
print(f"plain step: wall {pw:.2f} ms, sum of kernel durations {pb:.2f} ms;  hog step: wall {hw:.2f} ms, sum of kernel durations {hb:.2f} ms")
diff = sorted(((hby[k][1] - pby.get(k, [0, 0.0])[1], k) for k in hby), reverse=True)
print(f"{'kernel':62s} launches   plain us     hog us    diff us")
for d, k in diff[:25]:
    print(f"{k:62s} {hby[k][0]:6d} {pby.get(k, [0, 0.0])[1]:10.1f} {hby[k][1]:10.1f} {d:10.1f}")
# the tiled GEMM by grid size (tiles = workgroups): which launches pay?
def by_grid(st):
    d = defaultdict(lambda: [0, 0.0])
    for r in st[2]:
        if "gemm_kernel<256, 320" in r["Kernel_Name"]:
            tiles = int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) // (int(r["Workgroup_Size_X"]) * int(r.get("Workgroup_Size_Y", 1) or 1))
            k = (r["Kernel_Name"].split("<")[1].split(">")[0].replace(" ", ""), tiles)
            d[k][0] += 1
            d[k][1] += (r["e"] - r["s"]) / 1e3
    return d
pg, hg = by_grid(plain[-1]), by_grid(hog[-1])
print(f"\n{'tiled GEMM instantiation':34s} tiles  rounds launches  plain us/launch  hog us/launch   diff")
for k in sorted(hg, key=lambda k: -(hg[k][1] - pg.get(k, [0, 0.0])[1])):
    n = hg[k][0]
    p_ = pg.get(k, [n, 0.0])[1] / max(pg.get(k, [n, 0.0])[0], 1)
    print(f"{k[0]:34s} {k[1]:5d} {k[1] / 256:7.3f} {n:7d} {p_:14.1f} {hg[k][1] / n:14.1f} {100 * (hg[k][1] / n / max(p_, 1e-9) - 1):+6.1f} %")
print(f"sum of positive differences {sum(d for d, _ in diff if d > 0) / 1e3:.2f} ms, of all {sum(d for d, _ in diff) / 1e3:.2f} ms")

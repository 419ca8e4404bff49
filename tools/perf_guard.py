#!/usr/bin/env python3
"""Per-launch time guard for the dominant kernels (an 11 % regression of the 3x3-conv kernel went unnoticed for
half a round).  Compares the rocprofv3 --kernel-trace --stats averages of a fresh run with a committed profile and FAILS
(exit 1) when a guarded kernel's average launch got more than --tol (4 %) slower.  Boxes of the pool differ by 2-3 %, so
a failure on a kernel nobody touched deserves one re-run on another box before it is believed.

    python tools/perf_guard.py gpurun_out/rNN/kernel_stats.csv profiles/r03_kernel_stats.csv [--tol 0.04]"""
import argparse
import csv
import sys

GUARDED = (
    "gemm_kernel<256, 320, 4, 2, 1, false, true, 0>",      # 3x3 conv, levels 0-2 (the dominant kernel)
    "gemm_kernel<256, 320, 4, 2, 2, false, true, 0>",      # temporal conv
    "gemm_kernel<256, 320, 4, 2, 0, false, false, 0>",     # tiled Linear
    "flash_attn_kernel<2, false, true>",                   # spatial self-attention
    "tattn2_kernel<320, 24>",                              # K7, second design
    "ff_fused_kernel<320>",                                # K8
)


def load(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("new")
    ap.add_argument("ref")
    ap.add_argument("--tol", type=float, default=0.04)
    a = ap.parse_args()
    new, ref = load(a.new), load(a.ref)
    bad = 0
    for g in GUARDED:
        kn = [k for k in new if g in k]
        kr = [k for k in ref if g in k]
        if not kn or not kr:
            print(f"  {g:58s} not in {'the new run' if not kn else 'the reference'}: skipped")
            continue
        (cn, tn), (cr, tr) = new[kn[0]], ref[kr[0]]
        rel = tn / tr - 1.0
        # launch counts that are no multiple of each other = a different mix of shapes behind one instantiation: flagged
        note = "" if cn % cr == 0 or cr % cn == 0 else f"  (calls {cn} vs {cr}: shape mix differs)"
        flag = "FAIL" if rel > a.tol else "ok"
        bad += rel > a.tol
        print(f"  {g:58s} {tn / 1e3:9.1f} us vs {tr / 1e3:9.1f} us  {100 * rel:+6.1f} %  {flag}{note}")
    if bad:
        print(f"perf_guard: {bad} guarded kernel(s) more than {100 * a.tol:.0f} % slower than {a.ref}")
        sys.exit(1)
    print("perf_guard: ok")


if __name__ == "__main__":
    main()

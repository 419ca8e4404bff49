#!/usr/bin/env python3
"""Merge rocprofv3 --pmc passes of `bench.py` into a per-kernel table (markdown on stdout).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE                 --output-format csv -d out/p_fetch -- python bench.py ...
    rocprofv3 --kernel-trace --pmc WRITE_SIZE                 --output-format csv -d out/p_write -- python bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out/p_mfma -- python bench.py ...
    python tools/pmc_summary.py out/p_fetch out/p_write out/p_mfma

Units / corrections (MI355X_MICROARCH.md §HBM, §rocprofv3): FETCH_SIZE and WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over
the 8 XCDs.  MFMA utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024).
"""
import collections
import csv
import glob
import sys


def load(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")
    if not f:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum
    dur = collections.defaultdict(float)
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
            n[k] += 1
    return per, dur, n


def in_step(kernel: str) -> bool:
    """Is this one of the build's own kernels (what a denoising step launches)?  Everything torch launches while the
    profiled process synthesises its weights and inputs — `at::native::*` fills / copies / RNG, runtime blit kernels, any
    library GEMM — is outside a step and must not be summed into the step's HBM traffic (it was, +5 %)."""
    return not (kernel.startswith("at::native") or kernel.startswith("at::") or "__amd_rocclr_" in kernel or
                kernel.startswith("rocblas") or kernel.startswith("Cijk_") or kernel.startswith("void at::"))


def main():
    merged = collections.defaultdict(dict)
    durs, counts = {}, {}
    args = sys.argv[1:]
    json_out = None
    forwards = 2                    # UNet forwards in the profiled command (bench.py --steps 1 --warmup 1)
    if "--forwards" in args:
        i = args.index("--forwards")
        forwards = int(args[i + 1])
        del args[i:i + 2]
    if "--json" in args:            # per-kernel HBM bytes per launch, for bench.py's roofline.traffic
        i = args.index("--json")
        json_out = args[i + 1]
        del args[i:i + 2]
    for d in args:
        per, dur, n = load(d)
        for k, c in per.items():
            for name, v in c.items():
                merged[k][name] = v
                merged[k][name + "@s"] = dur[k]
            durs.setdefault(k, dur[k])
            counts.setdefault(k, n[k])
    rows = []
    for k, c in merged.items():
        rd = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024           # gfx950: x2 correction
        wr = c.get("WRITE_SIZE", 0.0) * 1024
        t_rd, t_wr = c.get("FETCH_SIZE@s", 0), c.get("WRITE_SIZE@s", 0)
        bw = (rd / t_rd if t_rd else 0) + (wr / t_wr if t_wr else 0)
        util = None
        if c.get("GRBM_GUI_ACTIVE"):
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
        rows.append((durs[k], k, counts[k], rd, wr, bw, util))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    if json_out:
        import json
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import vdx  # noqa: F401
        from vdx._lib import source_sha
        out = {k: {"launches": n, "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
                   "mfma_busy": util} for t, k, n, rd, wr, bw, util in rows if n}
        # identity of what was profiled: bench.py reports `roofline.traffic` from this file only when the kernel
        # sources it runs hash to the same value (a stale profile gives traffic = null, not a wrong number)
        out["_meta"] = {"source_sha": source_sha(), "forwards": forwards,
                        # the build's own kernels only: what the `forwards` UNet forwards of the profiled command moved
                        "hbm_bytes_all_kernels": sum(rd + wr for t, k, n, rd, wr, bw, util in rows if in_step(k)),
                        "hbm_bytes_outside_the_step": sum(rd + wr for t, k, n, rd, wr, bw, util in rows if not in_step(k)),
                        "kernels": sorted(k for t, k, n, rd, wr, bw, util in rows if n)}
        json.dump(out, open(json_out, "w"), indent=1)
    print("| kernel | launches | time share | HBM read GB (x2-corrected) | HBM write GB | HBM GB/s | MFMA busy |")
    print("|---|---|---|---|---|---|---|")
    for t, k, n, rd, wr, bw, util in rows[:32]:
        u = f"{100 * util:.1f} %" if util is not None else "-"
        print(f"| `{k[:72]}` | {n} | {100 * t / tot:.1f} % | {rd / 1e9:.2f} | {wr / 1e9:.2f} | {bw / 1e9:.0f} | {u} |")


if __name__ == "__main__":
    main()

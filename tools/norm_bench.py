#!/usr/bin/env python3
"""Dev tool: the HBM-bound normalisation kernels at the shapes of the XL step (24 f @ 72x128 latent), one line per
(kernel, shape): time per launch and achieved GB/s against the bytes the pass has to move.

    python tools/norm_bench.py                 # times whole calls with HIP events
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/norm_bench.py && python tools/norm_bench.py --trace DIR
                                               # per-kernel split (stats / finalize / apply) from the trace
"""
import glob
import os
import sys

LEVELS = [("L0", 442368, 320), ("L1", 110592, 640), ("L2", 27648, 1280), ("L3", 6912, 1280)]
REPS = 10


def shapes():
    """(tag, kind, M, C, n_samples) in launch order."""
    out = []
    for tag, M, C in LEVELS:
        out.append((tag, "gn spatial (48 samples)", M, C, 48))
        out.append((tag, "gn temporal (2 samples)", M, C, 2))
        out.append((tag, "layernorm", M, C, 0))
    out.append(("L0", "gn spatial, 2 sources 320+320", 442368, 640, 48))
    return out


def from_trace(d):
    import csv
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gn_partial", "gn_finalize", "gn_apply", "layernorm"))]
    i = 0
    for tag, kind, M, C, ns in shapes():
        per = 3 if ns else 1
        blk = keep[i:i + per * (REPS + 2)]
        i += per * (REPS + 2)
        blk = blk[per * 2:]                                  # drop the two warm-up calls
        for k in range(per):
            ds = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in blk[k::per])
            us = ds[len(ds) // 2] / 1e3
            name = blk[k]["Kernel_Name"].split("(")[0][-34:]
            nbytes = M * C * 2 * (1 if "partial" in name else 0 if "finalize" in name else 2)
            print(f"{tag} {kind:32s} {name:36s} {us:9.1f} us  {nbytes / us / 1e3 if nbytes else 0:8.0f} GB/s")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        return from_trace(sys.argv[2])
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import vdx  # noqa: F401
    from vdx import ops
    dev = torch.device("cuda:0")
    for tag, kind, M, C, ns in shapes():
        two = "2 sources" in kind
        x = torch.randn(M, C // 2 if two else C, device=dev).half()
        x2 = torch.randn(M, C // 2, device=dev).half() if two else None
        g, b = (torch.randn(C, device=dev) * 0.1 + 1).half(), (torch.randn(C, device=dev) * 0.1).half()
        out = torch.empty(M, C, device=dev, dtype=torch.float16)

        def call():
            if ns:
                ops.groupnorm(x, g, b, groups=32, n_samples=ns, rows_per_sample=M // ns, eps=1e-5, silu_act=True, x2=x2, out=out)
            else:
                ops.layernorm(x, g, b, M=M, out=out)
        call(); call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REPS
        nbytes = M * C * 2 * (3 if ns else 2)
        print(f"{tag} {kind:32s} whole call {us:9.1f} us  {nbytes / us / 1e3:8.0f} GB/s of {nbytes / 1e6:.0f} MB", flush=True)


if __name__ == "__main__":
    main()

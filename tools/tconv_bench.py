#!/usr/bin/env python3
"""Dev tool: one link of TemporalConvLayer's chain — GroupNorm5d + SiLU + Conv3d (3,1,1) — fused (K3: statistics pass +
csrc/tconv_fused.hip) against the un-fused kernels (statistics + apply pass + TCONV3 GEMM), XL shapes, one process,
interleaved rounds, median.    python tools/tconv_bench.py [--frames 24,16,12]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", default="24,16,12")
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda:0")
for F in (int(f) for f in args.frames.split(",")):
    for lvl, (S, C) in enumerate([(9216, 320), (2304, 640), (576, 1280), (144, 1280)]):
        B = 2
        M = B * F * S
        x = torch.randn(M, C, device=dev).half()
        g, b_ = (torch.randn(C, device=dev) * 0.2 + 1).half(), (torch.randn(C, device=dev) * 0.3).half()
        w = (torch.randn(C, 3 * C, device=dev) / (3 * C) ** 0.5).half()
        bias = (torch.randn(C, device=dev) * 0.1).half()
        out = torch.empty(M, C, device=dev, dtype=torch.float16)
        nbuf = torch.empty(M, C, device=dev, dtype=torch.float16)

        def fused():
            ops.tconv_gn(x, g, b_, w, bias=bias, residual=x, groups=32, B=B, F=F, S=S, eps=1e-5, out=out)

        def unfused():
            ops.groupnorm(x, g, b_, groups=32, n_samples=B, rows_per_sample=F * S, eps=1e-5, silu_act=True, out=nbuf)
            ops.gemm(nbuf, w, M=M, mode=ops.TCONV3, bias=bias, residual=x, tconv=(F, S), out=out)

        def stats_only():
            ops.groupnorm(x, g, b_, groups=32, n_samples=B, rows_per_sample=F * S, eps=1e-5, silu_act=True, out=nbuf)

        def gemm_only():
            ops.gemm(nbuf, w, M=M, mode=ops.TCONV3, bias=bias, residual=x, tconv=(F, S), out=out)
        fns = {"fused": fused, "un-fused": unfused, "GroupNorm (3 kernels)": stats_only, "TCONV3 GEMM": gemm_only}
        if not ops.tconv_gn_supported(C, C, F):
            del fns["fused"]
        ts = {k: [] for k in fns}
        for r in range(args.rounds + 1):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts[k].append(e0.elapsed_time(e1))
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        fl = 2.0 * M * C * 3 * C
        line = f"F {F:2d} level {lvl} M {M:7d} C {C:4d}: " + "  ".join(f"{k} {v:6.3f} ms" for k, v in med.items())
        if "fused" in med:
            line += f"   fused {fl / med['fused'] / 1e9:6.0f} TFLOP/s  x{med['un-fused'] / med['fused']:.2f}"
        print(line, flush=True)
        del x, out, nbuf

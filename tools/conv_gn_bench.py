#!/usr/bin/env python3
"""Dev tool: conv(SiLU(GroupNorm(cat(x, skip)))) of a ResNet block — fused (K1: statistics pass + csrc/conv_fused.hip)
against the un-fused kernels (statistics + apply pass + CONV3X3 GEMM), XL shapes, one process, interleaved rounds, median.
    python tools/conv_gn_bench.py [--frames 24,16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", default="24,16")
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
dev = torch.device("cuda:0")
SHAPES = [("L0 320->320", 72, 128, 320, 0, 320), ("L0 640+320->320", 72, 128, 640, 320, 320), ("L0 320+320->320", 72, 128, 320, 320, 320),
          ("L1 640->640", 36, 64, 640, 0, 640), ("L2 1280->1280", 18, 32, 1280, 0, 1280)]
for F in (int(f) for f in args.frames.split(",")):
    for name, hh, ww, c1, c2, cout in SHAPES:
        n = 2 * F
        M = n * hh * ww
        C = c1 + c2
        xa = torch.randn(M, c1, device=dev).half()
        xb = torch.randn(M, c2, device=dev).half() if c2 else None
        g, b_ = (torch.randn(C, device=dev) * 0.2 + 1).half(), (torch.randn(C, device=dev) * 0.3).half()
        w = (torch.randn(cout, 9 * C, device=dev) / (9 * C) ** 0.5).half()
        bias = (torch.randn(cout, device=dev) * 0.1).half()
        te = (torch.randn(2, cout, device=dev) * 0.3).half()
        out = torch.empty(M, cout, device=dev, dtype=torch.float16)
        nbuf = torch.empty(M, C, device=dev, dtype=torch.float16)

        def fused():
            ops.conv3x3_gn(xa, g, b_, w, x2=xb, bias=bias, bias2=te, rows_per_bias2=F * hh * ww, groups=32, n_img=n, h=hh, wd=ww, eps=1e-5, out=out)

        def norm():
            ops.groupnorm(xa, g, b_, groups=32, n_samples=n, rows_per_sample=hh * ww, eps=1e-5, silu_act=True, x2=xb, out=nbuf)

        def conv():
            ops.gemm(nbuf, w, M=M, mode=ops.CONV3X3, bias=bias, bias2=te, rows_per_bias2=F * hh * ww, conv=(n, hh, ww, hh, ww, 1, False), out=out)

        def unfused():
            norm()
            conv()
        fns = {"fused": fused, "un-fused": unfused, "GroupNorm (3 kernels)": norm, "CONV3X3 GEMM": conv}
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
        fl = 2.0 * M * cout * 9 * C
        print(f"F {F:2d} {name:18s} M {M:7d}: " + "  ".join(f"{k} {v:6.3f} ms" for k, v in med.items())
              + f"   fused {fl / med['fused'] / 1e9:6.0f} TFLOP/s  x{med['un-fused'] / med['fused']:.2f}", flush=True)
        del xa, xb, out, nbuf, w

#!/usr/bin/env python3
"""Per-shape timing of vdx_gemm_f16 on the GEMM shapes of the Zeroscope-XL step (24 f @ 72x128).
Dev tool (not part of the bench contract): prints ms, TFLOP/s and effective GB/s per shape."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")
F_, B = int(os.environ.get("GEMM_BENCH_FRAMES", "24")), 2      # frames of the chunk (16 / 12: the windows of BASELINE cfg4/5)
LEVELS = [(72, 128, 320), (36, 64, 640), (18, 32, 1280), (9, 16, 1280)]


def rnd(*shape):
    return (torch.randn(*shape, device=dev, dtype=torch.float16) * 0.1)


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    rows = []
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    for lvl, (h, w, C) in enumerate(LEVELS):
        n_img = B * F_
        M = n_img * h * w
        cases = [
            (f"L{lvl} conv3x3 {C}->{C}", dict(mode=ops.CONV3X3, cin=C, N=C, conv=(n_img, h, w, h, w, 1, False))),
            (f"L{lvl} tconv3 {C}", dict(mode=ops.TCONV3, cin=C, N=C, tconv=(F_, h * w))),
            (f"L{lvl} linear {C}->{C} +res", dict(mode=ops.PLAIN, cin=C, N=C, res=True)),
            (f"L{lvl} qk {C}->{2 * C}", dict(mode=ops.PLAIN, cin=C, N=2 * C)),
            (f"L{lvl} qkv {C}->{3 * C}", dict(mode=ops.PLAIN, cin=C, N=3 * C)),
            (f"L{lvl} geglu {C}->{8 * C}", dict(mode=ops.PLAIN, cin=C, N=8 * C, geglu=True)),
            (f"L{lvl} ff2 {4 * C}->{C} +res", dict(mode=ops.PLAIN, cin=4 * C, N=C, res=True)),
        ]
        if lvl == 0:   # transformer_in (inner width 512)
            cases += [
                ("Tin proj_in 320->512", dict(mode=ops.PLAIN, cin=320, N=512)),
                ("Tin qkv 512->1536", dict(mode=ops.PLAIN, cin=512, N=1536)),
                ("Tin out 512->512 +res", dict(mode=ops.PLAIN, cin=512, N=512, res=True)),
                ("Tin geglu 512->4096", dict(mode=ops.PLAIN, cin=512, N=4096, geglu=True)),
                ("Tin ff2 2048->512 +res", dict(mode=ops.PLAIN, cin=2048, N=512, res=True)),
                ("Tin proj_out 512->320 +res", dict(mode=ops.PLAIN, cin=512, N=320, res=True)),
            ]
        # the V projection of spatial self-attention, issued with swapped operands so that V arrives transposed for the
        # flash kernel: "rows" = the C output channels, "weights" = the M activation rows
        cases.append((f"L{lvl} v^T {C}x{M}", dict(mode=ops.PLAIN, cin=C, N=M, rows=C)))
        for name, c in cases:
            if only and only not in name:
                continue
            taps = {ops.PLAIN: 1, ops.CONV3X3: 9, ops.TCONV3: 3}[c["mode"]]
            K = taps * c["cin"]
            M = c.get("rows", n_img * h * w)
            a = rnd(M, c["cin"])
            wgt = rnd(c["N"], K)
            bias = rnd(c["N"])
            res = rnd(M, c["N"]) if c.get("res") else None
            n_out = c["N"] // 2 if c.get("geglu") else c["N"]
            out = torch.empty(M, n_out, device=dev, dtype=torch.float16)
            flops = 2.0 * M * c["N"] * K
            byts = 2.0 * (M * c["cin"] + M * n_out + (M * c["N"] if res is not None else 0) + c["N"] * K)
            per = []
            ws_ok = c["mode"] == ops.PLAIN and K in (320, 512, 640)
            for v in (1, 2, 3, 4, 7, 8, 0):
                if v == 7 and not ws_ok:
                    per.append(float("inf"))
                    continue
                fn = lambda: ops.gemm(a, wgt, M=M, mode=c["mode"], bias=bias, residual=res, out=out, variant=v,
                                      geglu=c.get("geglu", False), conv=c.get("conv"), tconv=c.get("tconv"))
                try:
                    per.append(timeit(fn))
                except Exception:   # variant not applicable to this shape
                    per.append(float("inf"))
            auto = ops.gemm_kernel_name(M, c["N"], K, c["mode"], c.get("geglu", False))
            rows.append((name, M, c["N"], K, per, flops, byts, auto))
            del a, wgt, out, res
    names = ["128x128", "256x320", "ring4", "128x320", "ws", "128x320w8", "auto"]
    print(f"{'shape':30s} {'M':>7s} {'N':>6s} {'K':>6s} | ms: " + " ".join(f"{n:>8s}" for n in names) +
          " | best TF/s  GB/s | auto")
    for r in rows:
        best = min(r[4])
        print(f"{r[0]:30s} {r[1]:7d} {r[2]:6d} {r[3]:6d} |     " + " ".join(f"{m:8.3f}" for m in r[4]) +
              f" | {r[5] / best / 1e9:8.1f} {r[6] / best / 1e6:6.0f} | {names[r[4].index(best)]} auto={r[7][:24]}")


if __name__ == "__main__":
    main()

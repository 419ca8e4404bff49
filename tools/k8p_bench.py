#!/usr/bin/env python3
"""K8 with proj_out as its tail (csrc/ff_fused.hip, PO) against K8 + the proj_out GEMM, level 0 of the XL step, same process,
interleaved rounds, median:  python tools/k8p_bench.py [frames ...]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda", 0)
inner = 320
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, device=dev, generator=g) * k).half()       # noqa: E731
blob = packing.pack_k8(r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1), r(inner, k=0.2) + 1, r(inner, k=0.1))
wp, bp = r(inner, inner, k=0.05), r(inner, k=0.1)
tail = packing.pack_k8_proj(wp, bp)


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for F in [int(a) for a in sys.argv[1:]] or [24, 16, 12]:
    M = 2 * F * 9216
    t, x = r(M, inner, k=1.5), r(M, inner)
    y, out = torch.empty_like(t), torch.empty_like(t)

    def fused():
        ops.ff_block(t, blob, M=M, proj=(tail, x, M), out=out)

    def pair():
        ops.ff_block(t, blob, M=M, out=y)
        ops.gemm(y, wp, M=M, bias=bp, residual=x, out=out)

    def k8_only():
        ops.ff_block(t, blob, M=M, out=y)

    a, b, c = [], [], []
    for _ in range(5):
        a.append(timeit(fused))
        b.append(timeit(pair))
        c.append(timeit(k8_only))
    fa, fb, fc = statistics.median(a), statistics.median(b), statistics.median(c)
    fl = 2.0 * M * 13 * inner * inner
    print(f"F {F:2d}: K8 + proj_out fused {fa:.3f} ms ({fl / fa / 1e9:.0f} TFLOP/s)   K8 {fc:.3f} + GEMM = {fb:.3f} ms   x{fb / fa:.2f}", flush=True)

#!/usr/bin/env python3
"""K5 (csrc/xattn.hip) against the four kernels it replaces, level 0 of the XL step (inner 320, 77 text tokens), same process,
interleaved rounds, median:  python tools/k5_bench.py [frames ...]"""
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda", 0)
inner, heads, cross, pad, kv_len = 320, 5, 1024, 128, 77
g = torch.Generator().manual_seed(0)
r = lambda *s, k=1.0: (torch.randn(*s, generator=g) * k).half().to(dev)       # noqa: E731
gamma, beta, bo = r(inner, k=0.2) + 1, r(inner, k=0.1), r(inner, k=0.1)
wq, wo, wk, wv = r(inner, inner, k=0.09), r(inner, inner, k=0.05), r(inner, cross, k=0.05), r(inner, cross, k=0.04)
blob = packing.pack_k5(wq, wo, gamma, beta, bo, 0.125).to(dev)


def timeit(fn, n=20):
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
    for n_items in (2, 1):
        rows = F * 9216
        M = n_items * rows
        t = r(M, inner, k=1.5)
        ehs_pad = torch.zeros(n_items * pad, cross, dtype=torch.float16, device=dev)
        ehs_pad.view(n_items, pad, cross)[:, :kv_len] = r(n_items, kv_len, cross)
        k_rows = ops.gemm(ehs_pad, wk, M=n_items * pad)
        vt = ops.gemm(wv, ehs_pad, M=inner)
        kvb = packing.pack_k5_kv(k_rows, vt, n_items, pad)
        out = torch.empty_like(t)

        def fused():
            ops.cross_attn_block(t, blob, kvb, kv_len=kv_len, n_items=n_items, rows_per_item=rows, out=out)

        def chain():
            ln = ops.layernorm(t, gamma, beta, M=M)
            q = ops.gemm(ln, wq, M=M)
            o = ops.flash_attn(q, k_rows, vt, n_seq=n_items, sq=rows, skv=kv_len, skv_pad=pad, heads=heads, seq_per_kv=1, scale=0.125)
            ops.gemm(o, wo, M=M, bias=bo, residual=t, out=out)

        a, b = [], []
        for _ in range(5):
            a.append(timeit(fused))
            b.append(timeit(chain))
        fa, fb = statistics.median(a), statistics.median(b)
        flops = 2.0 * M * (2 * inner * inner + 2 * kv_len * inner)
        print(f"F {F:2d} items {n_items}: K5 {fa:.3f} ms ({flops / fa / 1e9:.0f} TFLOP/s algorithmic, {3 * M * inner * 2 / fa / 1e6:.0f} GB/s of rows)   "
              f"un-fused chain {fb:.3f} ms   x{fb / fa:.2f}", flush=True)

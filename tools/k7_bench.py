#!/usr/bin/env python3
"""Dev tool: the fused temporal-attention sub-block (K7, csrc/tattn_fused.hip) against the four kernels it replaces,
on the XL step shapes (B = 2, 72x128 latent at level 0 / transformer_in), interleaved rounds in one process."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).half()


def bench(fns, rounds=7):
    ts = {k: [] for k in fns}
    for k, f in fns.items():
        f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for k, f in fns.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            f()
            e1.record()
            torch.cuda.synchronize()
            ts[k].append(e0.elapsed_time(e1))
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


for inner, name in ((320, "level 0"), (512, "transformer_in")):
    for F in (24, 16, 12):
        B, HW = 2, 72 * 128
        M = B * F * HW
        heads = inner // 64
        t = rnd(M, inner)
        gamma, beta, bo = rnd(inner, scale=0.2) + 1, rnd(inner, scale=0.1), rnd(inner, scale=0.1)
        wq, wk, wv, wo = (rnd(inner, inner, scale=0.06) for _ in range(4))
        wqkv = torch.cat([wq, wk, wv], 0).contiguous()
        pq, po = packing.pack_k7_qkv(wq, wk, wv).contiguous(), packing.pack_k7_out(wo).contiguous()
        out = torch.empty_like(t)

        def fused():
            ops.temporal_attn_block(t, gamma, beta, pq, po, bo, B=B, F=F, HW=HW, scale=0.125, out=out)

        blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).contiguous() if inner in packing.K7B_WIDTHS else None

        def fused2():
            ops.temporal_attn_block2(t, blob, B=B, F=F, HW=HW, out=out)

        def unfused():
            ln = ops.layernorm(t, gamma, beta, M=M)
            qkv = ops.gemm(ln, wqkv, M=M)
            o = ops.temporal_attn(qkv, B=B, F=F, HW=HW, heads=heads, scale=0.125)
            ops.gemm(o, wo, M=M, bias=bo, residual=t, out=out)

        fns = {"fused": fused, "unfused": unfused}
        if blob is not None:
            fns["fused2"] = fused2
        r = bench(fns)
        fl = 2.0 * M * 4 * inner * inner + 4.0 * M * F * inner
        print(f"{name:15s} inner {inner} F {F:2d} M {M:7d}: fused {r['fused']:.3f} ms ({fl / r['fused'] / 1e9:6.0f} TFLOP/s, "
              f"{2 * M * inner * 2 / r['fused'] / 1e6:5.0f} GB/s algorithmic)   un-fused chain {r['unfused']:.3f} ms   x{r['unfused'] / r['fused']:.2f}"
              + (f"   second design {r['fused2']:.3f} ms ({fl / r['fused2'] / 1e9:6.0f} TFLOP/s)" if blob is not None else ""),
              flush=True)
        del t, out

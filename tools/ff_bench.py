#!/usr/bin/env python3
"""Dev tool: the fused feed-forward sub-block (K8, csrc/ff_fused.hip) against the three kernels it replaces, on the XL
step's level-0 shapes (B = 2, 72x128 latent, inner 320), interleaved rounds in one process."""
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
    for f in fns.values():
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


inner = 320
for F in (24, 16, 12):
    M = 2 * F * 72 * 128
    t = rnd(M, inner)
    gamma, beta = rnd(inner, scale=0.2) + 1, rnd(inner, scale=0.1)
    w1, b1 = rnd(8 * inner, inner, scale=0.06), rnd(8 * inner, scale=0.1)
    w2, b2 = rnd(inner, 4 * inner, scale=0.03), rnd(inner, scale=0.1)
    blob = packing.pack_k8(w1, b1, w2, b2, gamma, beta)
    wp, bp = packing.pack_geglu(w1, b1)
    w2p = packing.pack_conv1x1(w2)
    out = torch.empty_like(t)

    def fused():
        ops.ff_block(t, blob, M=M, out=out)

    def unfused():
        ln = ops.layernorm(t, gamma, beta, M=M)
        gg = ops.gemm(ln, wp, M=M, bias=bp, geglu=True)
        ops.gemm(gg, w2p, M=M, bias=b2, residual=t, out=out)

    r = bench({"fused": fused, "unfused": unfused})
    fl = 2.0 * M * inner * 12 * inner
    print(f"level 0 inner {inner} F {F:2d} M {M:7d}: fused {r['fused']:.3f} ms ({fl / r['fused'] / 1e9:6.0f} TFLOP/s)   "
          f"un-fused chain {r['unfused']:.3f} ms   x{r['unfused'] / r['fused']:.2f}", flush=True)
    del t, out

#!/usr/bin/env python3
"""Dev tool: one small launch of the second fused temporal-attention kernel (csrc/tattn2.hip) against fp32, with the
error broken down by row tile / column group so that an indexing fault shows its shape.  Library from VDX_LIB_PATH."""
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

import torch
import torch.nn.functional as Fn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
inner, heads = 320, 5
for (B, Fr, HW) in ((1, 24, 8), (2, 24, 20), (1, 16, 12 * 30)):
    g = torch.Generator().manual_seed(1)
    M = B * Fr * HW
    h = lambda x: x.half().float()    # noqa: E731
    t = h(torch.randn(M, inner, generator=g) * 1.5 + 0.3)
    gamma, beta = h(1 + 0.2 * torch.randn(inner, generator=g)), h(0.1 * torch.randn(inner, generator=g))
    wq, wk, wv, wo = (h(torch.randn(inner, inner, generator=g) * s_) for s_ in (0.09, 0.09, 0.06, 0.05))
    bo = h(0.1 * torch.randn(inner, generator=g))
    ln = Fn.layer_norm(t, (inner,), gamma, beta, 1e-5)
    seq = lambda x: x.reshape(B, Fr, HW, heads, 64).permute(0, 2, 3, 1, 4)   # noqa: E731  [B][HW][heads][F][64]
    a = torch.softmax(seq(ln @ wq.t()) @ seq(ln @ wk.t()).transpose(-1, -2) * 0.125, -1) @ seq(ln @ wv.t())
    o = a.permute(0, 3, 1, 2, 4).reshape(M, inner)
    ref = t + o @ wo.t() + bo
    blob = packing.pack_k7b(wq, wk, wv, wo, gamma, beta, bo, 0.125).to(dev)
    out = ops.temporal_attn_block2(t.half().to(dev), blob, B=B, F=Fr, HW=HW).float().cpu()
    err = (out - ref).abs()
    print(f"B {B} F {Fr} HW {HW}: max err {err.max():.4g}  nan {int(torch.isnan(out).sum())}  bad(>0.05) {int((err > 0.05).sum())} / {err.numel()}")
    e = err.reshape(B, Fr, HW, inner)
    print("  by column group of 64:", [f"{e[..., c:c + 64].max():.3g}" for c in range(0, inner, 64)])
    print("  by frame:", [f"{e[:, f].max():.3g}" for f in range(Fr)])
    print("  by pixel (first 16):", [f"{e[:, :, p_].max():.3g}" for p_ in range(min(HW, 16))])
    print("  by batch:", [f"{e[b_].max():.3g}" for b_ in range(B)])

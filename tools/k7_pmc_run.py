#!/usr/bin/env python3
"""Dev tool: a few launches of the fused temporal-attention kernels (K7) and of the fused feed-forward kernel (K8) at
level-0 size, for rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
for inner in (320, 512):
    B, F, HW = 2, 24, 72 * 128
    M = B * F * HW
    t = torch.randn(M, inner, device=dev).half()
    v = lambda s=0.1: (torch.randn(inner, device=dev) * s).half()   # noqa: E731
    w = [(torch.randn(inner, inner, device=dev) * 0.06).half() for _ in range(4)]
    pq, po = packing.pack_k7_qkv(*w[:3]).contiguous(), packing.pack_k7_out(w[3]).contiguous()
    out = torch.empty_like(t)
    g, b_, bo = v() + 1, v(), v()
    for _ in range(4):
        ops.temporal_attn_block(t, g, b_, pq, po, bo, B=B, F=F, HW=HW, scale=0.125, out=out)
    torch.cuda.synchronize()
    if inner in packing.K7B_WIDTHS:        # the second design (csrc/tattn2.hip), F = 24 and 16
        blob = packing.pack_k7b(*w, g, b_, bo, 0.125).contiguous()
        for F2 in (24, 16):
            M2 = B * F2 * HW
            for _ in range(4):
                ops.temporal_attn_block2(t[:M2], blob, B=B, F=F2, HW=HW, out=out[:M2])
        torch.cuda.synchronize()

# K8 (csrc/ff_fused.hip), F = 24 and 16
inner = 320
r = lambda *sh, k=1.0: (torch.randn(*sh, device=dev) * k).half()   # noqa: E731
blob = packing.pack_k8(r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1), r(inner, k=0.2) + 1, r(inner, k=0.1))
for F2 in (24, 16):
    M2 = 2 * F2 * 72 * 128
    t = r(M2, inner)
    out = torch.empty_like(t)
    for _ in range(4):
        ops.ff_block(t, blob, M=M2, out=out)
    torch.cuda.synchronize()
    del t, out

#!/usr/bin/env python3
"""Dev tool: wall time of the fused temporal-attention kernel alone (library from VDX_LIB_PATH), level-0 and
transformer_in shapes, median of 9."""
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
res = []
for inner in (320, 512):
    B, F, HW = 2, 24, 72 * 128
    M = B * F * HW
    t = torch.randn(M, inner, device=dev).half()
    v = lambda s=0.1: (torch.randn(inner, device=dev) * s).half()   # noqa: E731
    w = [(torch.randn(inner, inner, device=dev) * 0.06).half() for _ in range(4)]
    pq, po = packing.pack_k7_qkv(*w[:3]).contiguous(), packing.pack_k7_out(w[3]).contiguous()
    out = torch.empty_like(t)
    g, b_, bo = v() + 1, v(), v()
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.temporal_attn_block(t, g, b_, pq, po, bo, B=B, F=F, HW=HW, scale=0.125, out=out)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
    res.append(f"inner {inner}: {sorted(ts)[len(ts) // 2]:.3f} ms")
print(os.path.basename(os.environ.get("VDX_LIB_PATH", "libvdx_hip.so")), " | ".join(res))

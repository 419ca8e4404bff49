#!/usr/bin/env python3
"""Dev tool: wall time of the fused feed-forward kernel alone (library from VDX_LIB_PATH: the product or a -DK8_ABL_*
timing-only build made by tools/k8_abl.sh), level-0 shape, F = 24 and 16, median of 9."""
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
res = []
inner = 320
r = lambda *s, k=1.0: (torch.randn(*s, device=dev) * k).half()   # noqa: E731
blob = packing.pack_k8(r(8 * inner, inner, k=0.06), r(8 * inner, k=0.1), r(inner, 4 * inner, k=0.03), r(inner, k=0.1), r(inner, k=0.2) + 1, r(inner, k=0.1))
for F in (24, 16):
    M = 2 * F * 72 * 128
    t = r(M, inner)
    out = torch.empty_like(t)
    ts = []
    for i in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.ff_block(t, blob, M=M, out=out)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            ts.append(e0.elapsed_time(e1))
    res.append(f"F {F}: {sorted(ts)[len(ts) // 2]:.3f} ms")
print(os.path.basename(os.environ.get("VDX_LIB_PATH", "libvdx_hip.so")), " | ".join(res))

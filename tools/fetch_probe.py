#!/usr/bin/env python3
"""Dev tool (run under `rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv`): a few launches of single GEMM
shapes, so that per-dispatch L2-fill bytes can be compared with the operand bytes.  Prints the dispatch order."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")
n_img, F_ = 48, 24
cases = []
for lvl, (h, w, C) in enumerate([(72, 128, 320), (36, 64, 640), (18, 32, 1280)]):
    M = n_img * h * w
    cases.append((f"L{lvl} conv3x3 {C}", dict(M=M, mode=ops.CONV3X3, cin=C, N=C, K=9 * C, conv=(n_img, h, w, h, w, 1, False))))
    cases.append((f"L{lvl} tconv3 {C}", dict(M=M, mode=ops.TCONV3, cin=C, N=C, K=3 * C, tconv=(F_, h * w))))
for name, c in cases:
    a = torch.randn(c["M"], c["cin"], device=dev, dtype=torch.float16) * 0.1
    wgt = torch.randn(c["N"], c["K"], device=dev, dtype=torch.float16) * 0.1
    out = torch.empty(c["M"], c["N"], device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.gemm(a, wgt, M=c["M"], mode=c["mode"], out=out, conv=c.get("conv"), tconv=c.get("tconv"))
    torch.cuda.synchronize()
    opb = 2.0 * (c["M"] * c["cin"] + c["N"] * c["K"])
    print(f"{name}: operand read bytes {opb / 1e9:.3f} GB, out {2.0 * c['M'] * c['N'] / 1e9:.3f} GB")

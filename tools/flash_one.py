#!/usr/bin/env python3
"""Dev tool: run the L0 spatial self-attention shape a few times, V as rows of one q|k|v matrix (what the UNet runs), for
rocprofv3 --pmc passes (tools/flash_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")
hw, C, n_seq = 9216, 320, 48
M = n_seq * hw
qkv = torch.randn(M, 3 * C, device=dev, dtype=torch.float16)
out = torch.empty(M, C, device=dev, dtype=torch.float16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw, heads=5, seq_per_kv=1,
                   scale=0.125, out=out, v_rows=True)
torch.cuda.synchronize()

#!/usr/bin/env python3
"""Dev tool: cycles of one range of steps of csrc/tattn2.hip from a -DK7B_STAMPS -DK7B_SS=a -DK7B_SE=b build (VDX_LIB_PATH)."""
import ctypes as C
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import _lib, ops, packing  # noqa: E402

dev = torch.device("cuda:0")
inner, F, B, HW = 320, 24, 2, 72 * 128
M = B * F * HW
t = torch.randn(M, inner, device=dev).half()
v = lambda s=0.1: (torch.randn(inner, device=dev) * s).half()   # noqa: E731
w = [(torch.randn(inner, inner, device=dev) * 0.06).half() for _ in range(4)]
blob = packing.pack_k7b(*w, v() + 1, v(), v(), 0.125).contiguous()
out = torch.empty_like(t)
for _ in range(3):
    ops.temporal_attn_block2(t, blob, B=B, F=F, HW=HW, out=out)
torch.cuda.synchronize()
buf = np.zeros(1024 * 16, np.uint64)
lib = _lib.load()
lib.vdx_debug_read_k7b_stamps.argtypes = [C.c_void_p]
assert lib.vdx_debug_read_k7b_stamps(buf.ctypes.data) == 0
s = buf.reshape(1024, 16)[:200].astype(np.float64)
med = np.median(s, axis=0) / 9.0
print(f"{os.path.basename(os.environ.get('VDX_LIB_PATH', ''))}: per tile: first half {med[0]:.0f} | wait+barrier {med[1]:.0f} | second half {med[2]:.0f} | end hooks {med[3]:.0f} | sum {med[:4].sum():.0f} (whole tile {med[12]:.0f})")

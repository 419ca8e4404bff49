#!/usr/bin/env python3
"""Dev tool: phase shares of the second fused temporal-attention kernel from its stamped diagnostic build
(hipcc -DK7B_STAMPS csrc/tattn2.hip -> libvdx_hip_k7bstamps.so; shares only, the stamps cost time)."""
import ctypes as C
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

import numpy as np
import torch

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("VDX_LIB_PATH", os.path.join(here, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd", "libvdx_hip_k7bstamps.so"))
sys.path.insert(0, here)
import vdx  # noqa: E402,F401
from vdx import _lib, ops, packing  # noqa: E402

dev = torch.device("cuda:0")
inner = 320
for F in (24, 16):
    B, HW = 2, 72 * 128
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
    tiles = M // 192 / 256
    med = np.median(s, axis=0) / tiles
    names = ["first half (reads, 24/12/24 MFMA, riders)", "wait + barrier", "second half (issue, reads, MFMA)", "end hooks"]
    print(f"F {F}: cycles per tile (median over 200 blocks, {tiles:.1f} tiles per block): total {med[12]:.0f}")
    for k, kn in enumerate(("q|k steps (25)", "v steps (25)", "out steps (15)")):
        print(f"  {kn:16s}: " + " | ".join(f"{names[j]} {med[4 * k + j]:.0f}" for j in range(4)) + f" | sum {med[4 * k:4 * k + 4].sum():.0f}")

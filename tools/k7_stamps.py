#!/usr/bin/env python3
"""Dev tool: per-phase cycle totals of the fused temporal-attention kernel (diagnostic library `make stamps`, wave 0
of each block): P0 load+LayerNorm, P1 wait/barrier/DMA-issue, P1 LDS reads + MFMA, attention, P2, P3 wait, P3 MFMA,
P3 epilogue.  Shares only — the stamps forbid overlaps the product kernel has."""
import ctypes
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
TAG = sys.argv[1] if len(sys.argv) > 1 else "0"
HW_ARG = int(sys.argv[2]) if len(sys.argv) > 2 else 72 * 128
os.environ["VDX_LIB_PATH"] = os.path.join(PKG, f"libvdx_hip_stamps{TAG}.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vdx  # noqa: E402,F401
from vdx import ops, packing  # noqa: E402

dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ["VDX_LIB_PATH"])
raw.vdx_debug_read_k7_stamps.argtypes = [ctypes.c_void_p]
NAMES = ["P0 load+LN", "P1 acquire", "P1 reads+MFMA", "attention", "P2", "P3 acquire", "P3 reads+MFMA", "P3 epilogue"]
for inner in (320, 512):
    B, F, HW = 2, 24, HW_ARG
    M = B * F * HW
    t = (torch.randn(M, inner, device=dev)).half()
    v = lambda s=0.1: (torch.randn(inner, device=dev) * s).half()   # noqa: E731
    w = [(torch.randn(inner, inner, device=dev) * 0.06).half() for _ in range(4)]
    pq, po = packing.pack_k7_qkv(*w[:3]).contiguous(), packing.pack_k7_out(w[3]).contiguous()
    out = torch.empty_like(t)
    g, b_, bo = v() + 1, v(), v()
    fn = lambda: ops.temporal_attn_block(t, g, b_, pq, po, bo, B=B, F=F, HW=HW, scale=0.125, out=out)   # noqa: E731
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    assert raw.vdx_debug_read_k7_stamps(buf.ctypes.data) == 0
    act = buf.sum(1) > 0
    iss = np.median((buf[act, 0] >> np.uint64(32)).astype(np.float64))
    buf[:, 0] &= np.uint64(0xffffffff)
    med = np.median(buf[act].astype(np.float64), axis=0)
    print(f"   (of the acquire phases: barrier wait {med[4]:.0f}, DMA issue {iss:.0f} cycles; 'P2' row below shows the barrier wait)")
    print(f"[lib {TAG}, HW {HW}] inner {inner}: {e0.elapsed_time(e1):.3f} ms (stamped build), {int(act.sum())} blocks sampled; median cycles per block {med.sum():.0f}")
    for n, c in zip(NAMES, med):
        print(f"   {n:16s} {c:9.0f}  {100 * c / med.sum():5.1f} %")

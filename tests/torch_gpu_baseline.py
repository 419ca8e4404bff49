#!/usr/bin/env python3
"""Measurement aid (not a test, not collected): what the reference's arithmetic costs on this GPU through stock
PyTorch-ROCm — the oracle's torch restatement of diffusers' UNet3DConditionModel (`oracle/unet3d_ref.py`) run in fp16
on the device with torch's own kernels (hipBLASLt / MIOpen / SDPA), one CFG forward at Zeroscope-XL size.  This is what
`fsdp_chunked_coherent.py:140` executes per step with diffusers on an MI355X, give or take diffusers' own overheads.

    python tests/torch_gpu_baseline.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig  # noqa: E402

F_ = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda:0")
with torch.device("meta"):
    m = UNet3DConditionModelRef(UNet3DConfig.zeroscope())
m = m.to_empty(device=dev).half().eval()
with torch.no_grad():
    for p in m.parameters():
        p.normal_(0.0, 0.02)
    x = torch.randn(2, 4, F_, 72, 128, device=dev, dtype=torch.float16)
    e = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
    t = torch.tensor(500, device=dev)
    for i in range(2):
        t0 = time.time()
        y = m(x, t, e)
        torch.cuda.synchronize()
        print(f"warm-up forward {i}: {time.time() - t0:.2f} s", flush=True)
    torch.cuda.reset_peak_memory_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 3
    for _ in range(n):
        y = m(x, t, e)
    e1.record()
    torch.cuda.synchronize()
    out = y.sample if hasattr(y, "sample") else y
    print(f"stock PyTorch-ROCm fp16, {F_} frames @72x128 latent: {e0.elapsed_time(e1) / n:.1f} ms per CFG forward, "
          f"finite {bool(torch.isfinite(out.float()).all())}, peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

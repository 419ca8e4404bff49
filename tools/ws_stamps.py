#!/usr/bin/env python3
"""Dev tool: per-chunk cycle split of the weights-stationary GEMM (diagnostic library, wave 0 of each block):
barrier wait, DMA issue, and everything between two chunk barriers (LDS reads + MFMA + epilogue + vmcnt wait)."""
import ctypes
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
os.environ["VDX_LIB_PATH"] = os.path.join(PKG, "libvdx_hip_stamps0.so")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ["VDX_LIB_PATH"])
raw.vdx_debug_read_ws_stamps.argtypes = [ctypes.c_void_p]


def run(name, M, N, K, geglu=False, res=False):
    a = torch.randn(M, K, device=dev, dtype=torch.float16) * 0.1
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.1
    b = torch.randn(N, device=dev, dtype=torch.float16) * 0.1
    r = torch.randn(M, N, device=dev, dtype=torch.float16) if res else None
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, M=M, bias=b, residual=r, out=out, geglu=geglu, variant=7)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = np.zeros((256, 4), dtype=np.uint64)
    assert raw.vdx_debug_read_ws_stamps(buf.ctypes.data) == 0
    act = buf[:, 3] > 0
    nch = buf[act, 3].astype(np.float64)
    bar, iss, rest = (buf[act, i].astype(np.float64) / nch for i in range(3))
    print(f"{name:26s} {e0.elapsed_time(e1):7.3f} ms  blocks {int(act.sum())} chunks/block {nch.mean():6.1f} | cycles per chunk: "
          f"barrier {np.median(bar):7.0f}  dma-issue {np.median(iss):6.0f}  compute+epilogue+vmcnt {np.median(rest):7.0f}")


M = 48 * 72 * 128
run("L0 geglu 320->2560", M, 2560, 320, geglu=True)
run("L0 qkv 320->960", M, 960, 320)
run("L0 linear+res 320", M, 320, 320, res=True)
run("Tin qkv 512->1536", M, 1536, 512)
run("Tin geglu 512->4096", M, 4096, 512, geglu=True)
run("L1 geglu 640->5120", M // 4, 5120, 640, geglu=True)
run("L1 linear+res 640", M // 4, 640, 640, res=True)
run("L1 linear 640", M // 4, 640, 640)
run("L1 qkv 640->1920", M // 4, 1920, 640)

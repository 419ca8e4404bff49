#!/usr/bin/env python3
"""Dev tool: where one GEMM tile's time goes.  Runs the diagnostic library (`make -C <pkg>/csrc stamps`,
-DVDX_STAMPS: s_memtime at block start / after the prologue barrier / after the K loop / after the
epilogue stores drained, s_memrealtime at both ends, HW_ID) and prints per-phase medians and the
idle gap between consecutive blocks on the same CU."""
import ctypes
import os
os.environ.setdefault("VDX_ALLOW_LAB_BUILD", "1")      # lab tool: may load a stamps / ablation build
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd")
ABL = os.environ.get("ABL", "0")   # 0 = full kernel, 1 = DMA only, 2 = MFMA + LDS reads only
os.environ["VDX_LIB_PATH"] = os.path.join(PKG, f"libvdx_hip_stamps{ABL}.so")
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")
raw = ctypes.CDLL(os.environ["VDX_LIB_PATH"])
raw.vdx_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float16) * 0.1


def run(name, M, N, K, mode=ops.PLAIN, res=False, geglu=False, conv=None, tconv=None, variant=None, cin=None):
    variant = variant or int(os.environ.get("VARIANT", "2"))
    cin = cin or K
    a, w, bias = rnd(M, cin), rnd(N, K), rnd(N)
    r = rnd(M, N) if res else None
    out = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, M=M, mode=mode, bias=bias, residual=r, out=out, variant=variant, geglu=geglu,
                          conv=conv, tconv=tconv)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    nb = min(32768, ((M + 255) // 256) * ((N + 319) // 320))
    buf = np.zeros((nb, 8), dtype=np.uint64)
    assert raw.vdx_debug_read_stamps(buf.ctypes.data, nb) == 0
    t = buf[:, :4].astype(np.int64)
    r0, r3 = buf[:, 4].astype(np.int64), buf[:, 5].astype(np.int64)
    hw = buf[:, 6]
    cu_key = ((hw >> np.uint64(32)) & np.uint64(15)) * np.uint64(65536) + (hw & np.uint64(0xFF00))  # xcc, se/sh/cu
    pro, main, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    clk = np.median((t[:, 3] - t[:, 0]) / np.maximum(r3 - r0, 1)) * 100.0  # MHz
    gaps, per_cu = [], []
    for k in np.unique(cu_key):
        idx = np.where(cu_key == k)[0]
        o = idx[np.argsort(r0[idx])]
        per_cu.append(len(o))
        gaps += list((r0[o][1:] - r3[o][:-1]) * 10.0)  # ns
    span = (r3.max() - r0.min()) / 100.0  # us
    us = lambda c: np.median(c) / clk
    print(f"{name:28s} {ms:7.3f} ms  span {span:7.1f} us  clk {clk:5.0f} MHz  CUs {len(per_cu)} tiles/CU {np.mean(per_cu):.1f} | "
          f"prologue {us(pro):6.2f}  K-loop {us(main):6.2f}  epilogue {us(epi):6.2f}  gap {np.median(gaps) / 1e3 if gaps else 0:6.2f} us "
          f"| tile total {us(t[:, 3] - t[:, 0]):6.2f} us (p90 {np.percentile(t[:, 3] - t[:, 0], 90) / clk:6.2f})")


def main():
    n_img, h, w, C = 48, 72, 128, 320
    M = n_img * h * w
    run("L0 linear 320->320 +res", M, 320, 320, res=True)
    run("L0 linear 320->320", M, 320, 320)
    run("L0 qkv 320->960", M, 960, 320)
    run("L0 geglu 320->2560", M, 2560, 320, geglu=True)
    run("L0 plain 320->2560", M, 2560, 320)
    run("L0 ff2 1280->320 +res", M, 320, 1280, res=True)
    run("L0 conv3x3 320->320", M, 320, 2880, mode=ops.CONV3X3, cin=320, conv=(n_img, h, w, h, w, 1, False))
    run("L0 tconv3 320", M, 320, 960, mode=ops.TCONV3, cin=320, tconv=(24, h * w))
    run("L1 conv3x3 640->640", M // 4, 640, 5760, mode=ops.CONV3X3, cin=640, conv=(n_img, 36, 64, 36, 64, 1, False))
    run("L1 geglu 640->5120", M // 4, 5120, 640, geglu=True)


if __name__ == "__main__":
    main()

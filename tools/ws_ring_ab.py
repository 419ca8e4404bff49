#!/usr/bin/env python3
"""Times the weights-stationary GEMM shapes of the XL step with the library VDX_LIB_PATH names (A/B of ring depths:
run once per library in the same gpurun call).  Prints one line per shape: us per launch, TFLOP/s, GB/s of A + out (+ residual)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
shapes = [  # M, N, K, residual, geglu   (launches per 24-frame step in profiles/r05_bench_profile_all.json)
    (110592, 640, 640, True, False), (110592, 640, 640, False, False), (110592, 1920, 640, False, False),
    (110592, 5120, 640, False, True), (442368, 320, 320, False, False), (442368, 320, 320, True, False),
    (442368, 960, 320, False, False), (221184, 320, 320, True, False)]
print(os.environ.get("VDX_LIB_PATH", "product library"))
for M, N, K, res, geglu in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.05
    r = torch.randn(M, N, device=dev, dtype=torch.float16, generator=g) if res else None
    kw = dict(M=M, residual=r)
    if geglu:
        kw["geglu"] = True
    out = ops.gemm(a, w, **kw)
    name = ops.gemm_kernel_name(M, N, K, ops.PLAIN, geglu, residual=res)
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, out=out, **kw)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    nbytes = 2 * (M * K + M * (N // 2 if geglu else N) * (2 if res else 1))
    print(f"{M:7d} x {N:5d} x {K:4d} res={int(res)} geglu={int(geglu)}  {best:8.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s  {nbytes / best / 1e3:6.0f} GB/s  "
          f"checksum {float(out.float().abs().mean()):.6f}  {name[:60]}")

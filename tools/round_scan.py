#!/usr/bin/env python3
"""Launch time of the tiled 256x320 GEMM against the tile count around whole rounds of 256 (is an exact-fit grid a cliff?)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
for N, K, mts in ((640, 1920, list(range(352, 392, 4)) + [384, 388, 400, 432]), (1280, 1280, list(range(56, 72, 2)) + [108, 112, 120, 128]),
                  (320, 2880, list(range(740, 776, 4)))):
    w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.03
    a_all = torch.randn(max(mts) * 256, K, device=dev, dtype=torch.float16, generator=g)
    nt = (N + 319) // 320
    print(f"N = {N}, K = {K}: {nt} column tile(s)")
    for mt in sorted(set(mts)):
        M = mt * 256
        a = a_all[:M]
        fn = lambda: ops.gemm(a, w, M=M, variant=2)     # noqa: E731
        for _ in range(2):
            fn()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8 * 1e3)
        tiles = mt * nt
        print(f"  {mt:4d} m-tiles = {tiles:4d} tiles = {tiles / 256:5.3f} rounds: {best:7.1f} us  {best / tiles * 256:6.1f} us per 256 tiles  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s")

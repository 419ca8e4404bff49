#!/usr/bin/env python3
"""Does a kernel on a side stream run BESIDE the step's kernels on this part, or do the two streams take turns?  (The first
RCCL-contention sweep of round 6 showed the step growing by exactly the occupancy hog's duration whatever R.)
Compute stream: `n` launches of one kernel family; side stream: one occupancy hog of `R` workgroups for about as long.
Serial = t_compute + t_hog; overlapped = max of the two (+ what the held CUs cost)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def main():
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    M = 110592
    a640 = torch.randn(M, 640, device=dev, dtype=torch.float16, generator=g)
    w640 = torch.randn(640, 640, device=dev, dtype=torch.float16, generator=g) * 0.04
    a1280 = torch.randn(27648, 1280, device=dev, dtype=torch.float16, generator=g)
    w1280 = torch.randn(1280, 1280, device=dev, dtype=torch.float16, generator=g) * 0.03
    qkv = torch.randn(2 * 2304, 3 * 640, device=dev, dtype=torch.float16, generator=g)
    fams = {
        "gemm_ws 110592x640x640 (persistent, one block per CU)": lambda: ops.gemm(a640, w640, M=M),
        "tiled gemm 27648x1280x1280 (256x320 tiles, 144 KB LDS)": lambda: ops.gemm(a1280, w1280, M=27648),
        "flash 2x2304 keys, 10 heads": lambda: ops.flash_attn(qkv[:, :640], qkv[:, 640:1280], qkv[:, 1280:], n_seq=2, sq=2304, skv=2304,
                                                           skv_pad=2304, heads=10, seq_per_kv=1, scale=0.125, v_rows=True),
    }
    print(ops.gemm_kernel_name(M, 640, 640, ops.PLAIN, False), "|", ops.gemm_kernel_name(27648, 1280, 1280, ops.PLAIN, False))
    n = 40
    for name, fn in fams.items():
        for _ in range(3):
            fn()
        tc = min(timed(lambda: [fn() for _ in range(n)]) for _ in range(3))
        us = int(tc * 1e3)
        for R, lds in ((16, 64 << 10), (16, 0), (64, 64 << 10)):
            def hog_only():
                ops.occupancy_hog(R, lds, us, side)
                side.synchronize()
            th = min(timed(hog_only) for _ in range(2))

            def both():
                ops.occupancy_hog(R, lds, us, side)
                for _ in range(n):
                    fn()
                side.synchronize()
            tb = min(timed(both) for _ in range(3))
            print(f"{name:58s} x{n}: compute {tc:7.2f} ms | hog R={R:3d} lds={lds >> 10:3d}K {th:7.2f} ms | both {tb:7.2f} ms "
                  f"-> {'SERIAL' if tb > 0.85 * (tc + th) else 'overlapped'} (+{100 * (tb / max(tc, th) - 1):.0f} % over the longer)")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How fragile is an exact-fit launch of the tiled GEMM?  Times one launch family alone and beside an occupancy hog of R
workgroups (one or eight CUs held), for row counts that give whole rounds of 256 tiles, whole rounds of 248, and neither."""
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


dev = torch.device("cuda", 0)
side = torch.cuda.Stream(device=dev, priority=-1)
g = torch.Generator(device=dev).manual_seed(0)
N, K = 640, 1920
w = torch.randn(N, K, device=dev, dtype=torch.float16, generator=g) * 0.03
a_all = torch.randn(110592, K, device=dev, dtype=torch.float16, generator=g)
reps = 12
for M, what in ((98304, "768 tiles = 3 rounds of 256"), (95232, "744 tiles = 3 rounds of 248"), (91136, "712 tiles"), (110592, "864 tiles (planner: 768 + a tail)"),
                (65536, "512 tiles = 2 rounds"), (63488, "496 tiles = 2 rounds of 248")):
    a = a_all[:M]
    v = 0 if M == 110592 else 2          # the planner's own choice for the whole product; the 256x320 tile otherwise
    fn = lambda: ops.gemm(a, w, M=M, variant=v)     # noqa: E731
    for _ in range(3):
        fn()
    tc = min(timed(lambda: [fn() for _ in range(reps)]) for _ in range(3))
    line = f"M = {M:6d} ({what:34s}) alone {tc / reps * 1e3:7.1f} us"
    for R in (1, 8):
        def both():
            ops.occupancy_hog(R, 64 << 10, int(tc * 1e3) + 200, side)
            time.sleep(0.0002)
            for _ in range(reps):
                fn()
            side.synchronize()
        tb = min(timed(both) for _ in range(3))
        line += f" | {R} CU(s) held {(tb - 0.2) / reps * 1e3:7.1f} us ({100 * ((tb - 0.2) / tc - 1):+5.1f} %)"
    print(line)

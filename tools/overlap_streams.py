#!/usr/bin/env python3
"""Which HIP streams run BESIDE torch's current (compute) stream, and which take turns with it?  HIP maps streams onto a
few hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams on one queue execute in submission order.  For each of
the first `n` streams torch hands out: a 2 ms occupancy hog of ONE workgroup on it + 2 ms of GEMMs on the compute stream.
    python tools/overlap_streams.py [n] [--pg]     (--pg: create a world-1 NCCL process group first, as bench --rehearse-dist does)"""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 8
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
    if "--pg" in sys.argv:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29547")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        x = torch.ones(8, device=dev)
        dist.all_reduce(x)
    g = torch.Generator(device=dev).manual_seed(0)
    a = torch.randn(27648, 1280, device=dev, dtype=torch.float16, generator=g)
    w = torch.randn(1280, 1280, device=dev, dtype=torch.float16, generator=g) * 0.03
    fn = lambda: ops.gemm(a, w, M=27648)     # noqa: E731
    for _ in range(3):
        fn()
    reps = 22
    tc = min(timed(lambda: [fn() for _ in range(reps)]) for _ in range(3))
    us = int(tc * 1e3)
    if "--pg" in sys.argv:
        # torch.distributed runs a collective on the process group's OWN stream: does that one run beside the compute stream?
        import torch.distributed as dist
        src = torch.zeros(256 << 20, dtype=torch.float16, device=dev)
        dst = torch.empty_like(src)
        side0 = torch.cuda.Stream(device=dev, priority=-1)
        def gathers():
            with torch.cuda.stream(side0):
                for _ in range(4):
                    dist.all_gather_into_tensor(dst, src)
        tg = min(timed(gathers) for _ in range(3))
        def both_pg():
            gathers()
            for _ in range(reps):
                fn()
        tbg = min(timed(both_pg) for _ in range(3))
        print(f"NCCL process-group stream (TORCH_NCCL_HIGH_PRIORITY={os.environ.get('TORCH_NCCL_HIGH_PRIORITY')}): 4 world-1 all-gathers of 512 MB {tg:.2f} ms, "
              f"compute {tc:.2f} ms, both {tbg:.2f} ms -> {'SERIAL' if tbg > 0.85 * (tc + tg) else 'beside'}")
        del src, dst
    streams = [torch.cuda.Stream(device=dev) for _ in range(n)] + [torch.cuda.Stream(device=dev, priority=-1) for _ in range(n)]
    for i, s in enumerate(streams):
        def both():
            ops.occupancy_hog(1, 0, us, s)
            for _ in range(reps):
                fn()
            s.synchronize()
        tb = min(timed(both) for _ in range(3))

        def both2():          # compute first, hog second: the order the store's prefetch is enqueued in
            for _ in range(reps):
                fn()
            ops.occupancy_hog(1, 0, us, s)
            s.synchronize()
        tb2 = min(timed(both2) for _ in range(3))
        print(f"stream {i} ({'high' if i >= n else 'normal'} priority, {s.cuda_stream:#x}): compute {tc:.2f} ms, hog {us / 1e3:.2f} ms -> hog first {tb:.2f} ms ({'SERIAL' if tb > 1.7 * tc else 'beside'}), "
              f"compute first {tb2:.2f} ms ({'SERIAL' if tb2 > 1.7 * tc else 'beside'})")


if __name__ == "__main__":
    main()

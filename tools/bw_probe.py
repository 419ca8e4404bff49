import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx
from vdx import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
M = 442368
for (N, K, res, v) in [(64, 320, False, 5), (64, 320, False, 1), (320, 320, False, 2), (320, 320, True, 2), (320, 64, False, 2), (320, 64, True, 2), (320, 128, True, 2), (320, 640, True, 2)]:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    r = torch.randn(M, N, device=dev, dtype=torch.float16) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ms = timeit(lambda: ops.gemm(a, w, M=M, residual=r, out=out, variant=v))
    byts = 2.0 * (M * K + M * N * (2 if res else 1))
    print(f"N={N:4d} K={K:4d} res={res!s:5} v{v}: {ms:.3f} ms  {byts/ms/1e6:6.0f} GB/s  {2.0*M*N*K/ms/1e9:6.0f} TF/s")
# reference: plain copy and add kernels from torch
x = torch.randn(M, 320, device=dev, dtype=torch.float16); y = torch.empty_like(x); z = torch.randn_like(x)
ms = timeit(lambda: y.copy_(x)); print(f"torch copy  : {ms:.3f} ms {2*x.numel()*2/ms/1e6:6.0f} GB/s")
ms = timeit(lambda: torch.add(x, z, out=y)); print(f"torch add   : {ms:.3f} ms {3*x.numel()*2/ms/1e6:6.0f} GB/s")

import os, sys, torch
sys.path.insert(0, os.getcwd())
import vdx
from vdx import ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for M, N, K in [(2048, 1280, 1280), (2048, 1280, 3840), (2048, 1280, 5120), (2048, 3840, 1280), (4608, 1280, 1280), (4608, 1280, 3840), (4608, 1280, 5120), (4608, 3840, 1280), (8192, 640, 1920), (8192, 640, 2560)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half(); out = torch.empty(M, N, device=dev, dtype=torch.float16)
    r = {}
    for v in (1, 9, 8, 2):
        try:
            r[v] = t(lambda: ops.gemm(a, w, M=M, variant=v, out=out))
        except Exception as ex:
            r[v] = float("nan")
    o1 = ops.gemm(a, w, M=M, variant=1).clone(); o9 = ops.gemm(a, w, M=M, variant=9)
    fl = 2.0 * M * N * K
    print(f"{M:6d} {N:5d} {K:6d}: " + "  ".join(f"v{v} {r[v]*1e3:7.1f} us ({fl/r[v]/1e9:5.0f} TF)" for v in r) + f"   bits v1==v9: {bool(torch.equal(o1, o9))}", flush=True)

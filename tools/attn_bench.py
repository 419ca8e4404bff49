#!/usr/bin/env python3
"""Dev tool: time flash_attn / temporal_attn / norms on the XL step shapes (24 f @ 72x128)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*s):
    return torch.randn(*s, device=dev, dtype=torch.float16)


for lvl, (hw, C) in enumerate([(9216, 320), (2304, 640), (576, 1280), (144, 1280)]):
    n_seq, heads = 48, C // 64
    M = n_seq * hw
    qk = rnd(M, 2 * C)
    vt = rnd(C, M)
    out = torch.empty(M, C, device=dev, dtype=torch.float16)
    ms = timeit(lambda: ops.flash_attn(qk[:, :C], qk[:, C:], vt, n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw, heads=heads,
                                       seq_per_kv=1, scale=0.125, out=out))
    fl = 4.0 * n_seq * heads * hw * hw * 64
    print(f"L{lvl} self-attn  S={hw:5d} heads={heads:2d}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  (V^T operand)")
    # V as rows of one q|k|v matrix (what the UNet runs since round 3), and the whole chain either way
    qkv = rnd(M, 3 * C)
    msr = timeit(lambda: ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw,
                                        heads=heads, seq_per_kv=1, scale=0.125, out=out, v_rows=True))
    print(f"L{lvl} self-attn  V as rows        : {msr:8.3f} ms  {fl / msr / 1e9:7.1f} TFLOP/s")
    ln, w3 = rnd(M, C), rnd(3 * C, C) * 0.05

    def chain_t():
        a = ops.gemm(ln, w3[:2 * C], M=M)
        b_ = ops.gemm(w3[2 * C:], ln, M=C)
        ops.flash_attn(a[:, :C], a[:, C:], b_, n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw, heads=heads, seq_per_kv=1, scale=0.125, out=out)

    def chain_r():
        a = ops.gemm(ln, w3, M=M)
        ops.flash_attn(a[:, :C], a[:, C:2 * C], a[:, 2 * C:], n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw, heads=heads, seq_per_kv=1,
                       scale=0.125, out=out, v_rows=True)
    mt, mr = timeit(chain_t), timeit(chain_r)
    print(f"L{lvl} projection + attention      : q|k GEMM + V^T GEMM + flash {mt:8.3f} ms   q|k|v GEMM + flash(rows) {mr:8.3f} ms")
    del qkv, ln, w3
    # peaked attention (q, k x4: scores ~ N(0, 16 nat), row maxima ~ +60 nat): the lazy softmax offset has to
    # move — the uniform-attention figure above is the never-moves best case
    qk4 = qk * 4
    ms4 = timeit(lambda: ops.flash_attn(qk4[:, :C], qk4[:, C:], vt, n_seq=n_seq, sq=hw, skv=hw, skv_pad=hw, heads=heads,
                                        seq_per_kv=1, scale=0.125, out=out))
    print(f"L{lvl} self-attn  peaked (q,k x4)   : {ms4:8.3f} ms  {fl / ms4 / 1e9:7.1f} TFLOP/s  ({ms4 / ms:.3f}x the uniform time)")
    del qk4
    q = rnd(M, C)
    k = rnd(2 * 128, C)
    vtx = rnd(C, 2 * 128)
    ms = timeit(lambda: ops.flash_attn(q, k, vtx, n_seq=n_seq, sq=hw, skv=77, skv_pad=128, heads=heads, seq_per_kv=24,
                                       scale=0.125, out=out))
    print(f"L{lvl} cross-attn S={hw:5d} kv=77     : {ms:8.3f} ms  {2.0 * M * C * 2 / ms / 1e6:7.0f} GB/s (q+out)")
    qkv = rnd(M, 3 * C)
    ms = timeit(lambda: ops.temporal_attn(qkv, B=2, F=24, HW=hw, heads=heads, scale=0.125, out=out))
    print(f"L{lvl} temporal-attn              : {ms:8.3f} ms  {2.0 * M * C * 4 / ms / 1e6:7.0f} GB/s")
    x = rnd(M, C)
    g, b = rnd(C), rnd(C)
    y = torch.empty_like(x)
    ms = timeit(lambda: ops.groupnorm(x, g, b, groups=32, n_samples=48, rows_per_sample=hw, eps=1e-5, silu_act=True, out=y))
    print(f"L{lvl} groupnorm4d+silu           : {ms:8.3f} ms  {2.0 * M * C * 3 / ms / 1e6:7.0f} GB/s (2R+1W)")
    ms = timeit(lambda: ops.groupnorm(x, g, b, groups=32, n_samples=2, rows_per_sample=24 * hw, eps=1e-5, silu_act=True, out=y))
    print(f"L{lvl} groupnorm5d+silu           : {ms:8.3f} ms  {2.0 * M * C * 3 / ms / 1e6:7.0f} GB/s (2R+1W)")
    ms = timeit(lambda: ops.layernorm(x, g, b, M=M, out=y))
    print(f"L{lvl} layernorm                  : {ms:8.3f} ms  {2.0 * M * C * 2 / ms / 1e6:7.0f} GB/s (1R+1W)")

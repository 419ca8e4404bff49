#!/usr/bin/env python3
"""Dev tool: candidate builds of the K1 kernel (tools/k1_abl.sh -> csrc/build/abl/libk1_<tag>.so) timed side by side in one
process on the level-0 shapes of the XL step, interleaved rounds, median; each candidate's output is compared with the first
one's (same summation order -> expected bit-identical).  The product's CONV3X3 GEMM on a pre-normalised input runs beside them.
    python tools/k1_lab.py [--frames 24,16] [tag ...]"""
import argparse
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tags", nargs="*")
ap.add_argument("--frames", default="24,16")
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
abl = glob.glob(os.path.join(ROOT, "dec*", "csrc", "build", "abl"))[0]
tags = args.tags or sorted(os.path.basename(f)[6:-3] for f in glob.glob(os.path.join(abl, "libk1_*.so")))
libs = {}
for t in tags:
    lib = C.CDLL(os.path.join(abl, f"libk1_{t}.so"))
    fn = lib.vdx_conv3x3_gn_f16
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                   C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    libs[t] = fn
dev = torch.device("cuda:0")
SHAPES = [("320->320", 320, 0), ("640+320->320", 640, 320), ("320+320->320", 320, 320)]
hh, ww, cout = 72, 128, 320
for F in (int(f) for f in args.frames.split(",")):
    for name, c1, c2 in SHAPES:
        n = 2 * F
        M, Cc = n * hh * ww, c1 + c2
        xa = torch.randn(M, c1, device=dev).half()
        xb = torch.randn(M, c2, device=dev).half() if c2 else None
        ab = torch.stack([torch.rand(n, Cc, device=dev) + 0.5, torch.randn(n, Cc, device=dev) * 0.3], dim=-1).contiguous()
        w = (torch.randn(cout, 9 * Cc, device=dev) / (9 * Cc) ** 0.5).half()
        bias = (torch.randn(cout, device=dev) * 0.1).half()
        te = (torch.randn(2, cout, device=dev) * 0.3).half()
        res = torch.randn(M, cout, device=dev).half()
        outs = {t: torch.empty(M, cout, device=dev, dtype=torch.float16) for t in libs}
        nbuf = torch.randn(M, Cc, device=dev).half()
        gout = torch.empty(M, cout, device=dev, dtype=torch.float16)
        st = torch.cuda.current_stream().cuda_stream

        def run(t):
            rc = libs[t](xa.data_ptr(), c1, xb.data_ptr() if c2 else None, c2, c1, c2, ab.data_ptr(), w.data_ptr(), bias.data_ptr(),
                         te.data_ptr(), F * hh * ww, cout, res.data_ptr(), cout, outs[t].data_ptr(), cout, n, hh, ww, cout, st)
            assert rc == 0, (t, rc)

        def gemm():
            ops.gemm(nbuf, w, M=M, mode=ops.CONV3X3, bias=bias, bias2=te, rows_per_bias2=F * hh * ww, residual=res,
                     conv=(n, hh, ww, hh, ww, 1, False), out=gout)
        fns = {t: (lambda t=t: run(t)) for t in libs}
        fns["CONV3X3 GEMM"] = gemm
        ts = {k: [] for k in fns}
        for r in range(args.rounds + 1):
            for k, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                if r:
                    ts[k].append(e0.elapsed_time(e1))
        med = {k: sorted(v)[len(v) // 2] for k, v in ts.items()}
        first = outs[tags[0]]
        same = {t: bool(torch.equal(outs[t], first)) for t in tags[1:]}
        fl = 2.0 * M * cout * 9 * Cc
        print(f"F {F:2d} {name:14s}: " + "  ".join(f"{k} {v:6.3f} ms ({fl / v / 1e9:5.0f})" for k, v in med.items())
              + "   bit-identical to " + tags[0] + ": " + str(same), flush=True)

#!/usr/bin/env python3
"""Dev tool: the tiled-GEMM launches of the XL step's main shapes, replayed on the product library and on candidate builds
(tools/gemm_abl.sh -> csrc/build/abl/libvdx_<tag>.so) side by side: one process, interleaved rounds, median; outputs compared
bit for bit with the product's.   python tools/gemm_lab.py [--frames 24] [tag ...]"""
import argparse
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx import _lib, ops  # noqa: E402
from vdx._lib import GemmArgs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("tags", nargs="*")
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--rounds", type=int, default=7)
args = ap.parse_args()
abl = glob.glob(os.path.join(ROOT, "dec*", "csrc", "build", "abl"))[0]
tags = args.tags or sorted(os.path.basename(f)[7:-3] for f in glob.glob(os.path.join(abl, "libvdx_*.so")))
lib = _lib.load()
fns = {"product": lib.vdx_gemm_f16}
for t in tags:
    l2 = C.CDLL(os.path.join(abl, f"libvdx_{t}.so"))
    l2.vdx_gemm_f16.restype = C.c_int
    l2.vdx_gemm_f16.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    fns[t] = l2.vdx_gemm_f16
captured = []
orig = lib.vdx_gemm_f16


def capture(gref, st):
    captured.append(GemmArgs.from_buffer_copy(bytes(gref._obj)))
    return orig(gref, st)


dev = torch.device("cuda:0")
F = args.frames
n = 2 * F


def rnd(*s):
    return torch.randn(*s, device=dev).half()


# (name, rows, N, K, mode, conv geometry | tconv geometry)
SHAPES = [("L0 conv3x3 320->320", n * 9216, 320, 2880, ops.CONV3X3, (n, 72, 128, 72, 128, 1, False)),
          ("L0 conv3x3 960->320", n * 9216, 320, 8640, ops.CONV3X3, (n, 72, 128, 72, 128, 1, False)),
          ("L1 conv3x3 640->640", n * 2304, 640, 5760, ops.CONV3X3, (n, 36, 64, 36, 64, 1, False)),
          ("L2 conv3x3 1280->1280", n * 576, 1280, 11520, ops.CONV3X3, (n, 18, 32, 18, 32, 1, False)),
          ("L1 tconv3 640", n * 2304, 640, 1920, ops.TCONV3, (F, 2304)),
          ("L2 tconv3 1280", n * 576, 1280, 3840, ops.TCONV3, (F, 576)),
          ("L0 ff2 1280->320", n * 9216, 320, 1280, ops.PLAIN, None),
          ("L1 ff2 2560->640", n * 2304, 640, 2560, ops.PLAIN, None),
          ("L2 qkv 1280->3840", n * 576, 3840, 1280, ops.PLAIN, None),
          ("L2 geglu-as-plain 1280->10240", n * 576, 10240, 1280, ops.PLAIN, None),
          ("L2 ff2 5120->1280", n * 576, 1280, 5120, ops.PLAIN, None)]
for name, M, N, K, mode, geo in SHAPES:
    cin = K // 9 if mode == ops.CONV3X3 else K // 3 if mode == ops.TCONV3 else K
    a, w, bias, res = rnd(M, cin), (rnd(N, K) / K ** 0.5).half(), rnd(N), rnd(M, N)
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    kw = dict(M=M, mode=mode, bias=bias, residual=res, out=out)
    if mode == ops.CONV3X3:
        kw["conv"] = geo
    elif mode == ops.TCONV3:
        kw["tconv"] = geo
    captured.clear()
    lib.vdx_gemm_f16 = capture
    ops.gemm(a, w, **kw)
    lib.vdx_gemm_f16 = orig
    calls = list(captured)
    outs = {t: torch.empty_like(out) for t in fns}
    st = torch.cuda.current_stream().cuda_stream

    def run(t):
        for g in calls:
            g.out = outs[t].data_ptr()
            rc = fns[t](C.byref(g), st)
            assert rc == 0, (t, rc)
    ts = {t: [] for t in fns}
    for r in range(args.rounds + 1):
        for t in fns:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(t)
            e1.record()
            torch.cuda.synchronize()
            if r:
                ts[t].append(e0.elapsed_time(e1))
    med = {t: sorted(v)[len(v) // 2] for t, v in ts.items()}
    same = {t: bool(torch.equal(outs[t], outs["product"])) for t in fns if t != "product"}
    fl = 2.0 * M * N * K
    print(f"{name:30s} M {M:7d} launches {len(calls)}: " + "  ".join(f"{t} {v:6.3f} ms ({fl / v / 1e9:5.0f})" for t, v in med.items())
          + f"   same bits: {same}", flush=True)
    del a, w, res, out, outs

#!/usr/bin/env python3
"""Dev tool: side-by-side timing of flash-attention builds (csrc/build/abl/libflash_<tag>.so from tools/flash_abl.sh) in
ONE process, interleaved rounds (median and min per build), on the spatial self-attention shapes of the XL step, V as
rows of one q|k|v matrix (what the UNet runs).  Builds whose tag does not start with "no" / "mfmaonly" are also checked
against torch (fp32 softmax attention on a sample of sequences / heads).  Prints the phase stamps of the `stamps` build.

    python tools/flash_lab.py [--levels 0,1] [--rounds 7] [--tags base,swap,...] [--frames 24]"""
import argparse
import ctypes as C
import glob
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--levels", default="0")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--tags", default="")
ap.add_argument("--frames", type=int, default=24)
ap.add_argument("--peaked", action="store_true", help="q, k x4: the lazy softmax offset has to move")
args = ap.parse_args()
dev = torch.device("cuda:0")
abl = glob.glob(os.path.join(ROOT, "dec*", "csrc", "build", "abl"))[0]
libs = {}
for f in sorted(glob.glob(os.path.join(abl, "libflash_*.so"))):
    tag = os.path.basename(f)[len("libflash_"):-3]
    if args.tags and tag not in args.tags.split(","):
        continue
    lib = C.CDLL(f)
    _vp, _i, _f = C.c_void_p, C.c_int, C.c_float
    lib.vdx_flash_attn_rows_f16.restype = _i
    lib.vdx_flash_attn_rows_f16.argtypes = [_vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _i, _vp]
    lib.vdx_last_error.restype = C.c_char_p
    libs[tag] = lib
order = sorted(libs, key=lambda t: (t != "base", t))
print("builds:", " ".join(order), flush=True)


def run(lib, qkv, out, Cc, n_seq, hw, heads):
    st = torch.cuda.current_stream().cuda_stream
    p = qkv.data_ptr()
    rc = lib.vdx_flash_attn_rows_f16(p, 3 * Cc, p + 2 * Cc, 3 * Cc, p + 4 * Cc, 3 * Cc, out.data_ptr(), Cc, n_seq, hw, hw, hw,
                                     heads, 1, 0.125, 0, st)
    if rc:
        raise RuntimeError(lib.vdx_last_error().decode())


SHAPES = [(9216, 320), (2304, 640), (576, 1280)]
for lvl in (int(x) for x in args.levels.split(",")):
    hw, Cc = SHAPES[lvl]
    n_seq, heads = 2 * args.frames, Cc // 64
    M = n_seq * hw
    torch.manual_seed(lvl)
    qkv = torch.randn(M, 3 * Cc, device=dev, dtype=torch.float16)
    if args.peaked:
        qkv[:, :2 * Cc] *= 4
    outs = {t: torch.zeros(M, Cc, device=dev, dtype=torch.float16) for t in order}
    times = {t: [] for t in order}
    for r in range(args.rounds + 1):
        for t in order:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(libs[t], qkv, outs[t], Cc, n_seq, hw, heads)
            e1.record()
            torch.cuda.synchronize()
            if r:
                times[t].append(e0.elapsed_time(e1))
    # reference on a sample: sequences 0 and n_seq-1, heads 0 and heads-1, 512 query rows spread over the sequence
    refs = {}
    for sq in (0, n_seq - 1):
        for hd in (0, heads - 1):
            blk = qkv[sq * hw:(sq + 1) * hw].float()
            q = blk[::max(hw // 512, 1), hd * 64:(hd + 1) * 64]
            k = blk[:, Cc + hd * 64:Cc + (hd + 1) * 64]
            v = blk[:, 2 * Cc + hd * 64:2 * Cc + (hd + 1) * 64]
            refs[(sq, hd)] = torch.softmax(q @ k.t() * 0.125, -1) @ v
    fl = 4.0 * n_seq * heads * hw * hw * 64
    base_med = None
    print(f"level {lvl}: {n_seq} seq x {hw} tokens x {heads} heads, {fl / 1e12:.2f} TFLOP" + ("  (peaked)" if args.peaked else ""))
    for t in order:
        ts = sorted(times[t])
        med, mn = ts[len(ts) // 2], ts[0]
        base_med = base_med or med
        err = ""
        if not (t.startswith("no") or t.startswith("mfmaonly")):
            worst = 0.0
            for (sq, hd), ref in refs.items():
                got = outs[t][sq * hw:(sq + 1) * hw][::max(hw // 512, 1), hd * 64:(hd + 1) * 64].float()
                worst = max(worst, float((got - ref).abs().max() / ref.abs().max()))
            same = "" if t == "base" or "base" not in outs else ("  bits==base" if torch.equal(outs[t], outs["base"]) else "  bits!=base")
            err = f"  max rel err {worst:.2e}{same}"
        print(f"  {t:14s} median {med:7.3f} ms  min {mn:7.3f}  {fl / med / 1e9:7.1f} TFLOP/s  x{med / base_med:5.3f} of base{err}", flush=True)
    for stag in [t for t in libs if t.startswith("stamps")]:
        import numpy as np
        lib = libs[stag]
        ppk = "pp" in stag
        lib.vdx_flash_stamps_read.restype = C.c_int
        lib.vdx_flash_stamps_read.argtypes = [C.c_void_p, C.c_size_t]
        nb = min(n_seq * heads * ((hw + 511) // 512), 4096) * 2 if ppk else min(n_seq * heads * ((hw + 255) // 256), 8192)
        buf = np.zeros((nb * 4, 8), dtype=np.uint64)
        run(lib, qkv, outs[stag], Cc, n_seq, hw, heads)
        torch.cuda.synchronize()
        assert lib.vdx_flash_stamps_read(buf.ctypes.data, buf.nbytes) == 0
        b = buf.astype(np.float64)
        tiles = b[:, 5]
        ok = tiles > 0
        per = b[ok, :4] / tiles[ok, None]
        print("  stamps (shader cycles per K/V tile and wave, median [p10 .. p90] over %d waves; %d tiles per wave):" % (ok.sum(), int(np.median(tiles[ok]))))
        names = ("matrix segment (S + P.V MFMAs)", "barrier after it", "vector segment (DMA + softmax)", "barrier after it (incl. vmcnt(0))") if ppk else \
            ("DMA issue", "S MFMAs + row max", "exp / cvt / sums + P.V MFMAs", "barrier (incl. vmcnt(0))")
        for i, nm in enumerate(names):
            col = per[:, i]
            print(f"    {nm:36s} {np.median(col):8.0f}  [{np.percentile(col, 10):7.0f} .. {np.percentile(col, 90):7.0f}]")
        tot = b[ok, 4] / tiles[ok]
        print(f"    {'whole tile (loop only)':30s} {np.median(tot):8.0f}   = 1024 MFMA cycles -> {1024 / np.median(tot) * 100:.1f} % of one wave's share; x2 waves per SIMD")
    del outs, qkv

#!/usr/bin/env python3
"""Dev experiment: the CFG step's two batch items as TWO forwards on two HIP streams (tails of one half's launches filled by
the other half's blocks, HBM-bound passes of one beside MFMA-bound kernels of the other) against the one batch-2 forward.
    python tools/dual_stream_ab.py [frames]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cfg = UNet3DConfig.zeroscope()
model = UNet3DConditionModel(cfg)
model.load_diffusers_state_dict(synthetic_state_dict(cfg, seed=0, device=dev), device=dev)
lat = torch.randn(2, 4, frames, 72, 128, device=dev, dtype=torch.float16)
ehs = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
halves = [(lat[i:i + 1].contiguous(), ehs[i:i + 1].contiguous()) for i in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def joint():
    return model(lat, 500, ehs)


def serial_halves():
    return [model(l, 500, e) for l, e in halves]


def dual():
    cur = torch.cuda.current_stream()
    outs = []
    for st, (l, e) in zip((s1, s2), halves):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(model(l, 500, e))
    cur.wait_stream(s1)
    cur.wait_stream(s2)
    return outs


def timed(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n, out


for rnd in range(2):
    tj, wj, oj = timed(joint)
    ts, ws, os_ = timed(serial_halves)
    td, wd, od = timed(dual)
    same_s = all(torch.equal(os_[i].sample[0], oj.sample[i]) for i in range(2))
    same_d = all(torch.equal(od[i].sample[0], oj.sample[i]) for i in range(2))
    print(f"{frames} frames: batch-2 forward {tj:.2f} ms (wall {wj:.1f})   two batch-1 forwards, one stream {ts:.2f} (wall {ws:.1f})   "
          f"two streams {td:.2f} (wall {wd:.1f})   bits equal to batch-2: serial {same_s}, dual {same_d}", flush=True)

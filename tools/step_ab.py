#!/usr/bin/env python3
"""Dev tool: same-process A/B of model switches on the XL step (boxes of the pool differ by +-4 %, so only pairs measured
in one process mean anything).  Usage: step_ab.py [frames] attr=a,b [attr=a,b ...]   e.g.  step_ab.py 24 spatial_v_rows=1,0 ff_block_bytes=0,134217728
(bool attributes take 0 / 1; others take integers, 0 = None)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
frames = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 24
switches = [a.split("=") for a in sys.argv[1:] if "=" in a]
cfg = UNet3DConfig.zeroscope()
model = UNet3DConditionModel(cfg)
model.load_diffusers_state_dict(synthetic_state_dict(cfg, seed=0, device=dev), device=dev)
lat = torch.randn(2, 4, frames, 72, 128, device=dev, dtype=torch.float16)
ehs = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)


def step_ms(n=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        model(lat, 500, ehs)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, vals in switches:
    cast = (lambda v: bool(int(v))) if isinstance(getattr(model, name), bool) else (lambda v: int(v) or None)
    vals = [cast(v) for v in vals.split(",")]
    if len(vals) == 1:          # a setting for the comparisons that follow
        setattr(model, name, vals[0])
        continue
    res = {v: [] for v in vals}
    for v in vals:
        setattr(model, name, v)
        step_ms(1)
    for _ in range(5):
        for v in vals:
            setattr(model, name, v)
            res[v].append(step_ms())
    print(f"{frames} frames, {name}: " + "   ".join(f"{v}: {sorted(r)[len(r) // 2]:.2f} ms (min {min(r):.2f})" for v, r in res.items()), flush=True)
    setattr(model, name, vals[0])

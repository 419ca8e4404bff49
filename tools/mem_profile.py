#!/usr/bin/env python3
"""Dev tool: where the activation peak of one XL forward (24 f) sits — peak allocated bytes per building block."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 24
SHARD = len(sys.argv) > 2 and sys.argv[2] == "shard"       # memory-lean sharded mode (world of one: full shards stay resident)
cfg = UNet3DConfig.zeroscope()
model = UNet3DConditionModel(cfg)
model.load_diffusers_state_dict(synthetic_state_dict(cfg, seed=0, device=dev), device=dev)
if SHARD:
    model.shard_(0, 1)
lat = torch.randn(2, 4, FRAMES, 72, 128, device=dev, dtype=torch.float16)
ehs = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
torch.cuda.empty_cache()
base = torch.cuda.memory_allocated()
print(f"resident before forward (weights + inputs): {base / 2**30:.2f} GiB")
rec = []


def wrap(name):
    fn = getattr(model, name)

    def f(*a, **k):
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        before = torch.cuda.memory_allocated()
        out = fn(*a, **k)
        torch.cuda.synchronize()
        rec.append((name, a[0], (before - base) / 2**30, (torch.cuda.max_memory_allocated() - base) / 2**30))
        a = k = None
        return out
    setattr(model, name, f)


for n in ("_resnet", "_temp_conv", "_spatial_transformer", "_temporal_transformer"):
    wrap(n)
model(lat, 500, ehs)
torch.cuda.synchronize()
rec.sort(key=lambda r: -r[3])
print("block                                   live-at-entry GiB   peak-inside GiB   (above weights+inputs)")
for name, p, before, peak in rec[:12]:
    print(f"{name:22s} {p:34s} {before:8.2f} {peak:16.2f}")

#!/usr/bin/env python3
"""Dev tool: per-kernel GPU time of one XL step at two frame counts (default 24 and 16), and which kernels carry the
distance of t16 / t24 from 16 / 24.  torch.profiler kernel activities, one process.
    python tools/kernel_ratio.py [24 16]"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
frames = [int(a) for a in sys.argv[1:] if a.isdigit()] or [24, 16]
cfg = UNet3DConfig.zeroscope()
model = UNet3DConditionModel(cfg)
model.load_diffusers_state_dict(synthetic_state_dict(cfg, seed=0, device=dev), device=dev)
ehs = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
tab = {}
for F in frames:
    lat = torch.randn(2, 4, F, 72, 128, device=dev, dtype=torch.float16)
    for _ in range(2):
        model(lat, 500, ehs)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        model(lat, 500, ehs)
        torch.cuda.synchronize()
    agg = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        if ev.device_type == torch.autograd.DeviceType.CUDA:
            a = agg[ev.name[:70]]
            a[0] += ev.device_time / 1e3
            a[1] += 1
    tab[F] = agg
fa, fb = frames
names = set(tab[fa]) | set(tab[fb])
rows = []
for n in names:
    ta, ca = tab[fa].get(n, (0.0, 0))
    tb, cb = tab[fb].get(n, (0.0, 0))
    rows.append((tb - ta * fb / fa, n, ta, ca, tb, cb))
rows.sort(reverse=True)
print(f"{'kernel':70s} {'ms@' + str(fa):>8s} {'n':>4s} {'ms@' + str(fb):>8s} {'n':>4s} {'ratio':>6s} {'excess':>7s}")
for ex, n, ta, ca, tb, cb in rows:
    if max(ta, tb) < 0.05:
        continue
    print(f"{n:70s} {ta:8.2f} {ca:4d} {tb:8.2f} {cb:4d} {tb / ta if ta else 0:6.3f} {ex:7.2f}")
sa, sb = sum(v[0] for v in tab[fa].values()), sum(v[0] for v in tab[fb].values())
print(f"{'sum of kernels':70s} {sa:8.2f}      {sb:8.2f}      {sb / sa:6.3f} {sb - sa * fb / fa:7.2f}")

#!/usr/bin/env python3
"""Dev tool: where one Zeroscope-XL step (72x128 latents, CFG batch 2; frames on the command line,
default 24; two frame counts add a ratio table) spends its time, by building
block and resolution level.  Wraps the UNet's block methods with HIP events (serialises nothing: events
are recorded on the launch stream and read after the step)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
cfg = UNet3DConfig.zeroscope()
model = UNet3DConditionModel(cfg)
model.load_diffusers_state_dict(synthetic_state_dict(cfg, seed=0, device=dev), device=dev)
FRAMES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [24]
H, W = 72, 128
ehs = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)

records = []


def wrap(name, level_of):
    fn = getattr(model, name)

    def timed(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        records.append((name, level_of(a), e0, e1))
        return out
    setattr(model, name, timed)


# rows per frame identify the level: 9216 / 2304 / 576 / 144
wrap("_resnet", lambda a: a[6] * a[7])
wrap("_temp_conv", lambda a: a[4])
wrap("_spatial_transformer", lambda a: a[5] * a[6])
wrap("_temporal_transformer", lambda a: a[4])

tables = {}
for F in FRAMES:
    lat = torch.randn(2, 4, F, H, W, device=dev, dtype=torch.float16)
    for it in range(3):
        records.clear()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        model(lat, 500, ehs)
        t1.record()
        torch.cuda.synchronize()
    total = t0.elapsed_time(t1)
    agg = collections.OrderedDict()
    for name, lvl, e0, e1 in records:
        k = (name, lvl)
        ms, n = agg.get(k, (0.0, 0))
        agg[k] = (ms + e0.elapsed_time(e1), n + 1)
    acc = 0.0
    print(f"{'block':24s} {'rows/frame':>10s} {'calls':>5s} {'ms':>8s} {'ms/call':>8s}")
    for (name, lvl), (ms, n) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        acc += ms
        print(f"{name:24s} {lvl:10d} {n:5d} {ms:8.2f} {ms / n:8.3f}")
    print(f"blocks {acc:.1f} ms of step {total:.1f} ms (rest: conv_in/out, up/downsamplers, time embedding, transformer_in counted as temporal@9216)")
    tables[F] = (dict(agg), total)
if len(FRAMES) == 2:
    fa, fb = FRAMES
    (ta, tota), (tb, totb) = tables[fa], tables[fb]
    print(f"\nratio t{fb} / t{fa} (ideal {fb / fa:.3f}); excess = ms at {fb} frames above the ideal share")
    for k in sorted(ta, key=lambda k: -(tb[k][0] - ta[k][0] * fb / fa)):
        print(f"{k[0]:24s} {k[1]:10d} {ta[k][0]:8.2f} {tb[k][0]:8.2f}  ratio {tb[k][0] / ta[k][0]:.3f}  excess {tb[k][0] - ta[k][0] * fb / fa:6.2f} ms")
    print(f"{'step':35s} {tota:8.2f} {totb:8.2f}  ratio {totb / tota:.3f}  excess {totb - tota * fb / fa:6.2f} ms")

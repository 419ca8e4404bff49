#!/usr/bin/env python3
"""Peak HBM of the MONOLITHIC single-GPU denoising step (weights resident, whole clip in one window) at the total
frame counts of the BASELINE configurations: the denominator of north_star's "per-device peak HBM <= 15 % of the
monolithic single-GPU footprint" (measured as the reference measures it, `torch.cuda.max_memory_allocated`,
fsdp_chunked_coherent.py:255).  Writes profiles/monolithic_peaks.json {frames: GiB}; bench.py reads it for
`peak_hbm_frac_of_monolithic`.

    python tools/mono_peaks.py [24 48 96]
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402
from vdx._lib import source_sha  # noqa: E402
from vdx.scheduler import DDIMScheduler  # noqa: E402
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
frames = [int(a) for a in sys.argv[1:]] or [24, 48, 96]
cfg = UNet3DConfig.zeroscope()
unet = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
sched = DDIMScheduler()
sched.set_timesteps(50, device=dev)
emb = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
out = {}
for T in frames:
    torch.cuda.empty_cache()
    lat = torch.randn(1, 4, T, 72, 128, device=dev, dtype=torch.float16)
    for i in range(2):
        if i == 1:
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        x = ops.cfg_input(lat, None, 0.0)
        noise = unet(x, 981, encoder_hidden_states=emb).sample
        lat = sched.step_cfg(noise, 981, lat, 7.5)
        del x, noise
    e1.record()
    torch.cuda.synchronize()
    out[str(T)] = round(torch.cuda.max_memory_allocated() / 2 ** 30, 3)
    print(f"monolithic {T:3d} frames: peak {out[str(T)]:.3f} GiB, {e0.elapsed_time(e1):.1f} ms/step, finite "
          f"{bool(torch.isfinite(lat.float()).all())}", flush=True)
    del lat
out["_meta"] = {"source_sha": source_sha(), "what": "torch.cuda.max_memory_allocated over one monolithic CFG step, GiB"}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "monolithic_peaks.json"), "w"), indent=1)

#!/usr/bin/env python3
"""Dev tool: the reference's default job on one GPU — 32 frames, 50 DDIM steps, hybrid_ctx (windows (0,16),(12,28),
(24,32)) at Zeroscope-XL size with seeded weights; prints wall time of the denoising, finiteness and peak HBM."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import vdx
from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser
from vdx.scheduler import DDIMScheduler
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
from vdx.weights import synthetic_state_dict
dev = torch.device("cuda:0")
cfg = UNet3DConfig.zeroscope()
unet = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
emb = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
dc = DiffuserConfig(num_frames=32, steps=50, device="cuda")        # the reference's defaults: 32 frames, 50 steps, hybrid_ctx
d = DistributedVideoDiffuser(dc, unet, DDIMScheduler(), emb[1:], emb[:1])
t0 = time.time(); lat, info = d(); torch.cuda.synchronize(); t1 = time.time()
print("ranges", info["ranges"], "denoise_s", round(info["denoise_s"], 2), "finite", bool(torch.isfinite(lat).all()), "lat std", float(lat.std()))
print("peak GiB", torch.cuda.max_memory_allocated() / 2**30)

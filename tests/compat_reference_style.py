"""Driver for tests/test_compat_gpu.py, run as `python -m vdx.compat.run tests/compat_reference_style.py`.

Written with the CALLS the reference's strategy script makes, in its order (`Distribution/strategies/
fsdp_chunked_coherent.py`: imports :3-22, pynvml :41-45, process group :50, pipeline :55-61, FSDP wrap :63-88,
scheduler / tokenizer / text encoder :95-103, context :105-127, `_denoise` :129-143, decode :219-225, boundary
metrics and mp4 :227-253, memory :255-262) — but it is NOT that file: the reference's Python does not travel to the
GPU box.  It exists to show the unchanged call surface works on the shims and the HIP modules, FSDP wrap included."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import cv2                                                        # -> vdx.compat.cv2_shim
import pynvml                                                     # -> vdx.compat.pynvml_shim
from torch.nn import ModuleList, Sequential, ModuleDict
from torch.distributed.fsdp import FullyShardedDataParallel as FSDP, CPUOffload, MixedPrecision
from torch.distributed.fsdp.fully_sharded_data_parallel import ShardingStrategy
from diffusers import DiffusionPipeline                           # -> vdx.compat.diffusers_shim

local_rank = int(os.getenv("LOCAL_RANK", 0))
pynvml.nvmlInit()
torch.cuda.set_device(local_rank)


def vram_mb():
    h = pynvml.nvmlDeviceGetHandleByIndex(local_rank)
    return pynvml.nvmlDeviceGetMemoryInfo(h).used // 1024 ** 2


def main():
    device = "cuda"
    dist.init_process_group("nccl")
    pipe = DiffusionPipeline.from_pretrained("synthetic:tiny", torch_dtype=torch.float16, low_cpu_mem_usage=True,
                                             use_safetensors=False, device_map=None)
    plain_unet = pipe.unet
    mp = MixedPrecision(torch.float16, torch.float16, torch.float16)

    def wrap_policy(module, recurse, nonwrapped_numel):
        if isinstance(module, (ModuleList, Sequential, ModuleDict)):
            return False
        return nonwrapped_numel >= 10_000_000

    fsdp_kwargs = dict(auto_wrap_policy=wrap_policy, sharding_strategy=ShardingStrategy.FULL_SHARD,
                       cpu_offload=CPUOffload(offload_params=True), mixed_precision=mp,
                       device_id=torch.cuda.current_device(), use_orig_params=False)
    unet = FSDP(pipe.unet, **fsdp_kwargs)
    text_encoder = FSDP(pipe.text_encoder, **fsdp_kwargs)
    vae = pipe.vae
    for nm, sm in vae.named_children():
        if any(p.requires_grad for p in sm.parameters()):
            setattr(vae, nm, FSDP(sm, **fsdp_kwargs))
    pipe.unet, pipe.text_encoder, pipe.vae = unet, text_encoder, vae

    steps, T, H, W = 3, 5, 128, 256
    pipe.scheduler.set_timesteps(steps, device=device)
    toks = pipe.tokenizer(["a panda", ""], padding="max_length", max_length=pipe.tokenizer.model_max_length,
                          truncation=True, return_tensors="pt")
    with torch.no_grad():
        emb = text_encoder(toks.input_ids.to(device))[0]
    cond_emb, uncond_emb = emb[:1], emb[1:]
    C = unet.config.in_channels
    torch.manual_seed(0)
    base = torch.randn(1, C, T, H // 8, W // 8, device=device, dtype=torch.float16) * pipe.scheduler.init_noise_sigma
    ctx = base.mean(dim=2, keepdim=True)
    dist.broadcast(ctx, src=0)

    def denoise(unet_, lat):
        for t in pipe.scheduler.timesteps:
            x = pipe.scheduler.scale_model_input(torch.cat([lat] * 2), t)
            x = x + 0.35 * ctx.repeat(1, 1, lat.shape[2], 1, 1)
            e = torch.cat([uncond_emb, cond_emb], dim=0)
            with torch.no_grad():
                noise = unet_(x, t, encoder_hidden_states=e).sample
            u, c = noise.chunk(2)
            lat = pipe.scheduler.step(u + 7.5 * (c - u), t, lat).prev_sample
        return lat

    lat = denoise(unet, base.clone())
    assert lat.shape == base.shape and bool(torch.isfinite(lat.float()).all())
    assert torch.equal(lat, denoise(plain_unet, base.clone())), "the FSDP wrapper changed the result"
    # the shim's UNet verifies on the device that the script's torch.cat batch holds the same tensor twice and computes the
    # text-independent blocks once; with that switched off the result must carry the same bits
    assert plain_unet.last_forward_shared_prefix and plain_unet.detect_cfg_duplicate
    plain_unet.detect_cfg_duplicate = False
    assert torch.equal(lat, denoise(plain_unet, base.clone())) and not plain_unet.last_forward_shared_prefix
    plain_unet.detect_cfg_duplicate = True
    frames = []
    for i in range(T):
        z = lat[:, :, i].float().to(device)                       # (the reference's blended latent is fp32)
        with torch.no_grad():
            img_lat = vae.decode(z / 0.18215).sample
        img = (img_lat[0].permute(1, 2, 0) * 0.5 + 0.5).clamp(0, 1)
        frames.append((img * 255).byte().cpu().numpy())
    assert frames[0].shape == (H, W, 3)
    prev_gray, next_gray = cv2.cvtColor(frames[1], cv2.COLOR_BGR2GRAY), cv2.cvtColor(frames[2], cv2.COLOR_BGR2GRAY)
    flow = cv2.calcOpticalFlowFarneback(prev_gray, next_gray, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    h, w = prev_gray.shape
    warp = cv2.remap(frames[1], (np.arange(w)[None, :] + flow[:, :, 0]).astype(np.float32),
                     (np.arange(h)[:, None] + flow[:, :, 1]).astype(np.float32), cv2.INTER_LINEAR)
    flow_err = float(np.mean(np.abs(warp.astype(np.float32) - frames[2].astype(np.float32))))
    out = sys.argv[1] if len(sys.argv) > 1 else "out.mp4"
    vw = cv2.VideoWriter(out, cv2.VideoWriter_fourcc(*"mp4v"), 8, (W, H))
    for f in frames:
        vw.write(cv2.cvtColor(f, cv2.COLOR_RGB2BGR))
    vw.release()
    peak_all = torch.tensor(torch.cuda.max_memory_allocated() // 1024 ** 2, device=device)
    dist.all_reduce(peak_all, op=dist.ReduceOp.MAX)
    print(f"COMPAT-OK frames {len(frames)} flow_err {flow_err:.3f} peak_mb {int(peak_all.item())} end_vram_mb {vram_mb()} "
          f"mp4_bytes {os.path.getsize(out)} unet {type(unet).__name__} in_channels {C}")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""`DistributedVideoDiffuser` — the hybrid FSDP + frame-chunked denoiser of
`Distribution/strategies/fsdp_chunked_coherent.py:47-276`, re-built on the HIP kernels.

Same configuration names as the reference's argparse (`:281-300`): num_frames, steps,
guidance_scale, chunk_size, overlap, height, width, mode {fsdp, chunk, hybrid, hybrid_ctx},
context_weight.  Differences in mechanism (results identical):
  * the per-step arithmetic (ctx injection, CFG combine, DDIM step) runs as two fused kernels;
  * denoised chunks stay on the device and are exchanged as fixed-shape fp16 tensors with
    `torch.distributed.all_gather` (RCCL on GPUs) instead of pickled CPU objects;
  * the linear-ramp blend (:204-217) runs on the device, in the reference's accumulation order.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import ops
from .planner import ChunkPlan, plan


@dataclass
class DiffuserConfig:
    num_frames: int = 32
    steps: int = 50
    guidance_scale: float = 7.5
    chunk_size: int = 0
    overlap: int = 4
    height: int = 576
    width: int = 1024
    mode: str = "hybrid_ctx"
    context_weight: float = 0.35
    device: str = "cuda"
    noise_device: Optional[str] = None     # None = like the reference: generate on `device`
    overlap_rule: str = "coherent"

    @property
    def use_fsdp(self):
        return self.mode in ("fsdp", "hybrid", "hybrid_ctx")

    @property
    def no_chunking(self):
        return self.mode == "fsdp"

    @property
    def use_ctx(self):
        return self.mode == "hybrid_ctx"


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def seeded_noise(shape, sigma, device, noise_device=None, dtype=torch.float16):
    """`torch.manual_seed(0); randn(...) * init_noise_sigma` (:180-182).  RNG streams are
    device-specific, so parity runs pass noise_device="cpu" (SURVEY.md §8 a2)."""
    nd = torch.device(noise_device) if noise_device is not None else torch.device(device)
    torch.manual_seed(0)
    base = torch.randn(*shape, device=nd, dtype=dtype)
    base *= sigma
    return base.to(device)


def ramp_weights(length: int, ov: int) -> torch.Tensor:
    """Per-frame blend weights of one chunk (:206-213), built with the same torch calls."""
    w = torch.ones(length)
    if ov > 0:
        ramp = torch.linspace(0, 1, ov)
        k = min(ov, length)
        w[:k] = ramp[:k]
        w[-k:] = torch.flip(ramp[:k], [0])
    return w


def gather_chunks(mine: List[torch.Tensor], chunk_plan: ChunkPlan, rank: int, world: int):
    """Exchange denoised chunks; returns [(s, e, tensor)] in the reference's blend order
    (rank-major, then the rank's own order — `for lst in gathered: for s,e,latc in lst`, :208-209).
    Fixed-shape exchange: every chunk is padded to `chunk_plan.chunk` frames."""
    per = chunk_plan.per_rank
    assert len(mine) == per
    if world == 1:
        return [(s, e, t) for (s, e), t in zip(chunk_plan.for_rank(0), mine)]
    ref = mine[0]
    _, C, _, H, W = ref.shape
    buf = ref.new_zeros((per, C, chunk_plan.chunk, H, W))
    for i, t in enumerate(mine):
        buf[i, :, :t.shape[2]] = t[0]
    bufs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    out = []
    for r in range(world):
        for i, (s, e) in enumerate(chunk_plan.for_rank(r)):
            out.append((s, e, bufs[r][i:i + 1, :, :e - s].contiguous()))
    return out


class DistributedVideoDiffuser:
    def __init__(self, cfg: DiffuserConfig, unet, scheduler, uncond_emb, cond_emb):
        self.cfg = cfg
        self.rank, self.world = _world()
        self.unet, self.scheduler = unet, scheduler
        self.uncond_emb, self.cond_emb = uncond_emb, cond_emb
        # reference :63-78 — modes fsdp / hybrid / hybrid_ctx shard the UNet's parameters
        if cfg.use_fsdp and self.world > 1 and isinstance(getattr(unet, "W", None), dict):
            unet.shard_(self.rank, self.world)
        scheduler.set_timesteps(cfg.steps, device=cfg.device)
        self.ctx = None
        if cfg.use_ctx:                                               # reference :105-127
            C = unet.config.in_channels
            shape = (1, C, cfg.num_frames, cfg.height // 8, cfg.width // 8)
            if self.rank == 0:
                full = seeded_noise(shape, scheduler.init_noise_sigma, cfg.device, cfg.noise_device)
                ctx = full.mean(dim=2, keepdim=True)
            else:
                ctx = torch.empty((1, C, 1, shape[3], shape[4]), device=cfg.device, dtype=torch.float16)
            if self.world > 1:
                dist.broadcast(ctx, src=0)
            self.ctx = ctx.contiguous()

    def denoise(self, lat: torch.Tensor) -> torch.Tensor:
        """Reference `_denoise` (:129-143) for one chunk."""
        cfg, sched = self.cfg, self.scheduler
        emb = torch.cat([self.uncond_emb, self.cond_emb], dim=0)
        lat = lat.contiguous()
        for t in sched._host_timesteps:
            x = ops.cfg_input(lat, self.ctx, cfg.context_weight)
            noise = self.unet(x, t, encoder_hidden_states=emb).sample
            lat = sched.step_cfg(noise, t, lat, cfg.guidance_scale)
        return lat

    def plan(self) -> ChunkPlan:
        cfg = self.cfg
        return plan(cfg.num_frames, self.world, cfg.chunk_size, cfg.overlap, cfg.no_chunking, cfg.overlap_rule)

    def blend(self, chunks: List[Tuple[int, int, torch.Tensor]], like: torch.Tensor, ov: int) -> torch.Tensor:
        """Reference :204-217 on the device."""
        T = like.shape[2]
        full = torch.zeros_like(like)
        weight = torch.zeros(T, dtype=torch.float32, device=like.device)
        for s, e, lat in chunks:
            ops.blend_accumulate(full, weight, lat.contiguous(), ramp_weights(e - s, ov).to(like.device), s, e)
        return ops.blend_finalize(full, weight)

    def decode_frames(self, lat: torch.Tensor, vae, batch: int = 8) -> List:
        """Reference :219-225: the blended latent (1,C,T,h,w) -> T uint8 (H,W,3) frames (numpy, host).
        `z/0.18215` is formed in the latent's dtype and cast to fp16 at the VAE boundary (what the reference's
        FSDP mixed-precision wrapper does to forward inputs); frames are decoded `batch` at a time instead of one
        by one (frames are independent samples of the decoder)."""
        frames = []
        T = lat.shape[2]
        for i0 in range(0, T, batch):
            z = lat[0, :, i0:i0 + batch].permute(1, 0, 2, 3) / 0.18215
            u8 = vae.decode_frames_u8(z.to(self.cfg.device, torch.float16).contiguous())
            frames += [f for f in u8.cpu().numpy()]
        return frames

    def __call__(self):
        cfg = self.cfg
        T, H, W = cfg.num_frames, cfg.height // 8, cfg.width // 8
        cp = self.plan()
        C = self.unet.config.in_channels
        base = seeded_noise((1, C, T, H, W), self.scheduler.init_noise_sigma, cfg.device, cfg.noise_device)
        t0 = time.time()
        mine = [self.denoise(base[:, :, s:e].clone()) for s, e in cp.for_rank(self.rank)]
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        denoise_s = time.time() - t0
        t0 = time.time()
        chunks = gather_chunks(mine, cp, self.rank, self.world)
        gather_s = time.time() - t0
        lat = self.blend(chunks, base, cp.overlap)
        return lat, {"chunk_size": cp.chunk, "overlap": cp.overlap, "ranges": list(cp.ranges),
                     "world_size": self.world, "num_frames": T, "denoise_s": denoise_s,
                     "net_gather_s": gather_s}

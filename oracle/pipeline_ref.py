"""ORACLE (test infrastructure): CPU restatement of the orchestration the reference
owns on the hot path — `Distribution/strategies/fsdp_chunked_coherent.py`:

  planner            :149-177   (variants `fsdp_chunked.py:136-171`, `chunk_only.py:80-105`)
  chunk -> rank      :184
  shared base noise  :180-182   (+ slice/clone :187)
  global context     :105-127
  CFG denoise loop   :129-143   (`fsdp.py:141-153` for the literal-7.5 variant)
  ramp blend         :204-217

Everything is written as straight-line torch-CPU code with the reference's
dtypes (fp16 tensors, fp32 0-d scalars).  Parity status (round 5 on): PINNED TO THE
EXECUTED REFERENCE — `tests/golden/make_ref_fixtures.py` ran the reference's four strategy
files unmodified (worlds 1-8 over gloo) and `tests/test_ref_exec_host.py` requires `==` /
`torch.equal` between this file and what they computed: 1 575 planner configurations, the
shared noise and its slices, the broadcast context, every denoised chunk of eight
"exact"-UNet jobs (3-50 steps), gather order, every blended frame, the CSV row.  What
stays unpinned is the diffusers boundary those jobs stand in for (oracle/__init__.py).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


class PlannerHang(ValueError):
    """The reference's `while i < T: i += sz - ov` never terminates when ov >= sz."""


def plan_chunks(T: int, world: int, chunk_size: int = 0, overlap: int = 4,
                no_chunking: bool = False, rule: str = "coherent"):
    """-> (cs, ov, ranges).  rule: "coherent" (`fsdp_chunked_coherent.py:158`) or
    "third" (`fsdp_chunked.py:143`, `chunk_only.py:86`: ov = min(overlap, cs // 3))."""
    if no_chunking:                                            # :150-151
        cs, ov = T, 0
    else:
        if chunk_size <= 0:                                    # :153-156
            min_chunk = max(4, T // (world * 2))
            max_chunk = min(16, T // world)
            cs = min(max_chunk, max(min_chunk, T // world))
        else:
            cs = chunk_size
        if rule == "coherent":
            ov = overlap if overlap > 0 else max(4, cs // 3)   # :158
        else:
            ov = min(overlap, cs // 3)

    def compute(sz):                                           # :160-165
        if sz - ov <= 0:
            raise PlannerHang(f"overlap {ov} >= chunk {sz}: the reference loop does not terminate")
        rng, i = [], 0
        while i < T:
            rng.append((i, min(i + sz, T)))
            i += sz - ov
        return rng

    ranges = compute(cs)
    if len(ranges) % world != 0:                               # :168-173
        for d in range(1, cs):
            test = compute(cs + d)
            if len(test) % world == 0:
                cs, ranges = cs + d, test
                break
    if len(ranges) % world != 0:                               # :174-177
        need = world - (len(ranges) % world)
        ranges = ranges + [ranges[-1]] * need
    return cs, ov, ranges


def my_ranges(ranges, world, rank):                            # :184
    return [r for i, r in enumerate(ranges) if i % world == rank]


def base_noise(T, C, H, W, init_noise_sigma=1.0, dtype=torch.float16):
    """:180-182 — generated on CPU (device RNG differs)."""
    torch.manual_seed(0)
    base = torch.randn(1, C, T, H, W, dtype=dtype)
    base *= init_noise_sigma
    return base


def global_context(T, C, H, W, init_noise_sigma=1.0, dtype=torch.float16):
    """:108-119 — same seed and shape as base_noise, mean over the frame axis."""
    torch.manual_seed(0)
    full = torch.randn(1, C, T, H, W, dtype=dtype)
    full *= init_noise_sigma
    return full.mean(dim=2, keepdim=True)


def denoise(unet, sched, lat, uncond_emb, cond_emb, guidance_scale=7.5, ctx=None,
            context_weight=0.35):
    """:129-143.  `unet(x, t, encoder_hidden_states=emb).sample`."""
    for t in sched.timesteps:
        x = sched.scale_model_input(torch.cat([lat] * 2), t)
        if ctx is not None:
            x = x + context_weight * ctx.repeat(1, 1, lat.shape[2], 1, 1)
        emb = torch.cat([uncond_emb, cond_emb], dim=0)
        with torch.no_grad():
            noise = unet(x, t, encoder_hidden_states=emb).sample
        u, c = noise.chunk(2)
        lat = sched.step(u + guidance_scale * (c - u), t, lat).prev_sample
    return lat


def ramp_blend(chunks: List[Tuple[int, int, torch.Tensor]], T: int, ov: int, like: torch.Tensor):
    """:204-217.  `chunks` = flattened gathered list of (s, e, cpu latent (1,C,len,H,W)).
    `like` gives full's dtype/shape (fp16 accumulator); result is fp32 like the reference."""
    full = torch.zeros_like(like)
    weight = torch.zeros((1, 1, T, 1, 1))
    ramp = torch.linspace(0, 1, ov).view(1, 1, ov, 1, 1) if ov > 0 else None
    for s, e, latc in chunks:
        length = e - s
        w = torch.ones((1, 1, length, 1, 1))
        if ov > 0:
            k = min(ov, length)
            w[:, :, :k] = ramp[:, :, :k]
            w[:, :, -k:] = torch.flip(ramp[:, :, :k], [2])
        full[:, :, s:e] += latc * w
        weight[:, :, s:e] += w
    return full / weight.clamp(min=1e-6)


def run_video(unet, sched, T, C, H, W, world, steps, uncond_emb, cond_emb, chunk_size=0, overlap=4,
              mode="hybrid_ctx", guidance_scale=7.5, context_weight=0.35, rule="coherent",
              dtype=torch.float16):
    """Whole `__call__` :145-217 for all ranks executed serially on CPU (what the N-rank job
    computes, without the transport)."""
    sched.set_timesteps(steps)
    no_chunk = mode == "fsdp"
    cs, ov, ranges = plan_chunks(T, world, chunk_size, overlap, no_chunk, rule)
    base = base_noise(T, C, H, W, sched.init_noise_sigma, dtype)
    ctx = global_context(T, C, H, W, sched.init_noise_sigma, dtype) if mode == "hybrid_ctx" else None
    gathered = []
    for rank in range(world):
        for s, e in my_ranges(ranges, world, rank):
            den = denoise(unet, sched, base[:, :, s:e].clone(), uncond_emb, cond_emb,
                          guidance_scale, ctx, context_weight)
            gathered.append((s, e, den))
    return ramp_blend(gathered, T, ov, base), (cs, ov, ranges)

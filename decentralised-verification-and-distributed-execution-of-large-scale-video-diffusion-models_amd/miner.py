"""The second caller of the same UNet / DDIM surface: the InferNet miner's denoising loop
(`InferNet/neurons/miner.py:516-586`) — batch 1, no classifier-free guidance, every intermediate latent z_t and
noise prediction eps_t recorded, because their fp16 BYTES are hashed into the Merkle leaves the validator later
re-checks (`miner.py:180-209`: leaf = sha256(t as 2 big-endian bytes + z bytes + eps bytes)).

Only the loop and the leaf hash are here (the denoising path and the exact bytes it commits to); the Merkle tree /
proof / network protocol around them are InferNet's own product and stay out of scope.  What matters for that
caller is bit-stability: the same inputs must give the same bytes on every run — all kernels on this path use
fixed-order reductions (tests/test_miner_gpu.py runs the loop twice and compares hashes).
"""
from __future__ import annotations

import hashlib
from typing import Dict, List

import torch

from ._lib import VdxError


def leaf_hash(t: int, z: torch.Tensor, eps: torch.Tensor) -> bytes:
    """miner.py:196-203 — sha256 over (timestep as 2 big-endian bytes, z fp16 bytes, eps fp16 bytes)."""
    return hashlib.sha256(int(t).to_bytes(2, "big") + z.cpu().numpy().tobytes() + eps.cpu().numpy().tobytes()).digest()


@torch.no_grad()
def denoise_with_trace(unet, scheduler, z: torch.Tensor, encoder_hidden_states: torch.Tensor, num_steps: int) -> Dict:
    """miner.py:516-586.  z (1,4,T,h,w) fp16 initial noise on the GPU; encoder_hidden_states (1,77,D) fp16.
    Returns {"z": final latent, "latents": [z_t], "noise_preds": [eps_t], "timesteps": [t], "alphas": [abar_t]}."""
    if z.dim() != 5 or z.shape[0] != 1 or z.dtype != torch.float16:
        raise VdxError(f"denoise_with_trace: z must be (1,C,T,h,w) fp16, got {tuple(z.shape)} {z.dtype}")
    scheduler.set_timesteps(num_steps, device=z.device)
    timesteps: List[int] = list(scheduler._host_timesteps)
    alphas = [float(scheduler.alphas_cumprod[t]) for t in timesteps]                     # :534-544
    latents, noise_preds = [], []
    z = z.contiguous()
    for t in timesteps:                                                                   # :573-586
        latents.append(z)
        eps = unet(z, t, encoder_hidden_states=encoder_hidden_states).sample
        noise_preds.append(eps)
        z = scheduler.step(eps, t, z).prev_sample
    return {"z": z, "latents": latents, "noise_preds": noise_preds, "timesteps": timesteps, "alphas": alphas}

"""ORACLE (test infrastructure): fp32 PyTorch-CPU restatement of the decoder half of diffusers'
`AutoencoderKL` (the Stable-Diffusion VAE that Zeroscope ships), plus the reference's frame mapping.

The reference never defines this model; it calls it once per frame after the blend:
  `Distribution/strategies/fsdp_chunked_coherent.py:219-225`
      z = lat[:,:,i].to(cfg.device)
      img_lat = self.vae.decode(z/0.18215).sample
      img = (img_lat[0].permute(1,2,0)*0.5+0.5).clamp(0,1)
      frames.append((img*255).byte().cpu().numpy())
The arithmetic lives in the un-vendored dependency `diffusers` (pins as in unet3d_ref.py), restated here from
its published behaviour (`post_quant_conv` 4->4 1x1, `conv_in` 4->512, mid block = ResNet,
single-head 512-channel attention over h*w tokens, ResNet; four up blocks (512, 512, 256, 128) of three ResNets
with nearest-x2 + conv between them; GroupNorm(32, eps 1e-6) + SiLU + `conv_out` 128->3), composed only from the
torch primitives diffusers composes.  PARITY UNPINNED (oracle/__init__.py); structural pins: decoder parameter
count 49 490 179 (+ 20 for post_quant_conv) and the diffusers state-dict key/shape table.

Module/attribute names reproduce the diffusers state-dict keys (`decoder.*`, `post_quant_conv.*`; attention
projections under the current names `to_q/to_k/to_v/to_out.0`, GroupNorm as `group_norm`).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)   # encoder order; the decoder walks it reversed
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def sd() -> "VaeConfig":
        return VaeConfig()

    @staticmethod
    def tiny(ch=(64, 64, 128, 128)) -> "VaeConfig":
        """Same topology, narrow widths: CPU-sized fixtures."""
        return VaeConfig(block_out_channels=tuple(ch))


class ResnetBlock2DRef(nn.Module):
    """diffusers `ResnetBlock2D(temb_channels=None, eps=1e-6, groups=32, output_scale_factor=1)`."""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=1e-6)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=1e-6)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))
        return (x if self.conv_shortcut is None else self.conv_shortcut(x)) + h


class VaeAttentionRef(nn.Module):
    """diffusers `Attention(C, heads=1, dim_head=C, norm_num_groups=32, eps=1e-6, bias=True,
    residual_connection=True, rescale_output_factor=1, upcast_softmax=True)` on an NCHW map."""

    def __init__(self, C, groups):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, C, eps=1e-6)
        self.to_q = nn.Linear(C, C)
        self.to_k = nn.Linear(C, C)
        self.to_v = nn.Linear(C, C)
        self.to_out = nn.ModuleList([nn.Linear(C, C)])

    def forward(self, x):
        b, c, hh, ww = x.shape
        t = self.group_norm(x).view(b, c, hh * ww).transpose(1, 2)
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        p = torch.softmax(q @ k.transpose(1, 2) * (1.0 / math.sqrt(c)), dim=-1)
        o = self.to_out[0](p @ v)
        return o.transpose(1, 2).reshape(b, c, hh, ww) + x


class UpsampleRef(nn.Module):
    def __init__(self, C):
        super().__init__()
        self.conv = nn.Conv2d(C, C, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class MidBlockRef(nn.Module):
    def __init__(self, C, groups):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(C, C, groups), ResnetBlock2DRef(C, C, groups)])
        self.attentions = nn.ModuleList([VaeAttentionRef(C, groups)])

    def forward(self, x):
        return self.resnets[1](self.attentions[0](self.resnets[0](x)))


class UpDecoderBlockRef(nn.Module):
    def __init__(self, cin, cout, n, groups, upsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2DRef(cin if j == 0 else cout, cout, groups) for j in range(n)])
        self.upsamplers = nn.ModuleList([UpsampleRef(cout)]) if upsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        return x if self.upsamplers is None else self.upsamplers[0](x)


class DecoderRef(nn.Module):
    def __init__(self, cfg: VaeConfig):
        super().__init__()
        rev = tuple(reversed(cfg.block_out_channels))
        g = cfg.norm_num_groups
        self.conv_in = nn.Conv2d(cfg.latent_channels, rev[0], 3, padding=1)
        self.mid_block = MidBlockRef(rev[0], g)
        blocks, prev = [], rev[0]
        for i, ch in enumerate(rev):
            blocks.append(UpDecoderBlockRef(prev, ch, cfg.layers_per_block + 1, g, upsample=i != len(rev) - 1))
            prev = ch
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(g, rev[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(rev[-1], cfg.out_channels, 3, padding=1)

    def forward(self, z):
        x = self.mid_block(self.conv_in(z))
        for b in self.up_blocks:
            x = b(x)
        return self.conv_out(F.silu(self.conv_norm_out(x)))


class AutoencoderKLRef(nn.Module):
    """Decode half only (`decode(z).sample`); the encoder and `quant_conv` are not on the reference's path."""

    def __init__(self, cfg: VaeConfig = VaeConfig()):
        super().__init__()
        self.config = cfg
        self.post_quant_conv = nn.Conv2d(cfg.latent_channels, cfg.latent_channels, 1)
        self.decoder = DecoderRef(cfg)

    def decode(self, z):
        return SimpleNamespace(sample=self.decoder(self.post_quant_conv(z)))


def synthetic_state_dict(cfg: VaeConfig, seed: int = 4321, std: float = 0.02, dtype=torch.float32):
    """diffusers-shaped decoder state dict with seeded synthetic values (fan-in scaled weights, norm gains
    near 1, small biases; second convolutions of residual branches halved so activations stay O(1))."""
    with torch.device("meta"):
        model = AutoencoderKLRef(cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, mod in model.named_modules():
        if isinstance(mod, nn.GroupNorm):
            sd[name + ".weight"] = (1.0 + 0.05 * torch.randn(mod.weight.shape, generator=g)).to(dtype)
            sd[name + ".bias"] = (std * torch.randn(mod.bias.shape, generator=g)).to(dtype)
        elif isinstance(mod, (nn.Linear, nn.Conv2d)):
            w = torch.randn(mod.weight.shape, generator=g) / math.sqrt(mod.weight[0].numel())
            if name.endswith(("conv2", "to_out.0")):
                w = w * 0.5
            sd[name + ".weight"] = w.to(dtype)
            sd[name + ".bias"] = (std * torch.randn(mod.bias.shape, generator=g)).to(dtype)
    return sd


def frames_from_latents(vae, lat: torch.Tensor):
    """`fsdp_chunked_coherent.py:219-225` on CPU: lat (1,4,T,h,w) -> list of T uint8 (H,W,3) arrays.
    The decode input is cast to the VAE's dtype (the reference's FSDP mixed-precision wrapper casts
    forward inputs to fp16); the mapping arithmetic runs in the sample's dtype as written."""
    frames = []
    dt = next(vae.parameters()).dtype
    for i in range(lat.shape[2]):
        z = lat[:, :, i]
        with torch.no_grad():
            img_lat = vae.decode((z / 0.18215).to(dt)).sample
        img = (img_lat[0].permute(1, 2, 0) * 0.5 + 0.5).clamp(0, 1)
        frames.append((img * 255).byte().cpu().numpy())
    return frames

"""ORACLE (test infrastructure): fp32 PyTorch-CPU restatement of diffusers'
`UNet3DConditionModel` for the Zeroscope / ModelScope-t2v configuration.

The reference never defines this model; it calls it:
  `Distribution/strategies/fsdp_chunked_coherent.py:140`
      noise = self.unet(x, t, encoder_hidden_states=emb).sample
and reads `unet.config.in_channels` (`:106,181,194`).  The arithmetic lives in
the un-vendored dependency `diffusers` (pins: `InferNet/requirements.txt:10`
`diffusers>=0.19.0`, `InferNet/setup.py:75` `>=0.21.0`), restated here from its
published behaviour as written out in SURVEY.md Appendix A, composed only from
the torch primitives diffusers itself composes.  PARITY UNPINNED (see
oracle/__init__.py); structural pins: parameter count and state-dict keys.

Module/attribute names reproduce the diffusers state-dict keys so that a real
checkpoint would load with `strict=True`.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


@dataclass
class UNet3DConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attention_head_dim: int = 64          # per-head width; heads = C / 64
    cross_attention_dim: int = 1024
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    transformer_in_heads: int = 8
    down_block_types: Tuple[str, ...] = (
        "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D")
    up_block_types: Tuple[str, ...] = (
        "UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")

    @staticmethod
    def zeroscope() -> "UNet3DConfig":
        return UNet3DConfig()

    @staticmethod
    def tiny(ch=(64, 128, 128, 128), cross=128, in_heads=2) -> "UNet3DConfig":
        """Same topology, narrow widths: used for CPU-sized end-to-end fixtures."""
        return UNet3DConfig(block_out_channels=tuple(ch), cross_attention_dim=cross,
                            transformer_in_heads=in_heads)


# --------------------------------------------------------------------------- #
# leaf pieces
# --------------------------------------------------------------------------- #
def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers `Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)`."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    a = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(a), torch.sin(a)], dim=-1)


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.linear_1 = nn.Linear(cin, cout)
        self.linear_2 = nn.Linear(cout, cout)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class Attention(nn.Module):
    """diffusers `Attention`: to_q/k/v no bias, to_out.0 with bias, scale d^-0.5."""

    def __init__(self, query_dim, heads, dim_head, cross_dim=None):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        kv = cross_dim if cross_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Identity()])

    def forward(self, x, ctx=None):
        ctx = x if ctx is None else ctx
        n, s, _ = x.shape
        q = self.to_q(x).view(n, s, self.heads, self.dim_head).transpose(1, 2)
        k = self.to_k(ctx).view(n, -1, self.heads, self.dim_head).transpose(1, 2)
        v = self.to_v(ctx).view(n, -1, self.heads, self.dim_head).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v)          # scale = dim_head ** -0.5
        o = o.transpose(1, 2).reshape(n, s, self.heads * self.dim_head)
        return self.to_out[0](o)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        a, g = self.proj(x).chunk(2, dim=-1)
        return a * F.gelu(g)                                 # exact (erf) GELU


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Identity(), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        return self.net[2](self.net[0](x))


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_dim=None, double_self_attention=False):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = Attention(dim, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = Attention(dim, heads, dim_head,
                               cross_dim=None if double_self_attention else cross_dim)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.double_self_attention = double_self_attention

    def forward(self, x, ehs=None):
        x = x + self.attn1(self.norm1(x))
        x = x + self.attn2(self.norm2(x), None if self.double_self_attention else ehs)
        x = x + self.ff(self.norm3(x))
        return x


class Transformer2DModel(nn.Module):
    """Spatial transformer, `use_linear_projection=True`."""

    def __init__(self, heads, dim_head, in_channels, cross_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, cross_dim=cross_dim)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, ehs):
        n, c, h, w = x.shape
        r = x
        x = self.norm(x).permute(0, 2, 3, 1).reshape(n, h * w, c)
        x = self.proj_in(x)
        for blk in self.transformer_blocks:
            x = blk(x, ehs)
        x = self.proj_out(x)
        x = x.reshape(n, h, w, c).permute(0, 3, 1, 2)
        return x + r


class TransformerTemporalModel(nn.Module):
    """Temporal transformer, `double_self_attention=True`."""

    def __init__(self, heads, dim_head, in_channels, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = nn.GroupNorm(groups, in_channels, eps=1e-6)
        self.proj_in = nn.Linear(in_channels, inner)
        self.transformer_blocks = nn.ModuleList(
            [BasicTransformerBlock(inner, heads, dim_head, double_self_attention=True)])
        self.proj_out = nn.Linear(inner, in_channels)

    def forward(self, x, num_frames):
        bf, c, h, w = x.shape
        b = bf // num_frames
        r = x
        x5 = x.reshape(b, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        x5 = self.norm(x5)                                     # 5-D GN: stats over (C/G, F, h, w)
        s = x5.permute(0, 3, 4, 2, 1).reshape(b * h * w, num_frames, c)
        s = self.proj_in(s)
        for blk in self.transformer_blocks:
            s = blk(s)
        s = self.proj_out(s)
        y = s.reshape(b, h, w, num_frames, c).permute(0, 3, 4, 1, 2).reshape(bf, c, h, w)
        return y + r


class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb, groups, eps):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=eps)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb, cout)
        self.norm2 = nn.GroupNorm(groups, cout, eps=eps)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None

    def forward(self, x, emb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(emb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return x + h


class TemporalConvLayer(nn.Module):
    """4 x [GN(32) -> SiLU -> Conv3d (3,1,1)] + identity."""

    def __init__(self, dim, groups):
        super().__init__()
        def conv():
            return nn.Conv3d(dim, dim, (3, 1, 1), padding=(1, 0, 0))
        self.conv1 = nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(), conv())
        self.conv2 = nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(), nn.Dropout(0.1), conv())
        self.conv3 = nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(), nn.Dropout(0.1), conv())
        self.conv4 = nn.Sequential(nn.GroupNorm(groups, dim), nn.SiLU(), nn.Dropout(0.1), conv())
        nn.init.zeros_(self.conv4[-1].weight)                  # identity at construction
        nn.init.zeros_(self.conv4[-1].bias)

    def forward(self, x, num_frames):
        bf, c, h, w = x.shape
        x5 = x.reshape(bf // num_frames, num_frames, c, h, w).permute(0, 2, 1, 3, 4)
        y = self.conv4(self.conv3(self.conv2(self.conv1(x5))))
        return (x5 + y).permute(0, 2, 1, 3, 4).reshape(bf, c, h, w)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, stride=2, padding=1)

    def forward(self, x):
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, padding=1)

    def forward(self, x, size=None):
        if size is None:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        else:
            x = F.interpolate(x, size=size, mode="nearest")
        return self.conv(x)


# --------------------------------------------------------------------------- #
# blocks
# --------------------------------------------------------------------------- #
class _DownBlock(nn.Module):
    def __init__(self, cfg, cin, cout, temb, cross_attn, add_down):
        super().__init__()
        g, eps, hd = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
        n = cfg.layers_per_block
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(cin if i == 0 else cout, cout, temb, g, eps) for i in range(n)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(cout, g) for _ in range(n)])
        if cross_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout // hd, hd, cout, cfg.cross_attention_dim, g) for _ in range(n)])
            self.temp_attentions = nn.ModuleList(
                [TransformerTemporalModel(cout // hd, hd, cout, g) for _ in range(n)])
        self.cross_attn = cross_attn
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None

    def forward(self, x, emb, ehs, nf):
        outs = []
        for i, (res, tc) in enumerate(zip(self.resnets, self.temp_convs)):
            x = tc(res(x, emb), nf)
            if self.cross_attn:
                x = self.temp_attentions[i](self.attentions[i](x, ehs), nf)
            outs.append(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
            outs.append(x)
        return x, outs


class _MidBlock(nn.Module):
    def __init__(self, cfg, c, temb):
        super().__init__()
        g, eps, hd = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb, g, eps) for _ in range(2)])
        self.temp_convs = nn.ModuleList([TemporalConvLayer(c, g) for _ in range(2)])
        self.attentions = nn.ModuleList([Transformer2DModel(c // hd, hd, c, cfg.cross_attention_dim, g)])
        self.temp_attentions = nn.ModuleList([TransformerTemporalModel(c // hd, hd, c, g)])

    def forward(self, x, emb, ehs, nf):
        x = self.temp_convs[0](self.resnets[0](x, emb), nf)
        x = self.temp_attentions[0](self.attentions[0](x, ehs), nf)
        x = self.temp_convs[1](self.resnets[1](x, emb), nf)
        return x


class _UpBlock(nn.Module):
    def __init__(self, cfg, cin, cout, prev, temb, cross_attn, add_up):
        super().__init__()
        g, eps, hd = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
        n = cfg.layers_per_block + 1
        res = []
        for i in range(n):
            skip = cin if i == n - 1 else cout
            rin = prev if i == 0 else cout
            res.append(ResnetBlock2D(rin + skip, cout, temb, g, eps))
        self.resnets = nn.ModuleList(res)
        self.temp_convs = nn.ModuleList([TemporalConvLayer(cout, g) for _ in range(n)])
        if cross_attn:
            self.attentions = nn.ModuleList(
                [Transformer2DModel(cout // hd, hd, cout, cfg.cross_attention_dim, g) for _ in range(n)])
            self.temp_attentions = nn.ModuleList(
                [TransformerTemporalModel(cout // hd, hd, cout, g) for _ in range(n)])
        self.cross_attn = cross_attn
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, x, skips: List[torch.Tensor], emb, ehs, nf, upsample_size=None):
        for i, (res, tc) in enumerate(zip(self.resnets, self.temp_convs)):
            x = torch.cat([x, skips.pop()], dim=1)
            x = tc(res(x, emb), nf)
            if self.cross_attn:
                x = self.temp_attentions[i](self.attentions[i](x, ehs), nf)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x, upsample_size)
        return x


# --------------------------------------------------------------------------- #
# top level
# --------------------------------------------------------------------------- #
class UNet3DConditionModelRef(nn.Module):
    def __init__(self, cfg: Optional[UNet3DConfig] = None):
        super().__init__()
        cfg = cfg or UNet3DConfig.zeroscope()
        self.cfg = cfg
        self.config = SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                                      cross_attention_dim=cfg.cross_attention_dim,
                                      block_out_channels=cfg.block_out_channels)
        ch = cfg.block_out_channels
        temb = ch[0] * 4
        self.conv_in = nn.Conv2d(cfg.in_channels, ch[0], 3, padding=1)
        self.time_embedding = TimestepEmbedding(ch[0], temb)
        self.transformer_in = TransformerTemporalModel(
            cfg.transformer_in_heads, cfg.attention_head_dim, ch[0], cfg.norm_num_groups)
        downs, c = [], ch[0]
        for i, t in enumerate(cfg.down_block_types):
            downs.append(_DownBlock(cfg, c, ch[i], temb, t.startswith("CrossAttn"), i != len(ch) - 1))
            c = ch[i]
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = _MidBlock(cfg, ch[-1], temb)
        rev = list(reversed(ch))
        ups, c = [], rev[0]
        for i, t in enumerate(cfg.up_block_types):
            prev, cout = c, rev[i]
            cin = rev[min(i + 1, len(ch) - 1)]
            ups.append(_UpBlock(cfg, cin, cout, prev, temb, t.startswith("CrossAttn"), i != len(ch) - 1))
            c = cout
        self.up_blocks = nn.ModuleList(ups)
        self.conv_norm_out = nn.GroupNorm(cfg.norm_num_groups, ch[0], eps=cfg.norm_eps)
        self.conv_out = nn.Conv2d(ch[0], cfg.out_channels, 3, padding=1)

    def forward(self, sample, timestep, encoder_hidden_states):
        """Call surface of `fsdp_chunked_coherent.py:140`; returns an object with `.sample`."""
        b, _, nf, h, w = sample.shape
        t = torch.as_tensor(timestep)
        if t.dim() == 0:
            t = t[None]
        t = t.expand(b)
        emb = self.time_embedding(timestep_embedding(t, self.cfg.block_out_channels[0]).to(sample.dtype))
        emb = emb.repeat_interleave(nf, dim=0)
        ehs = encoder_hidden_states.repeat_interleave(nf, dim=0)
        x = sample.permute(0, 2, 1, 3, 4).reshape(b * nf, -1, h, w)
        x = self.conv_in(x)
        x = self.transformer_in(x, nf)
        skips = [x]
        for blk in self.down_blocks:
            x, outs = blk(x, emb, ehs, nf)
            skips += outs
        x = self.mid_block(x, emb, ehs, nf)
        nup = 2 ** (len(self.cfg.block_out_channels) - 1)
        forward_size = (h % nup != 0) or (w % nup != 0)
        for i, blk in enumerate(self.up_blocks):
            size = None
            if forward_size and blk.upsamplers is not None:
                size = skips[-len(blk.resnets) - 1].shape[2:]
            x = blk(x, skips, emb, ehs, nf, size)
        x = self.conv_out(F.silu(self.conv_norm_out(x)))
        x = x.reshape(b, nf, -1, h, w).permute(0, 2, 1, 3, 4)
        return SimpleNamespace(sample=x)


# --------------------------------------------------------------------------- #
# synthetic weights (SURVEY §8d "Synthetic inputs"): shared by oracle and product tests
# --------------------------------------------------------------------------- #
def synthetic_state_dict(cfg: UNet3DConfig, seed: int = 1234, std: float = 0.02,
                         dtype=torch.float32) -> "dict[str, torch.Tensor]":
    """diffusers-shaped state dict with seeded synthetic values.

    Linear/Conv weights ~ N(0, std) with fan-in-aware scaling so activations stay O(1)
    through the residual stack; norm weights 1 (+ small jitter), biases small.  Values are
    drawn per key from a generator seeded by (seed, key-order) so the table is reproducible
    without shipping a checkpoint.
    """
    with torch.device("meta"):
        model = UNet3DConditionModelRef(cfg)
    g = torch.Generator().manual_seed(seed)
    sd = {}
    out_proj = ("to_out.0", "ff.net.2", "proj_out", "conv2", "conv4.3")
    for name, mod in model.named_modules():
        if isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
            sd[name + ".weight"] = (1.0 + 0.05 * torch.randn(mod.weight.shape, generator=g)).to(dtype)
            sd[name + ".bias"] = (std * torch.randn(mod.bias.shape, generator=g)).to(dtype)
        elif isinstance(mod, (nn.Linear, nn.Conv2d, nn.Conv3d)):
            fan_in = mod.weight[0].numel()
            w = torch.randn(mod.weight.shape, generator=g) / math.sqrt(fan_in)
            if name.endswith(out_proj):          # out-projections of residual branches
                w = w * 0.5
            sd[name + ".weight"] = w.to(dtype)
            if mod.bias is not None:
                sd[name + ".bias"] = (std * torch.randn(mod.bias.shape, generator=g)).to(dtype)
    return sd


def count_params(cfg: Optional[UNet3DConfig] = None) -> int:
    with torch.device("meta"):
        m = UNet3DConditionModelRef(cfg)
    return sum(p.numel() for p in m.parameters())

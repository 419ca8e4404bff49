"""diffusers state-dict table (names -> shapes) of `UNet3DConditionModel` for a config, and a
seeded synthetic fill — used by bench.py / smoke (no checkpoint or network exists in this
environment; SURVEY.md §8d "Synthetic inputs").  The table is what `load_diffusers_state_dict`
ingests; tests/test_host.py checks it key-for-key against the oracle's module tree.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .unet3d import UNet3DConfig

Shape = Tuple[int, ...]


def state_dict_spec(cfg: UNet3DConfig) -> Dict[str, Shape]:
    spec: Dict[str, Shape] = {}
    ch = cfg.block_out_channels
    temb = ch[0] * 4
    hd = cfg.attention_head_dim

    def wb(name, *wshape, bias=True):
        spec[name + ".weight"] = tuple(wshape)
        if bias:
            spec[name + ".bias"] = (wshape[0],)

    def norm(name, c):
        spec[name + ".weight"] = (c,)
        spec[name + ".bias"] = (c,)

    def attn(name, dim, kv):
        wb(name + ".to_q", dim, dim, bias=False)
        wb(name + ".to_k", dim, kv, bias=False)
        wb(name + ".to_v", dim, kv, bias=False)
        wb(name + ".to_out.0", dim, dim)

    def transformer(name, cin, heads, cross):
        inner = heads * hd
        norm(name + ".norm", cin)
        wb(name + ".proj_in", inner, cin)
        b = name + ".transformer_blocks.0"
        norm(b + ".norm1", inner); norm(b + ".norm2", inner); norm(b + ".norm3", inner)
        attn(b + ".attn1", inner, inner)
        attn(b + ".attn2", inner, cross if cross else inner)
        wb(b + ".ff.net.0.proj", inner * 8, inner)
        wb(b + ".ff.net.2", inner, inner * 4)
        wb(name + ".proj_out", cin, inner)

    def resnet(name, cin, cout):
        norm(name + ".norm1", cin)
        wb(name + ".conv1", cout, cin, 3, 3)
        wb(name + ".time_emb_proj", cout, temb)
        norm(name + ".norm2", cout)
        wb(name + ".conv2", cout, cout, 3, 3)
        if cin != cout:
            wb(name + ".conv_shortcut", cout, cin, 1, 1)

    def tconv(name, c):
        for i, leaf in ((1, 2), (2, 3), (3, 3), (4, 3)):
            norm(f"{name}.conv{i}.0", c)
            wb(f"{name}.conv{i}.{leaf}", c, c, 3, 1, 1)

    wb("conv_in", ch[0], cfg.in_channels, 3, 3)
    wb("time_embedding.linear_1", temb, ch[0])
    wb("time_embedding.linear_2", temb, temb)
    transformer("transformer_in", ch[0], cfg.transformer_in_heads, 0)
    c = ch[0]
    for i, t in enumerate(cfg.down_block_types):
        p = f"down_blocks.{i}"
        for j in range(cfg.layers_per_block):
            resnet(f"{p}.resnets.{j}", c if j == 0 else ch[i], ch[i])
            tconv(f"{p}.temp_convs.{j}", ch[i])
            if t.startswith("CrossAttn"):
                transformer(f"{p}.attentions.{j}", ch[i], ch[i] // hd, cfg.cross_attention_dim)
                transformer(f"{p}.temp_attentions.{j}", ch[i], ch[i] // hd, 0)
        if i != len(ch) - 1:
            wb(f"{p}.downsamplers.0.conv", ch[i], ch[i], 3, 3)
        c = ch[i]
    for j in range(2):
        resnet(f"mid_block.resnets.{j}", ch[-1], ch[-1])
        tconv(f"mid_block.temp_convs.{j}", ch[-1])
    transformer("mid_block.attentions.0", ch[-1], ch[-1] // hd, cfg.cross_attention_dim)
    transformer("mid_block.temp_attentions.0", ch[-1], ch[-1] // hd, 0)
    rev = list(reversed(ch))
    c = rev[0]
    for i, t in enumerate(cfg.up_block_types):
        p = f"up_blocks.{i}"
        prev, cout = c, rev[i]
        cin = rev[min(i + 1, len(ch) - 1)]
        n = cfg.layers_per_block + 1
        for j in range(n):
            skip = cin if j == n - 1 else cout
            rin = prev if j == 0 else cout
            resnet(f"{p}.resnets.{j}", rin + skip, cout)
            tconv(f"{p}.temp_convs.{j}", cout)
            if t.startswith("CrossAttn"):
                transformer(f"{p}.attentions.{j}", cout, cout // hd, cfg.cross_attention_dim)
                transformer(f"{p}.temp_attentions.{j}", cout, cout // hd, 0)
        if i != len(ch) - 1:
            wb(f"{p}.upsamplers.0.conv", cout, cout, 3, 3)
        c = cout
    norm("conv_norm_out", ch[0])
    wb("conv_out", cfg.out_channels, ch[0], 3, 3)
    return spec


def synthetic_state_dict(cfg: UNet3DConfig, seed: int = 1234, device="cpu", dtype=torch.float16):
    """Seeded synthetic values generated directly on `device` (fan-in scaled weights, norm gains
    near 1, small biases) — activations stay O(1) through the residual stack."""
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    out = {}
    for name, shape in state_dict_spec(cfg).items():
        is_norm = ("norm" in name.rsplit(".", 2)[-2]) or name.rsplit(".", 1)[0].endswith(
            (".conv1.0", ".conv2.0", ".conv3.0", ".conv4.0"))
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = torch.randn(shape, generator=g, device=dev, dtype=torch.float32) / math.sqrt(fan_in)
            if name.rsplit(".", 1)[0].endswith(("to_out.0", "ff.net.2", "proj_out", "conv2", "conv4.3")):
                t *= 0.5
        elif is_norm and name.endswith(".weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        else:
            t = 0.02 * torch.randn(shape, generator=g, device=dev, dtype=torch.float32)
        out[name] = t.to(dtype)
    return out

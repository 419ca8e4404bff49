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


def synthetic_vae_state_dict(cfg=None, seed: int = 7, device="cpu", dtype=torch.float16):
    """diffusers-shaped AutoencoderKL DECODER table (+ post_quant_conv) with seeded values: what `vdx.vae.AutoencoderKL.
    load_diffusers_state_dict` ingests when no checkpoint exists (bench / compat runs)."""
    from .vae import VaeConfig
    cfg = cfg or VaeConfig.sd()
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    rev = tuple(reversed(cfg.block_out_channels))
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (torch.randn(co, ci, k, k, generator=g, device=dev) / (ci * k * k) ** 0.5).to(dtype)
        sd[name + ".bias"] = (0.02 * torch.randn(co, generator=g, device=dev)).to(dtype)

    def norm(name, c):
        sd[name + ".weight"] = (1 + 0.05 * torch.randn(c, generator=g, device=dev)).to(dtype)
        sd[name + ".bias"] = (0.02 * torch.randn(c, generator=g, device=dev)).to(dtype)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    conv("decoder.conv_in", rev[0], cfg.latent_channels, 3)
    resnet("decoder.mid_block.resnets.0", rev[0], rev[0])
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", rev[0])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sd[f"{a}.{n}.weight"] = (torch.randn(rev[0], rev[0], generator=g, device=dev) / rev[0] ** 0.5).to(dtype)
        sd[f"{a}.{n}.bias"] = (0.02 * torch.randn(rev[0], generator=g, device=dev)).to(dtype)
    resnet("decoder.mid_block.resnets.1", rev[0], rev[0])
    prev = rev[0]
    for i, ch in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
        prev = ch
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg.out_channels, rev[-1], 3)
    return sd


def synthetic_clip_state_dict(cfg=None, seed: int = 11, device="cpu", dtype=torch.float16):
    """`transformers.CLIPTextModel`-shaped table (keys without the `text_model.` prefix) with seeded values."""
    from .clip_text import CLIPTextConfig
    cfg = cfg or CLIPTextConfig.sd2()
    dev = torch.device(device)
    g = torch.Generator(device=dev).manual_seed(seed)
    d, f = cfg.hidden_size, cfg.intermediate_size
    sd = {}

    def lin(name, co, ci):
        sd[name + ".weight"] = (torch.randn(co, ci, generator=g, device=dev) / ci ** 0.5).to(dtype)
        sd[name + ".bias"] = (0.02 * torch.randn(co, generator=g, device=dev)).to(dtype)

    def norm(name):
        sd[name + ".weight"] = (1 + 0.05 * torch.randn(d, generator=g, device=dev)).to(dtype)
        sd[name + ".bias"] = (0.02 * torch.randn(d, generator=g, device=dev)).to(dtype)

    sd["embeddings.token_embedding.weight"] = (0.5 * torch.randn(cfg.vocab_size, d, generator=g, device=dev)).to(dtype)
    sd["embeddings.position_embedding.weight"] = (0.1 * torch.randn(cfg.max_position_embeddings, d, generator=g, device=dev)).to(dtype)
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{i}"
        norm(p + ".layer_norm1"); norm(p + ".layer_norm2")
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(f"{p}.self_attn.{n}", d, d)
        lin(p + ".mlp.fc1", f, d)
        lin(p + ".mlp.fc2", d, f)
    norm("final_layer_norm")
    return sd

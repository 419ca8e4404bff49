#!/usr/bin/env python3
"""Dev tool: AutoencoderKL decode of the 24 XL frames (72x128 latents -> 576x1024), frames/s and the time by
kernel family (HIP events around every GEMM via ops.PROFILE)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vdx  # noqa: E402,F401
from vdx import ops  # noqa: E402
from vdx.vae import AutoencoderKL, VaeConfig  # noqa: E402

dev = torch.device("cuda:0")
cfg = VaeConfig.sd()


def synthetic(cfg):
    """diffusers-shaped decoder table, seeded, generated on the device (the oracle is test-only)."""
    g = torch.Generator(device=dev).manual_seed(7)
    rev = tuple(reversed(cfg.block_out_channels))
    sd = {}

    def conv(name, co, ci, k):
        sd[name + ".weight"] = (torch.randn(co, ci, k, k, generator=g, device=dev) / (ci * k * k) ** 0.5).half()
        sd[name + ".bias"] = (0.02 * torch.randn(co, generator=g, device=dev)).half()

    def norm(name, c):
        sd[name + ".weight"] = (1 + 0.05 * torch.randn(c, generator=g, device=dev)).half()
        sd[name + ".bias"] = (0.02 * torch.randn(c, generator=g, device=dev)).half()

    def resnet(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    conv("post_quant_conv", 4, 4, 1)
    conv("decoder.conv_in", rev[0], 4, 3)
    resnet("decoder.mid_block.resnets.0", rev[0], rev[0])
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", rev[0])
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sd[f"{a}.{n}.weight"] = (torch.randn(rev[0], rev[0], generator=g, device=dev) / rev[0] ** 0.5).half()
        sd[f"{a}.{n}.bias"] = (0.02 * torch.randn(rev[0], generator=g, device=dev)).half()
    resnet("decoder.mid_block.resnets.1", rev[0], rev[0])
    prev = rev[0]
    for i, ch in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch, 3)
        prev = ch
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", 3, rev[-1], 3)
    return sd


m = AutoencoderKL(cfg).load_diffusers_state_dict(synthetic(cfg), device=dev)
T, batch = 24, int(os.environ.get("BATCH", "8"))
z = torch.randn(T, 4, 72, 128, device=dev, dtype=torch.float16)


def run():
    return [m.decode_frames_u8(z[i:i + batch]) for i in range(0, T, batch)]


run()
torch.cuda.synchronize()
ops.PROFILE = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
out = run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for name, fl, a, b, mnk in ops.PROFILE:
    r = agg[name]
    r[0] += fl; r[1] += a.elapsed_time(b); r[2] += 1
gemm_ms = sum(v[1] for v in agg.values())
tf = sum(v[0] for v in agg.values()) / 1e12
print(f"decode {T} frames (batch {batch}): {ms:.1f} ms = {T / ms * 1e3:.1f} frames/s; GEMM {gemm_ms:.1f} ms, {tf:.1f} TFLOP "
      f"-> {tf / (ms * 1e-3):.0f} TFLOP/s overall; peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB; "
      f"frame mean {float(out[0].float().mean()):.1f}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
    print(f"  {k:50s} {v[2]:4d} launches {v[1]:8.2f} ms {v[0] / (v[1] * 1e-3) / 1e12:7.0f} TFLOP/s")

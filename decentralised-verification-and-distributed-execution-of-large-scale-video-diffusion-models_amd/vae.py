"""`AutoencoderKL` (decode half) on libvdx_hip.so — the call the reference makes per frame after the blend:

    fsdp_chunked_coherent.py:219-225
        img_lat = self.vae.decode(z/0.18215).sample
        img = (img_lat[0].permute(1,2,0)*0.5+0.5).clamp(0,1);  frames.append((img*255).byte().cpu().numpy())

Same call surface as diffusers (`vae.decode(z).sample`, `vae.config.scaling_factor`, `named_children()` through
nn.Module), same state-dict keys (oracle/vae_ref.py restates the arithmetic).  Everything is
the UNet's kernels on channels-last fp16 rows [n*h*w][C]:

  * post_quant_conv (1x1, 4->4) is folded EXACTLY into conv_in: the latent gets a fifth channel of ones, whose
    folded weights carry conv_in(bias of post_quant_conv) — zero padding then still pads the *output* of
    post_quant_conv, as in the reference (a plain bias fold would be wrong on the border pixels);
  * ResNet blocks = GroupNorm+SiLU kernels + implicit-GEMM 3x3 convolutions (residual / 1x1 shortcut fused);
  * mid-block attention (ONE head of 512 channels over h*w tokens): q and k GEMMs, V^T by the swapped GEMM (per
    image), scores = one GEMM per image (fp16 [tokens][tokens], as diffusers' baddbmm materialises them), row softmax
    kernel (fp32, scale applied in fp32), P.V^T GEMM, output projection with fused residual; the value bias is
    folded into the output projection's bias (softmax rows sum to 1);
  * upsamplers = the conv kernel's fused nearest-x2 gather;
  * the video path maps the last rows straight to uint8 HWC frames (`decode_frames_u8`), bit-exact with the
    reference's fp16 mapping given the same decoder output.
The encoder / quant_conv are not on the reference's path: their keys are accepted and dropped.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops, packing
from ._lib import VdxError


GN_PARTITION = 8      # GroupNorm statistics are reduced with the row-slab partition of an 8-frame batch whatever the
                      # batch: a frame decoded alone (as the reference does, :219-225) has the bits of its batched self


@dataclass
class VaeConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215

    @staticmethod
    def sd() -> "VaeConfig":
        return VaeConfig()


class AutoencoderKL(nn.Module):
    def __init__(self, cfg: Optional[VaeConfig] = None):
        super().__init__()
        self.cfg = cfg or VaeConfig()
        self.config = SimpleNamespace(**vars(self.cfg))
        self.W: Dict[str, torch.Tensor] = {}
        self._device = torch.device("cpu")
        for c in self.cfg.block_out_channels:
            if c % 64 != 0:
                raise VdxError(f"AutoencoderKL: channel width {c} is not a multiple of 64")

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor], device=None):
        """Ingest a diffusers-format AutoencoderKL state dict (decoder.* and post_quant_conv.*) and pack it."""
        dev = torch.device(device) if device is not None else self._device
        cfg = self.cfg
        W: Dict[str, torch.Tensor] = {}
        used = set()

        def put(name, t):
            W[name] = t.to(device=dev, dtype=torch.float16).contiguous()

        def get(k):
            used.add(k)
            if k not in sd:
                raise VdxError(f"missing key in state dict: {k}")
            return sd[k].to(dev)

        def norm(p):
            put(p + ".weight", get(p + ".weight"))
            put(p + ".bias", get(p + ".bias"))

        def conv3(p, pad_out=False):
            w, b = packing.pack_conv3x3(get(p + ".weight")), get(p + ".bias")
            put(p + ".weight", packing.pad_rows(w, 64) if pad_out else w)
            put(p + ".bias", packing.pad_rows(b, 64) if pad_out else b)

        def resnet(p, cin, cout):
            norm(p + ".norm1"); conv3(p + ".conv1"); norm(p + ".norm2"); conv3(p + ".conv2")
            if cin != cout:
                put(p + ".conv_shortcut.weight", packing.pack_conv1x1(get(p + ".conv_shortcut.weight")))
                put(p + ".conv_shortcut.bias", get(p + ".conv_shortcut.bias"))

        rev = tuple(reversed(cfg.block_out_channels))
        # conv_in o post_quant_conv, exactly: input channels [z0..z3, 1]
        wq, bq = get("post_quant_conv.weight").float().reshape(cfg.latent_channels, -1), get("post_quant_conv.bias").float()
        wi = get("decoder.conv_in.weight").float()                               # [512][4][3][3]
        folded = torch.cat([torch.einsum("ocyx,ci->oiyx", wi, wq),               # sum_c' W_in[o,c',ky,kx] W_pq[c',c]
                            torch.einsum("ocyx,c->oyx", wi, bq)[:, None]], 1)    # ones channel: W_in . b_pq
        put("decoder.conv_in.weight", packing.pack_conv_in(folded))
        put("decoder.conv_in.bias", get("decoder.conv_in.bias"))
        # mid block
        mb = "decoder.mid_block"
        resnet(mb + ".resnets.0", rev[0], rev[0])
        a = mb + ".attentions.0"
        norm(a + ".group_norm")
        for n in "qk":
            put(f"{a}.to_{n}.weight", packing.pack_conv1x1(get(f"{a}.to_{n}.weight")))
            put(f"{a}.to_{n}.bias", get(f"{a}.to_{n}.bias"))
        put(a + ".to_v.weight", packing.pack_conv1x1(get(a + ".to_v.weight")))    # issued as the swapped GEMM (V^T)
        wo = packing.pack_conv1x1(get(a + ".to_out.0.weight"))
        put(a + ".to_out.0.weight", wo)
        # P.(V + 1 b_v^T) = P.V + b_v  (rows of P sum to 1): the value bias moves behind the output projection
        put(a + ".to_out.0.bias", get(a + ".to_out.0.bias").float() + wo.float() @ get(a + ".to_v.bias").float())
        resnet(mb + ".resnets.1", rev[0], rev[0])
        # up blocks
        prev = rev[0]
        for i, ch in enumerate(rev):
            for j in range(cfg.layers_per_block + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else ch, ch)
            if i != len(rev) - 1:
                conv3(f"decoder.up_blocks.{i}.upsamplers.0.conv")
            prev = ch
        norm("decoder.conv_norm_out")
        conv3("decoder.conv_out", pad_out=True)
        extra = {k for k in sd if k not in used and not k.startswith(("encoder.", "quant_conv."))}
        if extra:
            raise VdxError(f"unexpected keys in state dict: {sorted(extra)[:5]} ... ({len(extra)})")
        self.W, self._device = W, dev
        return self

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        if self.W:
            probe = fn(torch.empty(0, dtype=torch.float16, device=self._device))
            self.W = {k: v.to(probe.device) for k, v in self.W.items()}
            self._device = probe.device
        return out

    def num_parameters(self) -> int:
        return sum(v.numel() for v in self.W.values())

    # ------------------------------------------------------------------------------------------
    def _resnet(self, p, x, n, hh, ww):
        W, g = self.W, self.cfg.norm_num_groups
        M, S = n * hh * ww, hh * ww
        geo = (n, hh, ww, hh, ww, 1, False)
        h = ops.groupnorm(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"], groups=g, n_samples=n, rows_per_sample=S,
                          eps=1e-6, silu_act=True, partition_samples=GN_PARTITION)
        h = ops.gemm(h, W[p + ".conv1.weight"], M=M, mode=ops.CONV3X3, bias=W[p + ".conv1.bias"], conv=geo)
        h = ops.groupnorm(h, W[p + ".norm2.weight"], W[p + ".norm2.bias"], groups=g, n_samples=n, rows_per_sample=S,
                          eps=1e-6, silu_act=True, partition_samples=GN_PARTITION)
        sc = x
        if p + ".conv_shortcut.weight" in W:
            sc = ops.gemm(x, W[p + ".conv_shortcut.weight"], M=M, bias=W[p + ".conv_shortcut.bias"])
        return ops.gemm(h, W[p + ".conv2.weight"], M=M, mode=ops.CONV3X3, bias=W[p + ".conv2.bias"], residual=sc, conv=geo)

    def _attention(self, p, x, n, hh, ww):
        W, g = self.W, self.cfg.norm_num_groups
        S, M, C = hh * ww, n * hh * ww, x.shape[1]
        if S % 64 != 0:
            raise VdxError(f"AutoencoderKL attention needs h*w % 64 == 0 (got {hh}x{ww})")
        t = ops.groupnorm(x, W[p + ".group_norm.weight"], W[p + ".group_norm.bias"], groups=g, n_samples=n,
                          rows_per_sample=S, eps=1e-6, silu_act=False, partition_samples=GN_PARTITION)
        q = ops.gemm(t, W[p + ".to_q.weight"], M=M, bias=W[p + ".to_q.bias"])
        k = ops.gemm(t, W[p + ".to_k.weight"], M=M, bias=W[p + ".to_k.bias"])
        o = torch.empty((M, C), dtype=torch.float16, device=x.device)
        scores = torch.empty((S, S), dtype=torch.float16, device=x.device)
        vt = torch.empty((C, S), dtype=torch.float16, device=x.device)
        scale = 1.0 / math.sqrt(C)
        for i in range(n):                                                        # one image at a time: [S][S] scores
            rows = slice(i * S, (i + 1) * S)
            ops.gemm(W[p + ".to_v.weight"], t[rows], M=C, out=vt)                 # V^T [C][S] (its bias: see load)
            ops.gemm(q[rows], k[rows], M=S, out=scores)                           # q . k^T  (k rows = the GEMM's W)
            ops.softmax_rows(scores, rows=S, cols=S, scale=scale)
            ops.gemm(scores, vt, M=S, out=o[rows])
        return ops.gemm(o, W[p + ".to_out.0.weight"], M=M, bias=W[p + ".to_out.0.bias"], residual=x)

    def _decode_rows(self, z):
        """z (n,4,h,w) fp16 on the GPU -> channels-last rows [n*H*W][64] (RGB in the first 3 columns), H, W."""
        cfg, W = self.cfg, self.W
        if not W:
            raise VdxError("AutoencoderKL: no weights loaded")
        if z.dim() != 4 or z.shape[1] != cfg.latent_channels:
            raise VdxError(f"AutoencoderKL.decode: expected (n,{cfg.latent_channels},h,w), got {tuple(z.shape)}")
        if not z.is_cuda:
            raise VdxError("AutoencoderKL.decode: expected a GPU tensor (the decode path has no CPU fallback)")
        n, _, hh, ww = z.shape
        z5 = torch.cat([z.to(torch.float16), torch.ones_like(z[:, :1], dtype=torch.float16)], 1)
        x = ops.conv_in(z5.unsqueeze(2).contiguous(), W["decoder.conv_in.weight"], W["decoder.conv_in.bias"])
        mb = "decoder.mid_block"
        x = self._resnet(mb + ".resnets.0", x, n, hh, ww)
        x = self._attention(mb + ".attentions.0", x, n, hh, ww)
        x = self._resnet(mb + ".resnets.1", x, n, hh, ww)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, n, hh, ww)
            if i != nb - 1:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = ops.gemm(x, W[p + ".weight"], M=n * 4 * hh * ww, mode=ops.CONV3X3, bias=W[p + ".bias"],
                             conv=(n, hh, ww, 2 * hh, 2 * ww, 1, True))
                hh, ww = 2 * hh, 2 * ww
        t = ops.groupnorm(x, W["decoder.conv_norm_out.weight"], W["decoder.conv_norm_out.bias"], groups=cfg.norm_num_groups,
                          n_samples=n, rows_per_sample=hh * ww, eps=1e-6, silu_act=True, partition_samples=GN_PARTITION)
        y = ops.gemm(t, W["decoder.conv_out.weight"], M=n * hh * ww, mode=ops.CONV3X3, bias=W["decoder.conv_out.bias"],
                     conv=(n, hh, ww, hh, ww, 1, False))
        return y, hh, ww

    @torch.no_grad()
    def decode(self, z, return_dict=True):
        """diffusers surface: `.decode(z).sample` -> (n,3,H,W) fp16."""
        y, H, Wd = self._decode_rows(z)
        n = z.shape[0]
        out = ops.rows_to_ncfhw(y, n, self.cfg.out_channels, 1, H, Wd).reshape(n, self.cfg.out_channels, H, Wd)
        return SimpleNamespace(sample=out) if return_dict else (out,)

    @torch.no_grad()
    def decode_frames_u8(self, z):
        """z (n,4,h,w) -> uint8 (n,H,W,3) on the GPU: decode + the reference's frame mapping (:224-225) in one pass."""
        y, H, Wd = self._decode_rows(z)
        return ops.rows_to_u8_frames(y, z.shape[0], H, Wd)

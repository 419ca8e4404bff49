"""`UNet3DConditionModel` on hand-written gfx950 kernels, behind the diffusers call surface the
reference uses (`Distribution/strategies/fsdp_chunked_coherent.py:140`):

    noise = unet(x, t, encoder_hidden_states=emb).sample          # x (2,4,F,h,w) fp16

and `unet.config.in_channels` (`:106,181,194`).  Architecture = diffusers UNet3DConditionModel for
the Zeroscope/ModelScope config; weights are ingested from a diffusers
state dict (same keys/shapes) and re-laid-out once for the kernels (packing.py).

Physical layout: every activation is a channels-last fp16 row matrix [B*F*h*w][C]; the permutes
diffusers performs between (BF,C,H,W), (B,C,F,H,W) and (BHW,F,C) are index arithmetic inside the
kernels (conv gathers, 5-D GroupNorm statistics, temporal attention strides).

There is NO CPU path: all arithmetic goes through libvdx_hip.so (ops.py raises otherwise).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops, packing
from ._lib import VdxError


@dataclass
class UNet3DConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    attention_head_dim: int = 64
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


TEXT_PAD = 128   # text tokens (77) padded to 2 key tiles; also keeps the K/V GEMM N % 64 == 0


class UNet3DConditionModel(nn.Module):
    """diffusers-compatible module; weights live in `self.W` (packed fp16 device tensors)."""

    def __init__(self, cfg: Optional[UNet3DConfig] = None):
        super().__init__()
        self.cfg = cfg or UNet3DConfig.zeroscope()
        c = self.cfg
        if c.attention_head_dim != 64:
            raise VdxError("attention kernels are built for head_dim 64")
        for ch in c.block_out_channels:
            if ch % 64 != 0:
                raise VdxError("block_out_channels must be multiples of 64")
        if c.cross_attention_dim % 64 != 0:
            raise VdxError("cross_attention_dim must be a multiple of 64")
        self.config = SimpleNamespace(in_channels=c.in_channels, out_channels=c.out_channels,
                                      cross_attention_dim=c.cross_attention_dim,
                                      block_out_channels=c.block_out_channels,
                                      sample_size=None)
        self.W: Dict[str, torch.Tensor] = {}
        self._temb_slices: Dict[str, Tuple[int, int]] = {}
        self._device = torch.device("cpu")
        self.dtype = torch.float16
        self.ff_block_bytes = None     # memory-lean feed-forward (see _ff); shard_() turns it on
        self.lean_concat = True        # (with ff_block_bytes set) concat GroupNorm -> conv1 over pieces of whole frames
        self.lean_attn = True          # (with ff_block_bytes set) spatial self-attention over image halves
        self.fold_norm_proj_in = True  # GroupNorm -> proj_in as per-sample weights where the Linear is weights-stationary
        self.fuse_ff = True            # K8 where the width allows (False: LayerNorm, GEGLU GEMM, GEMM + residual)
        self.fuse_cross_attn = True    # K5 where the width and the text length allow (False: LayerNorm, q GEMM, flash attention, GEMM + residual)
        self.fuse_proj_out = True      # (with K8) the transformer's proj_out + residual behind the feed-forward, in K8's kernel (False: a GEMM of its own)
        self.fuse_conv_gn = True       # K1 where it is faster — level 0 (False: GroupNorm apply pass + 3x3-conv GEMM; "always": wherever the shape allows, tests)
        self.fuse_tconv = True         # K3 where it is faster — level 0 (False: GroupNorm apply pass + temporal-conv GEMM; "always": wherever the shape allows, tests)
        self.spatial_v_rows = True     # spatial self-attention on one q|k|v projection, V as rows (False: round-2 form, A/B timing only)
        self.fuse_temporal_attention = True    # K7 where the shape allows (False: always the separate kernels)
        self.share_cfg_prefix = True           # a CFG batch built by ops.cfg_input: the blocks before the first cross-attention run on ONE item (forward)
        self.detect_cfg_duplicate = False      # untagged batch of two: COMPARE the halves on the device (one kernel + a host sync) before sharing
        self._text_ref = None                  # (encoder_hidden_states object, its version, (B, device), padded copy)
        self._text_kv: Dict[str, tuple] = {}   # cross-attention K / V^T of that text, per transformer

    # ------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------
    def _resnet_names(self) -> List[str]:
        c = self.cfg
        names = []
        for i, t in enumerate(c.down_block_types):
            names += [f"down_blocks.{i}.resnets.{j}" for j in range(c.layers_per_block)]
        names += ["mid_block.resnets.0", "mid_block.resnets.1"]
        for i, t in enumerate(c.up_block_types):
            names += [f"up_blocks.{i}.resnets.{j}" for j in range(c.layers_per_block + 1)]
        return names

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor], device=None):
        """Ingest a diffusers-format state dict (keys/shapes of SURVEY.md §8c (ii)) and pack it."""
        dev = torch.device(device) if device is not None else self._device
        W: Dict[str, torch.Tensor] = {}

        def put(name, t):
            W[name] = t.to(device=dev, dtype=torch.float16).contiguous()

        used = set()

        def get(k):
            used.add(k)
            if k not in sd:
                raise VdxError(f"missing key in state dict: {k}")
            return sd[k]

        def lin(prefix, bias=True):
            put(prefix + ".weight", packing.pack_conv1x1(get(prefix + ".weight")))
            if bias:
                put(prefix + ".bias", get(prefix + ".bias"))

        def norm(prefix):
            put(prefix + ".weight", get(prefix + ".weight"))
            put(prefix + ".bias", get(prefix + ".bias"))

        def attn_self(prefix, temporal):
            # one [3*inner][inner] projection for both kinds (the spatial flash kernel takes V as rows: round 3)
            q, k, v = (packing.pack_conv1x1(get(f"{prefix}.to_{n}.weight")) for n in "qkv")
            put(prefix + ".to_qkv.weight", torch.cat([q, k, v], 0))
            if temporal and q.shape[0] in packing.K7_GEOMETRY and q.shape[0] not in packing.K7B_WIDTHS:      # widths only the first fused sub-block kernel (K7) is built for
                put(prefix + ".k7_qkv", packing.pack_k7_qkv(q, k, v))
                put(prefix + ".k7_out", packing.pack_k7_out(packing.pack_conv1x1(get(f"{prefix}.to_out.0.weight"))))
            lin(prefix + ".to_out.0")

        def attn_cross(prefix):
            for n in "qkv":
                put(f"{prefix}.to_{n}.weight", packing.pack_conv1x1(get(f"{prefix}.to_{n}.weight")))
            lin(prefix + ".to_out.0")

        def ff(prefix):
            wp, bp = packing.pack_geglu(get(prefix + ".net.0.proj.weight"), get(prefix + ".net.0.proj.bias"))
            put(prefix + ".net.0.proj.weight", wp)
            put(prefix + ".net.0.proj.bias", bp)
            lin(prefix + ".net.2")

        def transformer(prefix, temporal):
            norm(prefix + ".norm")
            lin(prefix + ".proj_in")
            b = prefix + ".transformer_blocks.0"
            norm(b + ".norm1"); norm(b + ".norm2"); norm(b + ".norm3")
            attn_self(b + ".attn1", temporal)
            if temporal:
                attn_self(b + ".attn2", True)
            else:
                attn_cross(b + ".attn2")
            ff(b + ".ff")
            lin(prefix + ".proj_out")
            if sd[b + ".ff.net.2.weight"].shape[0] in packing.K8_WIDTHS:
                # K8: the feed-forward sub-block as one kernel, norm3 folded into its first projection (csrc/ff_fused.hip)
                put(b + ".ff.k8", packing.pack_k8(sd[b + ".ff.net.0.proj.weight"], sd[b + ".ff.net.0.proj.bias"], sd[b + ".ff.net.2.weight"],
                                                 sd[b + ".ff.net.2.bias"], sd[b + ".norm3.weight"], sd[b + ".norm3.bias"]))
            if sd[b + ".ff.net.2.weight"].shape[0] in packing.K8_WIDTHS and tuple(sd[prefix + ".proj_out.weight"].shape[:2]) == (sd[b + ".ff.net.2.weight"].shape[0],) * 2:
                # K8's tail: proj_out + the transformer's residual in the feed-forward's kernel (csrc/ff_fused.hip, PO)
                put(prefix + ".proj_out.k8p", packing.pack_k8_proj(packing.pack_conv1x1(sd[prefix + ".proj_out.weight"]), sd[prefix + ".proj_out.bias"]))
            if not temporal and sd[b + ".attn2.to_q.weight"].shape[0] in packing.K5_WIDTHS:
                # K5: the cross-attention sub-block as one kernel, norm2 folded into its query projection (csrc/xattn.hip)
                put(b + ".attn2.k5", packing.pack_k5(packing.pack_conv1x1(sd[b + ".attn2.to_q.weight"]), packing.pack_conv1x1(sd[b + ".attn2.to_out.0.weight"]),
                                                   sd[b + ".norm2.weight"], sd[b + ".norm2.bias"], sd[b + ".attn2.to_out.0.bias"], c.attention_head_dim ** -0.5))
            if temporal and sd[b + ".attn1.to_q.weight"].shape[0] in packing.K7B_WIDTHS:
                # K7, second design: one blob per attention sub-block with its LayerNorm folded in (csrc/tattn2.hip)
                for a, nm in (("attn1", "norm1"), ("attn2", "norm2")):
                    put(f"{b}.{a}.k7b", packing.pack_k7b(
                        *(packing.pack_conv1x1(sd[f"{b}.{a}.to_{n}.weight"]) for n in "qkv"),
                        packing.pack_conv1x1(sd[f"{b}.{a}.to_out.0.weight"]), sd[f"{b}.{nm}.weight"], sd[f"{b}.{nm}.bias"],
                        sd[f"{b}.{a}.to_out.0.bias"], c.attention_head_dim ** -0.5))

        def resnet(prefix):
            norm(prefix + ".norm1"); norm(prefix + ".norm2")
            put(prefix + ".conv1.weight", packing.pack_conv3x3(get(prefix + ".conv1.weight")))
            put(prefix + ".conv1.bias", get(prefix + ".conv1.bias"))
            put(prefix + ".conv2.weight", packing.pack_conv3x3(get(prefix + ".conv2.weight")))
            put(prefix + ".conv2.bias", get(prefix + ".conv2.bias"))
            if prefix + ".conv_shortcut.weight" in sd:
                lin(prefix + ".conv_shortcut")

        def tconv(prefix):
            for i, leaf in ((1, 2), (2, 3), (3, 3), (4, 3)):
                norm(f"{prefix}.conv{i}.0")
                put(f"{prefix}.conv{i}.weight", packing.pack_tconv3(get(f"{prefix}.conv{i}.{leaf}.weight")))
                put(f"{prefix}.conv{i}.bias", get(f"{prefix}.conv{i}.{leaf}.bias"))

        c = self.cfg
        put("conv_in.weight", packing.pack_conv_in(get("conv_in.weight")))
        put("conv_in.bias", get("conv_in.bias"))
        lin("time_embedding.linear_1"); lin("time_embedding.linear_2")
        transformer("transformer_in", True)
        for i, t in enumerate(c.down_block_types):
            p = f"down_blocks.{i}"
            for j in range(c.layers_per_block):
                resnet(f"{p}.resnets.{j}"); tconv(f"{p}.temp_convs.{j}")
                if t.startswith("CrossAttn"):
                    transformer(f"{p}.attentions.{j}", False)
                    transformer(f"{p}.temp_attentions.{j}", True)
            if i != len(c.block_out_channels) - 1:
                put(f"{p}.downsamplers.0.conv.weight", packing.pack_conv3x3(get(f"{p}.downsamplers.0.conv.weight")))
                put(f"{p}.downsamplers.0.conv.bias", get(f"{p}.downsamplers.0.conv.bias"))
        for j in range(2):
            resnet(f"mid_block.resnets.{j}"); tconv(f"mid_block.temp_convs.{j}")
        transformer("mid_block.attentions.0", False)
        transformer("mid_block.temp_attentions.0", True)
        for i, t in enumerate(c.up_block_types):
            p = f"up_blocks.{i}"
            for j in range(c.layers_per_block + 1):
                resnet(f"{p}.resnets.{j}"); tconv(f"{p}.temp_convs.{j}")
                if t.startswith("CrossAttn"):
                    transformer(f"{p}.attentions.{j}", False)
                    transformer(f"{p}.temp_attentions.{j}", True)
            if i != len(c.block_out_channels) - 1:
                put(f"{p}.upsamplers.0.conv.weight", packing.pack_conv3x3(get(f"{p}.upsamplers.0.conv.weight")))
                put(f"{p}.upsamplers.0.conv.bias", get(f"{p}.upsamplers.0.conv.bias"))
        norm("conv_norm_out")
        put("conv_out.weight", packing.pad_rows(packing.pack_conv3x3(get("conv_out.weight")), 64))
        put("conv_out.bias", packing.pad_rows(get("conv_out.bias"), 64))

        # all 22 time_emb_proj layers as ONE [sum(Cout)][temb] projection (one launch per forward)
        ws, bs, off = [], [], 0
        self._temb_slices = {}
        for name in self._resnet_names():
            w = packing.pack_conv1x1(get(name + ".time_emb_proj.weight"))
            ws.append(w); bs.append(get(name + ".time_emb_proj.bias"))
            self._temb_slices[name] = (off, w.shape[0])
            off += w.shape[0]
        put("time_emb_proj_all.weight", torch.cat(ws, 0))
        put("time_emb_proj_all.bias", torch.cat(bs, 0))

        missing = set(sd.keys()) - used
        if missing:
            raise VdxError(f"unexpected keys in state dict: {sorted(missing)[:5]} ... ({len(missing)})")
        self.W = W
        self._device = dev
        self._text_ref, self._text_kv = None, {}
        return self

    def _apply(self, fn, recurse=True):
        # nn.Module.to()/cuda()/half(): move the packed store (floating tensors stay fp16)
        out = super()._apply(fn, recurse)
        if isinstance(self.W, dict) and self.W:
            probe = fn(torch.empty(0, dtype=torch.float16, device=self._device))
            self.W = {k: v.to(probe.device) for k, v in self.W.items()}
            self._device = probe.device
        return out

    def num_parameters(self) -> int:
        return sum(self.W[k].numel() for k in self.W.keys())

    # ------------------------------------------------------------------------------------------
    # parameter sharding (reference: FSDP wrap, fsdp_chunked_coherent.py:63-88)
    # ------------------------------------------------------------------------------------------
    def unit_schedule(self) -> List[str]:
        """Shard units in the order `forward` uses them (one gather per unit per step)."""
        c = self.cfg
        units = ["transformer_in"]

        def layer(p, j, attn):
            units.append(f"{p}.resnets.{j}")
            units.append(f"{p}.temp_convs.{j}")
            if attn:
                units.append(f"{p}.attentions.{j}")
                units.append(f"{p}.temp_attentions.{j}")

        nlev = len(c.block_out_channels)
        for i, t in enumerate(c.down_block_types):
            for j in range(c.layers_per_block):
                layer(f"down_blocks.{i}", j, t.startswith("CrossAttn"))
            if i != nlev - 1:
                units.append(f"down_blocks.{i}.downsamplers.0")
        units += ["mid_block.resnets.0", "mid_block.temp_convs.0", "mid_block.attentions.0",
                  "mid_block.temp_attentions.0", "mid_block.resnets.1", "mid_block.temp_convs.1"]
        for i, t in enumerate(c.up_block_types):
            for j in range(c.layers_per_block + 1):
                layer(f"up_blocks.{i}", j, t.startswith("CrossAttn"))
            if i != nlev - 1:
                units.append(f"up_blocks.{i}.upsamplers.0")
        return units

    @staticmethod
    def unit_of(name: str) -> Optional[str]:
        """Packed-tensor name -> shard unit (None = small stem tensor kept replicated)."""
        parts = name.split(".")
        if parts[0] == "transformer_in":
            return "transformer_in"
        if parts[0] in ("down_blocks", "up_blocks"):
            return ".".join(parts[:4])          # <block>.<i>.<kind>.<j>
        if parts[0] == "mid_block":
            return ".".join(parts[:3])
        return None                              # conv_in/out, time embedding, conv_norm_out

    def shard_(self, rank: int, world: int, group=None, comm=None, transport=None, merge_bytes: int = 64 << 20,
               prefetch_depth: int = 2):
        """Keep 1/world of every unit on this GPU; gather per unit with prefetch (vdx/shard.py).  `comm`: gather through
        the C-ABI RCCL entry point instead of torch.distributed (vdx/comm.py); `transport`: "collective" (RCCL all-gather on the
        side stream: the default) or "peer" (mapped shard arenas, copy-engine pulls: opt-in, vdx/shard.py)."""
        from .shard import ShardedStore
        if not isinstance(self.W, dict):
            raise VdxError("weights are already sharded")
        self.W = ShardedStore(self.W, self.unit_of, self.unit_schedule(), rank, world, group, comm, transport, merge_bytes,
                              prefetch_depth)
        self.ff_block_bytes = 128 << 20
        if self._device.type == "cuda":
            torch.cuda.empty_cache()
        return self

    # ------------------------------------------------------------------------------------------
    # building blocks (all on row matrices)
    # ------------------------------------------------------------------------------------------
    def _resnet(self, p, x, x2, temb_all, n_img, F, hh, ww, part=0, ksplit_ok=True):
        """x may be a one- or two-element LIST [x] / [x, skip] the caller has handed over (its only references): the
        inputs are then released as soon as the shortcut and the first GroupNorm have consumed them — in the up path
        the block input AND the skip tensor, 2 x 189 MB at level 0 of a 16-frame chunk, before the convolutions
        allocate.  (Tensors passed as call arguments stay referenced by the caller's frame until the call returns.)"""
        if isinstance(x, list):
            owned = x
            x, x2 = owned[0], (owned[1] if len(owned) > 1 else None)
            owned.clear()
        W, g, eps = self.W, self.cfg.norm_num_groups, self.cfg.norm_eps
        M, S = n_img * hh * ww, hh * ww
        off, cout = self._temb_slices[p]
        if p + ".conv_shortcut.weight" in W:
            sc = ops.gemm(x, W[p + ".conv_shortcut.weight"], M=M, a2=x2, bias=W[p + ".conv_shortcut.bias"])
        else:
            if x2 is not None:
                raise VdxError(f"{p}: concat input needs a conv_shortcut")
            sc = x
        geo = (n_img, hh, ww, hh, ww, 1, False)
        B = n_img // F
        concat_in = x2 is not None
        c1_, c2_ = x.shape[1], (x2.shape[1] if x2 is not None else 0)
        k1 = (lambda cin1, cin2: self.fuse_conv_gn and (ops.conv3x3_gn_supported(cin1, cin2, cout) if self.fuse_conv_gn == "always"
                                                        else ops.conv3x3_gn_preferred(cin1, cin2, cout, n_img, hh, ww)))
        if k1(c1_, c2_):
            # K1: GroupNorm apply + SiLU inside the convolution (statistics pass + one kernel): the normalised tensor — for
            # the up blocks the [rows][C1 + C2] concat, the widest tensor of the block — is never written, so the memory-lean
            # pieces below have nothing to do either
            h = ops.conv3x3_gn(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"], W[p + ".conv1.weight"], x2=x2, bias=W[p + ".conv1.bias"],
                               bias2=temb_all[:, off:off + cout], rows_per_bias2=F * S, groups=g, n_img=n_img, h=hh, wd=ww, eps=eps,
                               partition_samples=part)
            del x, x2
            return ops.conv3x3_gn(h, W[p + ".norm2.weight"], W[p + ".norm2.bias"], W[p + ".conv2.weight"], bias=W[p + ".conv2.bias"],
                                  residual=sc, groups=g, n_img=n_img, h=hh, wd=ww, eps=eps, partition_samples=part)
        pieces = 0
        if self.ff_block_bytes and self.lean_concat and x2 is not None and M * cout * 2 > (64 << 20):
            # memory-lean mode, concat input (up path): the normalised concat [rows][C1 + C2] is the widest tensor of
            # the block; GroupNorm (4-D: statistics per image) -> conv1 runs over the images in pieces of whole frames
            # of ONE batch item (the time-embedding row of a piece is then row 0 of `temb_all[b:]`), sized to about
            # at most 160 MB (8 frames of the 960-channel level-0 concat: pieces of 73 728 rows still fill the chip),
            # so only a piece of it exists at a time.  Same kernels per image, same bits.
            cin = x.shape[1] + x2.shape[1]
            per_item = next((k for k in range(1, F + 1) if F % k == 0 and (F // k) * S * cin * 2 <= (160 << 20)), F)
            pieces = B * per_item
        if pieces > 1:
            h = torch.empty((M, cout), dtype=torch.float16, device=x.device)
            pn = n_img // pieces                    # images (frames) per piece
            pm = pn * S
            for i in range(pieces):
                r0 = i * pm
                n1 = ops.groupnorm(x[r0:r0 + pm], W[p + ".norm1.weight"], W[p + ".norm1.bias"], groups=g, n_samples=pn,
                                   rows_per_sample=S, eps=eps, silu_act=True, x2=x2[r0:r0 + pm], partition_samples=part or n_img)
                ops.gemm(n1, W[p + ".conv1.weight"], M=pm, mode=ops.CONV3X3, bias=W[p + ".conv1.bias"],
                         bias2=temb_all[r0 // (F * S):, off:off + cout], rows_per_bias2=F * S,
                         conv=(pn, hh, ww, hh, ww, 1, False), out=h[r0:r0 + pm])
                del n1
            del x, x2
        else:
            h = ops.groupnorm(x, W[p + ".norm1.weight"], W[p + ".norm1.bias"], groups=g, n_samples=n_img,
                              rows_per_sample=S, eps=eps, silu_act=True, x2=x2, partition_samples=part)
            del x, x2
            # (split-K tail allowed unless this is a product the memory-lean order cuts into pieces: a piece and the
            # whole must keep the same summation order, the sharded forward has the bits of the resident one)
            h = ops.gemm(h, W[p + ".conv1.weight"], M=M, mode=ops.CONV3X3, bias=W[p + ".conv1.bias"],
                         bias2=temb_all[:, off:off + cout], rows_per_bias2=F * S, conv=geo,
                         allow_ksplit=ksplit_ok and not (concat_in and M * cout * 2 > (64 << 20)))
        h = ops.groupnorm(h, W[p + ".norm2.weight"], W[p + ".norm2.bias"], groups=g, n_samples=n_img,
                          rows_per_sample=S, eps=eps, silu_act=True, partition_samples=part)
        return ops.gemm(h, W[p + ".conv2.weight"], M=M, mode=ops.CONV3X3, bias=W[p + ".conv2.bias"],
                        residual=sc, conv=geo, allow_ksplit=ksplit_ok)

    def _temp_conv(self, p, x, B, F, S, part=0, ksplit_ok=True):
        W, g = self.W, self.cfg.norm_num_groups
        M = B * F * S
        y = x
        for i in (1, 2, 3, 4):
            wi = W[f"{p}.conv{i}.weight"]
            if self.fuse_tconv and (ops.tconv_gn_supported(y.shape[1], wi.shape[0], F) if self.fuse_tconv == "always"
                                    else ops.tconv_gn_preferred(y.shape[1], wi.shape[0], B, F, S)):
                # K3: the GroupNorm apply + SiLU happen inside the convolution (statistics pass + one kernel; the
                # normalised tensor is never written)
                y = ops.tconv_gn(y, W[f"{p}.conv{i}.0.weight"], W[f"{p}.conv{i}.0.bias"], wi, bias=W[f"{p}.conv{i}.bias"],
                                 residual=x if i == 4 else None, groups=g, B=B, F=F, S=S, eps=1e-5, partition_samples=part)
                continue
            n = ops.groupnorm(y, W[f"{p}.conv{i}.0.weight"], W[f"{p}.conv{i}.0.bias"], groups=g, n_samples=B,
                              rows_per_sample=F * S, eps=1e-5, silu_act=True, partition_samples=part)
            y = ops.gemm(n, W[f"{p}.conv{i}.weight"], M=M, mode=ops.TCONV3, bias=W[f"{p}.conv{i}.bias"],
                         tconv=(F, S), residual=x if i == 4 else None, allow_ksplit=ksplit_ok)
        return y

    # Temporaries are released as soon as their consumer has been enqueued (`del`): kernels run in stream order,
    # so the caching allocator can hand the block to the next op.  Left to Python scoping, every intermediate of
    # a transformer block lived until the block returned: 5.6 GB of activations at the 24-frame peak instead of 3.
    def _ff(self, b, t, M, ksplit_ok=True):
        W = self.W
        inner = t.shape[1]
        if self.fuse_ff and b + ".ff.k8" in W and ops.ff_block_supported(inner):
            # K8 (level 0): LayerNorm -> GEGLU -> output projection + residual in one kernel; the [rows][4*inner]
            # intermediate (the largest tensor of the forward) never exists, so the memory-lean row blocks have nothing to do
            return ops.ff_block(t, W[b + ".ff.k8"], M=M)
        blk = M
        if self.ff_block_bytes:
            # memory-lean mode (set by shard_(): the per-device footprint is what the sharded modes are for): the
            # GEGLU intermediate [rows][4*inner] is the largest tensor of the forward (1.2 GB at 16 frames in
            # transformer_in); rows are independent, so the feed-forward runs over row blocks and only one block's
            # intermediate is alive at a time.  Same kernels, same bits.
            # (only where it matters for the peak: intermediates above 256 MB, i.e. levels 0 / 1 and transformer_in; the
            # blocks are made even — 18 432 rows used to be cut into 16 384 + 2 048)
            if M * 8 * inner > (256 << 20):
                nblk = -(-M * 8 * inner // self.ff_block_bytes)
                blk = max(16384, ((M + nblk - 1) // nblk + 4095) // 4096 * 4096)
        if blk >= M:
            ln = ops.layernorm(t, W[b + ".norm3.weight"], W[b + ".norm3.bias"], M=M)
            gg = ops.gemm(ln, W[b + ".ff.net.0.proj.weight"], M=M, bias=W[b + ".ff.net.0.proj.bias"], geglu=True)
            del ln
            return ops.gemm(gg, W[b + ".ff.net.2.weight"], M=M, bias=W[b + ".ff.net.2.bias"], residual=t,
                            allow_ksplit=ksplit_ok and M * 8 * inner <= (256 << 20))      # (never a shape the lean order cuts into row blocks)
        out = torch.empty_like(t[:M])
        for r0 in range(0, M, blk):
            r1 = min(r0 + blk, M)
            ln = ops.layernorm(t[r0:r1], W[b + ".norm3.weight"], W[b + ".norm3.bias"], M=r1 - r0)
            gg = ops.gemm(ln, W[b + ".ff.net.0.proj.weight"], M=r1 - r0, bias=W[b + ".ff.net.0.proj.bias"], geglu=True)
            del ln
            ops.gemm(gg, W[b + ".ff.net.2.weight"], M=r1 - r0, bias=W[b + ".ff.net.2.bias"], residual=t[r0:r1],
                     out=out[r0:r1])
            del gg
        return out

    def _ff_proj_out(self, b, p, t, x, M, xrows=None, ksplit_ok=True):
        """The end of a transformer: `proj_out(t + ff(norm3(t))) + x`.  x has M rows, or `xrows` = M / 2 (a shared-prefix batch:
        both halves of t pair with the same rows of x).  At level 0 ONE kernel (K8 with proj_out as its tail: the feed-forward's
        output never reaches HBM); otherwise the feed-forward, then the projection GEMM with its residual."""
        W = self.W
        xrows = xrows or M
        if self.fuse_ff and self.fuse_proj_out and b + ".ff.k8" in W and p + ".proj_out.k8p" in W and ops.ff_block_supported(t.shape[1]):
            return ops.ff_block(t, W[b + ".ff.k8"], M=M, proj=(W[p + ".proj_out.k8p"], x, xrows))
        t = self._ff(b, t, M, ksplit_ok=ksplit_ok)
        if xrows == M:
            return ops.gemm(t, W[p + ".proj_out.weight"], M=M, bias=W[p + ".proj_out.bias"], residual=x)
        out = torch.empty((M, W[p + ".proj_out.weight"].shape[0]), dtype=torch.float16, device=t.device)
        for i in range(M // xrows):     # two launches over half the rows each, the residual rows shared: no copy
            ops.gemm(t[i * xrows:(i + 1) * xrows], W[p + ".proj_out.weight"], M=xrows, bias=W[p + ".proj_out.bias"], residual=x, out=out[i * xrows:(i + 1) * xrows])
        return out

    def _norm_proj_in(self, p, x, n_samples, rows_per_sample, M, part=0):
        """`norm` -> `proj_in` at the entry of a transformer: GroupNorm (eps 1e-6, no activation) then a Linear.  Where the
        Linear runs on the weights-stationary kernels (levels 0 / 1, transformer_in) the norm is FOLDED into it — per-sample
        weights and an fp32 bias from the statistics, the GEMM reads the raw rows (ops.groupnorm_linear): the normalised
        tensor, one read and one write of the activation, never exists."""
        W, g = self.W, self.cfg.norm_num_groups
        w = W[p + ".proj_in.weight"]
        # (`part`: the samples of the batch these rows stand for — the CFG-shared prefix normalises ONE item with the slab
        # partition, and takes the fold, of the batch of two: its bits must not depend on how many items are computed)
        m_sel = M * (part // n_samples) if part else M
        if self.fold_norm_proj_in and m_sel >= 16384 and M % 64 == 0 and ops.groupnorm_linear_supported(x.shape[1], w.shape[0], rows_per_sample):
            return ops.groupnorm_linear(x, W[p + ".norm.weight"], W[p + ".norm.bias"], w, W[p + ".proj_in.bias"], groups=g,
                                        n_samples=n_samples, rows_per_sample=rows_per_sample, eps=1e-6, partition_samples=part)
        n = ops.groupnorm(x, W[p + ".norm.weight"], W[p + ".norm.bias"], groups=g, n_samples=n_samples,
                          rows_per_sample=rows_per_sample, eps=1e-6, silu_act=False, partition_samples=part)
        return ops.gemm(n, w, M=M, bias=W[p + ".proj_in.bias"])

    def _spatial_transformer(self, p, x, ehs_pad, n_img, F, hh, ww, dup=False):
        """`dup` (the first spatial transformer of a CFG batch whose two items are the same tensor, forward): x holds ONE
        item's rows [n_img/2 * S][C].  `norm` -> `proj_in` -> self-attention -> + and the LayerNorm and query projection
        of the cross-attention see no text, so they are computed once; the two items part at the cross-attention's keys /
        values.  Returns the rows of both items."""
        W, g = self.W, self.cfg.norm_num_groups
        S = hh * ww
        M_out = n_img * S
        if dup:
            n_img //= 2
        M = n_img * S
        C = x.shape[1]
        heads = C // 64
        scale = 64 ** -0.5
        b = p + ".transformer_blocks.0"
        t = self._norm_proj_in(p, x, n_img, S, M, part=2 * n_img if dup else 0)
        # --- self-attention: q | k | v from ONE projection; the flash kernel takes V as rows (transposed by its LDS read),
        # so there is no V^T product and no padded copy for token counts that are no multiple of 8 (latent 40x72 ->
        # 5x9 = 45 tokens at the mid block, InferNet/tests/test_pipeline.py:293; 16x16 -> 2x2 = 4, InferNet/neurons/miner.py:491-494)
        ln = ops.layernorm(t, W[b + ".norm1.weight"], W[b + ".norm1.bias"], M=M)
        wqkv = W[b + ".attn1.to_qkv.weight"]
        if self.ff_block_bytes and self.lean_attn and n_img % 2 == 0 and M * C * 2 > (128 << 20):
            # memory-lean mode: images are independent in the self-attention, so the projection and the attention run
            # over the images in two halves and only half of [rows][3C] is alive at a time.
            # (level 0 only: 0.6 ms per step on a 16-frame window for 45 MB of its peak — measured 4.33 against 4.374 GB
            # with the concat pieces on, gpurun_out r3n — which is what keeps the per-device share under 15 %)
            o = torch.empty((M, C), dtype=torch.float16, device=x.device)
            hm, hn = M // 2, n_img // 2
            for r0 in (0, hm):
                qkv = ops.gemm(ln[r0:r0 + hm], wqkv, M=hm)
                ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_seq=hn, sq=S, skv=S, skv_pad=S, heads=heads,
                               seq_per_kv=1, scale=scale, out=o[r0:r0 + hm], v_rows=True)
                del qkv
            del ln
        elif not self.spatial_v_rows and S % 8 == 0 and M % 64 == 0:
            # the round-2 form, kept for same-process A/B timing (tools/step_ab.py): q|k GEMM + V^T by the swapped GEMM
            qk = ops.gemm(ln, wqkv[:2 * C], M=M)
            vt = ops.gemm(wqkv[2 * C:], ln, M=C)
            del ln
            o = ops.flash_attn(qk[:, :C], qk[:, C:], vt, n_seq=n_img, sq=S, skv=S, skv_pad=S, heads=heads, seq_per_kv=1, scale=scale)
            del qk, vt
        else:
            qkv = ops.gemm(ln, wqkv, M=M)
            del ln
            o = ops.flash_attn(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], n_seq=n_img, sq=S, skv=S, skv_pad=S, heads=heads,
                               seq_per_kv=1, scale=scale, v_rows=True)
            del qkv
        t = ops.gemm(o, W[b + ".attn1.to_out.0.weight"], M=M, bias=W[b + ".attn1.to_out.0.bias"], residual=t)
        del o
        # --- cross-attention over the (padded) text tokens; all F frames of a sample share K/V
        nb = ehs_pad.shape[0] // TEXT_PAD
        k5 = self.fuse_cross_attn and b + ".attn2.k5" in W and ops.cross_attn_block_supported(C, self._text_len)
        kv = self._text_kv.get(p)
        if kv is None:
            k = ops.gemm(ehs_pad, W[b + ".attn2.to_k.weight"], M=ehs_pad.shape[0])
            vt = ops.gemm(W[b + ".attn2.to_v.weight"], ehs_pad, M=C)              # [C][nb*TEXT_PAD]
            kv = self._text_kv[p] = (k, vt, packing.pack_k5_kv(k, vt, nb, TEXT_PAD) if k5 else None)
        k, vt, kvb = kv
        if k5 and kvb is None:
            kvb = packing.pack_k5_kv(k, vt, nb, TEXT_PAD)
            self._text_kv[p] = (k, vt, kvb)
        if k5:
            # K5 (level 0): LayerNorm -> q -> softmax(q K^T) V -> to_out + residual in one kernel; the text keys / values were
            # packed once per prompt.  A shared-prefix batch (`dup`): the same rows against each item's text in turn.
            if dup:
                t2 = torch.empty((M_out, C), dtype=torch.float16, device=x.device)
                for i in range(nb):
                    ops.cross_attn_block(t, W[b + ".attn2.k5"], kvb[i:i + 1], kv_len=self._text_len, n_items=1, rows_per_item=M, out=t2[i * M:(i + 1) * M])
            else:
                t2 = ops.cross_attn_block(t, W[b + ".attn2.k5"], kvb, kv_len=self._text_len, n_items=nb, rows_per_item=M // nb)
            return self._ff_proj_out(b, p, t2, x, M_out, xrows=M)
        ln = ops.layernorm(t, W[b + ".norm2.weight"], W[b + ".norm2.bias"], M=M)
        q = ops.gemm(ln, W[b + ".attn2.to_q.weight"], M=M)
        del ln
        if dup:
            # one query tensor, the keys / values of each item in turn; the residual rows (and, at the block's end, the
            # transformer's input rows) are the shared item's for both — two launches over half the rows each, no copy
            t2 = torch.empty((M_out, C), dtype=torch.float16, device=x.device)
            for i in range(nb):
                o = ops.flash_attn(q, k[i * TEXT_PAD:(i + 1) * TEXT_PAD], vt[:, i * TEXT_PAD:(i + 1) * TEXT_PAD], n_seq=n_img, sq=S,
                                   skv=self._text_len, skv_pad=TEXT_PAD, heads=heads, seq_per_kv=n_img, scale=scale)
                ops.gemm(o, W[b + ".attn2.to_out.0.weight"], M=M, bias=W[b + ".attn2.to_out.0.bias"], residual=t, out=t2[i * M:(i + 1) * M])
                del o
            del q
            return self._ff_proj_out(b, p, t2, x, M_out, xrows=M)
        o = ops.flash_attn(q, k, vt, n_seq=n_img, sq=S, skv=self._text_len, skv_pad=TEXT_PAD, heads=heads,
                           seq_per_kv=n_img // nb, scale=scale)
        del q
        t = ops.gemm(o, W[b + ".attn2.to_out.0.weight"], M=M, bias=W[b + ".attn2.to_out.0.bias"], residual=t)
        del o
        return self._ff_proj_out(b, p, t, x, M)

    def _temporal_transformer(self, p, x, B, F, S, heads, part=0, ksplit_ok=True):
        W, g = self.W, self.cfg.norm_num_groups
        M = B * F * S
        scale = 64 ** -0.5
        b = p + ".transformer_blocks.0"
        t = self._norm_proj_in(p, x, B, F * S, M, part=part)
        fused = self.fuse_temporal_attention and f"{b}.attn1.k7_qkv" in W and ops.temporal_attn_block_supported(t.shape[1], F)
        fused2 = self.fuse_temporal_attention and f"{b}.attn1.k7b" in W and ops.temporal_attn_block2_supported(t.shape[1], F)
        for a, nm in (("attn1", "norm1"), ("attn2", "norm2")):
            if fused2:     # K7, second design (inner 320)
                t = ops.temporal_attn_block2(t, W[f"{b}.{a}.k7b"], B=B, F=F, HW=S)
                continue
            if fused:      # K7: LayerNorm -> q|k|v -> F x F attention -> to_out + residual in one kernel
                t = ops.temporal_attn_block(t, W[f"{b}.{nm}.weight"], W[f"{b}.{nm}.bias"], W[f"{b}.{a}.k7_qkv"],
                                            W[f"{b}.{a}.k7_out"], W[f"{b}.{a}.to_out.0.bias"], B=B, F=F, HW=S, scale=scale)
                continue
            ln = ops.layernorm(t, W[f"{b}.{nm}.weight"], W[f"{b}.{nm}.bias"], M=M)
            qkv = ops.gemm(ln, W[f"{b}.{a}.to_qkv.weight"], M=M)
            del ln
            o = ops.temporal_attn(qkv, B=B, F=F, HW=S, heads=heads, scale=scale)
            del qkv
            t = ops.gemm(o, W[f"{b}.{a}.to_out.0.weight"], M=M, bias=W[f"{b}.{a}.to_out.0.bias"], residual=t)
            del o
        return self._ff_proj_out(b, p, t, x, M, ksplit_ok=ksplit_ok)

    # ------------------------------------------------------------------------------------------
    def _time_embedding(self, timestep, B, device):
        """`Timesteps` on the device.  The timestep never has to exist on the host: a device tensor (the
        reference iterates `scheduler.timesteps`, 0-d int64 on the GPU, :132) is converted in place — no `float(t)`
        sync — and a Python number becomes a one-element fill."""
        if torch.is_tensor(timestep):
            t = timestep.to(device=device, dtype=torch.float32).reshape(-1)[:1]
            if t.data_ptr() % 16:
                t = t.clone()
        else:
            t = torch.full((1,), float(timestep), dtype=torch.float32, device=device)
        return ops.timestep_embedding(t.contiguous(), B, self.cfg.block_out_channels[0])

    @torch.no_grad()
    def forward(self, sample, timestep, encoder_hidden_states, **_unused):
        if not self.W:
            raise VdxError("UNet3DConditionModel: no weights loaded (load_diffusers_state_dict)")
        if not sample.is_cuda:
            raise VdxError("UNet3DConditionModel.forward needs GPU tensors: the path has no CPU fallback")
        c, W = self.cfg, self.W
        dev = sample.device
        wdev = self._device
        if wdev.type != dev.type or (wdev.index is not None and dev.index is not None and wdev.index != dev.index):
            raise VdxError(f"weights are on {wdev}, input on {dev}")
        B, Cin, F, H, Wd = sample.shape
        if Cin != c.in_channels:
            raise VdxError(f"sample has {Cin} channels, model expects {c.in_channels}")
        nlev = len(c.block_out_channels)
        # (a latent whose height or width is not divisible by 2^(levels-1) takes diffusers' `upsample_size` path: every
        # upsampler then resizes to the resolution of the skip tensor it will meet)
        if F > 128:
            raise VdxError("temporal attention kernel handles at most 128 frames per chunk")
        cfg_dup = ops.is_cfg_duplicate(sample)
        sample = sample.to(torch.float16).contiguous()
        ehs = encoder_hidden_states.to(device=dev, dtype=torch.float16)
        if ehs.shape[0] != B or ehs.shape[2] != c.cross_attention_dim or ehs.shape[1] > TEXT_PAD:
            raise VdxError(f"encoder_hidden_states shape {tuple(ehs.shape)} does not fit (B={B}, dim={c.cross_attention_dim})")
        self._text_len = ehs.shape[1]
        # The text keys / values of the 16 cross-attentions depend on the prompt only: they are projected once and kept
        # while the caller keeps passing the SAME tensor object, unmodified (identity + version, not the address: a new
        # tensor may reuse a freed address with other contents).  `pipeline.denoise` and bench.py pass one tensor for
        # all steps and hit the cache.  The reference's own loop rebuilds `emb = torch.cat([uncond, cond])` inside
        # every step (fsdp_chunked_coherent.py:138): on the unchanged script each step is a miss — 32 small K / V
        # projections and one padded copy, ~0.2 ms of a 180 ms step — never a stale hit.  Tensors without a version
        # counter (inference mode) are not cached.
        ref = self._text_ref
        try:
            ver = encoder_hidden_states._version
        except Exception:       # inference-mode tensors do not track versions
            ver = None
        if not (ref is not None and ver is not None and ref[0] is encoder_hidden_states and ref[1] == ver
                and ref[2] == (B, dev)):
            ehs_pad = torch.zeros((B * TEXT_PAD, c.cross_attention_dim), dtype=torch.float16, device=dev)
            ehs_pad.view(B, TEXT_PAD, -1)[:, :ehs.shape[1]] = ehs
            self._text_ref = (encoder_hidden_states, ver, (B, dev), ehs_pad)
            self._text_kv = {}
        ehs_pad = self._text_ref[3]
        n_img = B * F

        # time embedding -> all time_emb_proj outputs [B][sum Cout]
        temb = self._time_embedding(timestep, B, dev)
        e = ops.gemm(temb, W["time_embedding.linear_1.weight"], M=B, bias=W["time_embedding.linear_1.bias"])
        e = ops.gemm(ops.silu(e), W["time_embedding.linear_2.weight"], M=B, bias=W["time_embedding.linear_2.bias"])
        temb_all = ops.gemm(ops.silu(e), W["time_emb_proj_all.weight"], M=B, bias=W["time_emb_proj_all.bias"])

        hh, ww = H, Wd
        # The CFG-shared prefix.  `cat([lat]*2)` (+ the same context term, fsdp_chunked_coherent.py:133-137) makes the two
        # items of the batch the SAME tensor, and nothing before the first cross-attention sees the text: conv_in,
        # transformer_in, the first ResnetBlock / TemporalConvLayer and the first spatial transformer up to its
        # cross-attention's keys would compute every row twice.  When the caller's tensor is known to be such a
        # duplicate — it came out of `ops.cfg_input` and has not been written since (a tag, never a guess: comparing
        # the halves would cost a sync) — those blocks run on ONE item (B1 = 1) with the statistics partition of the
        # batch of two, and the rows fan out at the cross-attention.  Same kernels on the same rows in the same order:
        # the output has the bits of the duplicated forward (tests/test_unet_gpu.py::test_cfg_shared_prefix_*).
        # The blocks of the prefix never take the split-K tail in EITHER form (its plan depends on the row count).
        # A batch the caller built some other way (the reference's unchanged `torch.cat([lat]*2)` + ctx term, :133-137) carries
        # no tag.  With `detect_cfg_duplicate` (set by the diffusers shim, i.e. for the unchanged script; off here) the two
        # halves are COMPARED on the device — an exact test, one small kernel and one host sync per forward (~0.1 ms of a
        # 170 ms step) — and the prefix is shared when, and only when, they are equal.
        if not cfg_dup and self.detect_cfg_duplicate and self.share_cfg_prefix and B == 2:
            cfg_dup = bool(torch.equal(sample[0], sample[1]))
        elif cfg_dup and B == 2 and (self.detect_cfg_duplicate or os.environ.get("VDX_VERIFY_CFG_DUP") == "1"):
            # The tag rests on torch's version counter; a write that does not bump it (`x.data`, another library's raw-pointer
            # kernel, a vdx op handed the tensor as `out=`) leaves a STALE tag, and the prefix would then compute one item for
            # a batch whose items differ — wrong output, no error.  Under VDX_VERIFY_CFG_DUP=1 (tests/conftest.py sets it)
            # and in detect mode (which pays the sync anyway) the claim is checked: a stale tag raises.
            if not torch.equal(sample[0], sample[1]):
                raise ops.VdxError("UNet3DConditionModel.forward: the batch is tagged as a CFG duplicate (ops.cfg_input) but its two items "
                                   "differ: it was written after cfg_input by something that does not bump torch's version counter")
        dup = bool(self.share_cfg_prefix and B == 2 and cfg_dup and c.down_block_types[0].startswith("CrossAttn"))
        self.last_forward_shared_prefix = dup
        B1 = 1 if dup else B
        part = B if dup else 0
        x = ops.conv_in(sample[:B1], W["conv_in.weight"], W["conv_in.bias"])
        x = self._temporal_transformer("transformer_in", x, B1, F, hh * ww, c.transformer_in_heads, part=part, ksplit_ok=False)
        skips = [(x, hh, ww)]
        shared_skip = dup                       # skips[0] holds one item's rows; doubled where it is consumed (the last up ResNet)
        for i, t in enumerate(c.down_block_types):
            p = f"down_blocks.{i}"
            for j in range(c.layers_per_block):
                first = i == 0 and j == 0
                if first:
                    x = self._resnet(f"{p}.resnets.{j}", x, None, temb_all, B1 * F, F, hh, ww, part=part * F, ksplit_ok=False)
                    x = self._temp_conv(f"{p}.temp_convs.{j}", x, B1, F, hh * ww, part=part, ksplit_ok=False)
                else:
                    x = self._resnet(f"{p}.resnets.{j}", x, None, temb_all, n_img, F, hh, ww)
                    x = self._temp_conv(f"{p}.temp_convs.{j}", x, B, F, hh * ww)
                if t.startswith("CrossAttn"):
                    x = self._spatial_transformer(f"{p}.attentions.{j}", x, ehs_pad, n_img, F, hh, ww, dup=dup and first)
                    x = self._temporal_transformer(f"{p}.temp_attentions.{j}", x, B, F, hh * ww, x.shape[1] // 64)
                skips.append((x, hh, ww))
            if i != nlev - 1:
                ho, wo = (hh - 1) // 2 + 1, (ww - 1) // 2 + 1
                x = ops.gemm(x, W[f"{p}.downsamplers.0.conv.weight"], M=n_img * ho * wo, mode=ops.CONV3X3,
                             bias=W[f"{p}.downsamplers.0.conv.bias"], conv=(n_img, hh, ww, ho, wo, 2, False), allow_ksplit=True)
                hh, ww = ho, wo
                skips.append((x, hh, ww))
        # mid
        x = self._resnet("mid_block.resnets.0", x, None, temb_all, n_img, F, hh, ww)
        x = self._temp_conv("mid_block.temp_convs.0", x, B, F, hh * ww)
        x = self._spatial_transformer("mid_block.attentions.0", x, ehs_pad, n_img, F, hh, ww)
        x = self._temporal_transformer("mid_block.temp_attentions.0", x, B, F, hh * ww, x.shape[1] // 64)
        x = self._resnet("mid_block.resnets.1", x, None, temb_all, n_img, F, hh, ww)
        x = self._temp_conv("mid_block.temp_convs.1", x, B, F, hh * ww)
        # up
        for i, t in enumerate(c.up_block_types):
            p = f"up_blocks.{i}"
            for j in range(c.layers_per_block + 1):
                skip, sh, sw = skips.pop()
                if (sh, sw) != (hh, ww):
                    raise VdxError("skip connection resolution mismatch")
                if shared_skip and not skips:
                    skip = torch.cat([skip, skip])          # transformer_in's output stood for both items (one 2 x 141 MB copy at 24 frames)
                owned = [x, skip]
                x = skip = None
                x = self._resnet(f"{p}.resnets.{j}", owned, None, temb_all, n_img, F, hh, ww)
                x = self._temp_conv(f"{p}.temp_convs.{j}", x, B, F, hh * ww)
                if t.startswith("CrossAttn"):
                    x = self._spatial_transformer(f"{p}.attentions.{j}", x, ehs_pad, n_img, F, hh, ww)
                    x = self._temporal_transformer(f"{p}.temp_attentions.{j}", x, B, F, hh * ww, x.shape[1] // 64)
            if i != nlev - 1:
                th, tw = skips[-1][1], skips[-1][2]          # resolution of the next block's skip tensors
                if (th, tw) == (2 * hh, 2 * ww):
                    mode_up = 1                               # nearest x2, folded into the conv's gather
                elif hh <= th <= 2 * hh and ww <= tw <= 2 * ww:
                    mode_up = 2                               # F.interpolate(size=skip resolution, mode="nearest"), folded likewise
                else:
                    raise VdxError(f"upsampler {p}: {hh}x{ww} -> {th}x{tw}")
                x = ops.gemm(x, W[f"{p}.upsamplers.0.conv.weight"], M=n_img * th * tw, mode=ops.CONV3X3,
                             bias=W[f"{p}.upsamplers.0.conv.bias"], conv=(n_img, hh, ww, th, tw, 1, mode_up), allow_ksplit=True)
                hh, ww = th, tw
        # out
        n = ops.groupnorm(x, W["conv_norm_out.weight"], W["conv_norm_out.bias"], groups=c.norm_num_groups,
                          n_samples=n_img, rows_per_sample=hh * ww, eps=c.norm_eps, silu_act=True)
        y = ops.gemm(n, W["conv_out.weight"], M=n_img * hh * ww, mode=ops.CONV3X3, bias=W["conv_out.bias"],
                     conv=(n_img, hh, ww, hh, ww, 1, False))
        out = ops.rows_to_ncfhw(y, B, c.out_channels, F, hh, ww)
        return SimpleNamespace(sample=out)

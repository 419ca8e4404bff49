"""`CLIPTextModel` on libvdx_hip.so — the call the reference makes once per prompt before the denoising loop:

    fsdp_chunked_coherent.py:96-103
        ids = tokenizer([prompt, ""], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
        emb = self.text_encoder(ids.to(device))[0]            # (2, 77, 1024) last hidden state
    (InferNet/neurons/miner.py:553-565 makes the same call.)

The arithmetic is `transformers.CLIPTextModel` (pins: `transformers>=4.30.0`; Zeroscope ships the OpenCLIP ViT-H
text tower as packaged for Stable-Diffusion 2.x: hidden 1024, 23 layers, 16 heads of 64, MLP 4096 with exact
GELU, 77 positions, causal mask, final LayerNorm).  `transformers` is installed in this
image, so — unlike the diffusers operators — this module is checked against the real dependency
(tests/test_clip_gpu.py builds `transformers.CLIPTextModel` from a config with seeded weights on the CPU).

Same surface: `text_encoder(input_ids)[0]`, `.config`, `load_transformers_state_dict(model.state_dict())` (keys with
or without the `text_model.` prefix older transformers releases use).  Rows are [B*128][D]: each sequence's 77
tokens padded to 128 rows (two key tiles of the attention kernel); LayerNorm / GEMM / `vdx_flash_attn_f16` with the
causal flag / `vdx_gelu_f16` do the work.  Token + position embedding lookup is a gather on the device (index
plumbing, no arithmetic beyond one add).  The value bias is folded behind the output projection (softmax rows sum
to 1), so V^T comes from the swapped GEMM as in the UNet.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import ops
from ._lib import VdxError

SEQ_PAD = 128      # rows per sequence (77 tokens + zero rows)


@dataclass
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 23
    num_attention_heads: int = 16
    max_position_embeddings: int = 77
    layer_norm_eps: float = 1e-5

    @staticmethod
    def sd2() -> "CLIPTextConfig":
        return CLIPTextConfig()


class TextEncoderOutput(tuple):
    """`out[0]` / `out.last_hidden_state`, like transformers' ModelOutput for the one field the reference reads."""

    @property
    def last_hidden_state(self):
        return self[0]


class CLIPTextModel(nn.Module):
    def __init__(self, cfg: Optional[CLIPTextConfig] = None):
        super().__init__()
        self.cfg = cfg or CLIPTextConfig()
        self.config = SimpleNamespace(**vars(self.cfg))
        c = self.cfg
        if c.hidden_size != 64 * c.num_attention_heads:
            raise VdxError("CLIPTextModel: the attention kernel needs 64-wide heads (hidden = 64 * heads)")
        if c.hidden_size % 64 or c.intermediate_size % 64 or c.max_position_embeddings > SEQ_PAD:
            raise VdxError("CLIPTextModel: widths must be multiples of 64 and positions <= 128")
        self.W: Dict[str, torch.Tensor] = {}
        self._device = torch.device("cpu")

    @torch.no_grad()
    def load_transformers_state_dict(self, sd: Dict[str, torch.Tensor], device=None):
        dev = torch.device(device) if device is not None else self._device
        c = self.cfg
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}
        W: Dict[str, torch.Tensor] = {}
        used = set()

        def get(k):
            used.add(k)
            if k not in sd:
                raise VdxError(f"missing key in state dict: {k}")
            return sd[k].to(dev)

        def put(name, t):
            W[name] = t.to(device=dev, dtype=torch.float16).contiguous()

        put("tok", get("embeddings.token_embedding.weight"))
        put("pos", get("embeddings.position_embedding.weight"))
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}"
            for n in ("layer_norm1", "layer_norm2"):
                put(f"{p}.{n}.weight", get(f"{p}.{n}.weight"))
                put(f"{p}.{n}.bias", get(f"{p}.{n}.bias"))
            a = p + ".self_attn"
            put(a + ".qk.weight", torch.cat([get(a + ".q_proj.weight"), get(a + ".k_proj.weight")], 0))
            put(a + ".qk.bias", torch.cat([get(a + ".q_proj.bias"), get(a + ".k_proj.bias")], 0))
            put(a + ".v.weight", get(a + ".v_proj.weight"))                 # swapped GEMM -> V^T; its bias moves:
            wo = get(a + ".out_proj.weight")
            put(a + ".out.weight", wo)
            put(a + ".out.bias", get(a + ".out_proj.bias").float() + wo.float() @ get(a + ".v_proj.bias").float())
            for n in ("fc1", "fc2"):
                put(f"{p}.mlp.{n}.weight", get(f"{p}.mlp.{n}.weight"))
                put(f"{p}.mlp.{n}.bias", get(f"{p}.mlp.{n}.bias"))
        put("final_layer_norm.weight", get("final_layer_norm.weight"))
        put("final_layer_norm.bias", get("final_layer_norm.bias"))
        extra = {k for k in sd if k not in used and not k.endswith("position_ids")}
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

    @torch.no_grad()
    def forward(self, input_ids, attention_mask=None, **_unused):
        c, W = self.cfg, self.W
        if not W:
            raise VdxError("CLIPTextModel: no weights loaded")
        if attention_mask is not None:
            raise VdxError("CLIPTextModel: the reference passes no attention_mask (causal mask only)")
        if input_ids.dim() != 2 or input_ids.shape[1] > c.max_position_embeddings:
            raise VdxError(f"CLIPTextModel: input_ids must be (B, <= {c.max_position_embeddings})")
        dev = W["tok"].device
        if dev.type != "cuda":
            raise VdxError("CLIPTextModel: weights are not on a GPU (the encoder has no CPU fallback)")
        B, S = input_ids.shape
        D, H = c.hidden_size, c.num_attention_heads
        M = B * SEQ_PAD
        ids = input_ids.to(dev)
        x = torch.zeros((B, SEQ_PAD, D), dtype=torch.float16, device=dev)
        x[:, :S] = W["tok"][ids] + W["pos"][:S]                              # CLIPTextEmbeddings
        x = x.view(M, D)
        for i in range(c.num_hidden_layers):
            p = f"encoder.layers.{i}"
            a = p + ".self_attn"
            ln = ops.layernorm(x, W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], M=M, eps=c.layer_norm_eps)
            qk = ops.gemm(ln, W[a + ".qk.weight"], M=M, bias=W[a + ".qk.bias"])
            vt = ops.gemm(W[a + ".v.weight"], ln, M=D)                       # V^T [D][B*128]
            o = ops.flash_attn(qk[:, :D], qk[:, D:], vt, n_seq=B, sq=SEQ_PAD, skv=S, skv_pad=SEQ_PAD, heads=H,
                               seq_per_kv=1, scale=64 ** -0.5, causal=True)
            x = ops.gemm(o, W[a + ".out.weight"], M=M, bias=W[a + ".out.bias"], residual=x)
            ln = ops.layernorm(x, W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], M=M, eps=c.layer_norm_eps)
            hid = ops.gemm(ln, W[p + ".mlp.fc1.weight"], M=M, bias=W[p + ".mlp.fc1.bias"])
            ops.gelu(hid, out=hid)
            x = ops.gemm(hid, W[p + ".mlp.fc2.weight"], M=M, bias=W[p + ".mlp.fc2.bias"], residual=x)
        y = ops.layernorm(x, W["final_layer_norm.weight"], W["final_layer_norm.bias"], M=M, eps=c.layer_norm_eps)
        return TextEncoderOutput((y.view(B, SEQ_PAD, D)[:, :S].contiguous(),))

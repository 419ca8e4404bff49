"""Stand-in for the part of `diffusers` the reference's strategy scripts touch (`fsdp_chunked_coherent.py:22,55-61,
90-103,106,132-142,223`; `fsdp.py`, `chunk_only.py`, `fsdp_chunked.py` use the same surface):

    pipe = DiffusionPipeline.from_pretrained(model_id, torch_dtype=torch.float16, low_cpu_mem_usage=True,
                                             use_safetensors=False, device_map=None)
    pipe.unet / .text_encoder / .vae / .tokenizer / .scheduler        (assignable attributes)

The components are this package's HIP-backed modules (same call surfaces, same state-dict keys as diffusers /
transformers).  Weights:
  * `model_id` names a LOCAL directory in diffusers layout (unet/, vae/, text_encoder/, tokenizer/, scheduler/) ->
    the checkpoint is ingested (`.safetensors`, or `.bin` through `torch.load(weights_only=True)`) and the real CLIP
    tokenizer files are used;
  * otherwise (no network exists here: a hub id cannot be fetched) -> seeded SYNTHETIC weights of the Zeroscope
    architecture and a hash tokenizer, with a warning: the run then measures the system, it does not make a video
    anybody wants to watch.

Like the reference (`:55-57`) the pipeline is built on the CPU.  A module without `nn.Parameter`s is invisible to
`FSDP(...)`'s device placement, so the modules place themselves: weights move to the GPU at the first forward that
brings GPU tensors, and — when a process group with more than one rank exists and the script did not call
`.to(device)` itself (the reference's non-FSDP branch, `:79-81`) — the UNet's weights are sharded per unit over the
ranks (vdx/shard.py), which is this stack's counterpart of the FSDP wrap the script believes it applied.
"""
from __future__ import annotations

import json
import os
import warnings
import zlib
from types import SimpleNamespace

import torch
import torch.distributed as dist

from .. import clip_text as _clip
from .. import scheduler as _sched
from .. import unet3d as _unet
from .. import vae as _vae
from .. import weights as _weights

__version__ = "0.0-vdx-shim"


def _lazy_place(module, tensor, shard=False):
    if module._device.type == "cuda" or not tensor.is_cuda:
        return
    module.to(tensor.device)
    if shard and not getattr(module, "_placed_by_to", False) and dist.is_available() and dist.is_initialized() \
            and dist.get_world_size() > 1 and isinstance(module.W, dict):
        module.shard_(dist.get_rank(), dist.get_world_size())


class UNet3DConditionModel(_unet.UNet3DConditionModel):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        # the unchanged script builds its CFG batch with torch.cat (fsdp_chunked_coherent.py:133): no tag from ops.cfg_input, so
        # let the forward verify on the device that the two items are the same tensor and share the text-independent blocks
        self.detect_cfg_duplicate = True

    def to(self, *a, **k):
        self._placed_by_to = True           # the script placed the module itself: replicate, do not shard
        return super().to(*a, **k)

    def forward(self, sample, timestep, encoder_hidden_states, **kw):
        placed = getattr(self, "_placed_by_to", False)
        _lazy_place(self, sample, shard=True)
        self._placed_by_to = placed
        return super().forward(sample, timestep, encoder_hidden_states, **kw)


class CLIPTextModel(_clip.CLIPTextModel):
    def forward(self, input_ids, attention_mask=None, **kw):
        _lazy_place(self, input_ids)
        return super().forward(input_ids, attention_mask, **kw)


class AutoencoderKL(_vae.AutoencoderKL):
    def decode(self, z, return_dict=True):
        _lazy_place(self, z)
        return super().decode(z, return_dict)


DDIMScheduler = _sched.DDIMScheduler


class HashTokenizer:
    """Deterministic stand-in for CLIP's BPE tokenizer when no tokenizer files exist offline: BOS, one id per
    whitespace-separated word (crc32 into the vocabulary), EOS padding — the shapes and special ids of the real one
    (`model_max_length` 77, BOS 49406, EOS/pad 49407), not its segmentation."""
    model_max_length = 77
    bos_token_id, eos_token_id = 49406, 49407

    def __call__(self, text, padding="max_length", max_length=None, truncation=True, return_tensors="pt", **_kw):
        texts = [text] if isinstance(text, str) else list(text)
        n = max_length or self.model_max_length
        rows = []
        for t in texts:
            ids = [self.bos_token_id] + [1000 + zlib.crc32(w.lower().encode()) % 48000 for w in t.split()]
            ids = ids[:n - 1] + [self.eos_token_id]
            rows.append(ids + [self.eos_token_id] * (n - len(ids)))
        ids = torch.tensor(rows, dtype=torch.int64)
        return SimpleNamespace(input_ids=ids, attention_mask=(ids != self.eos_token_id).long())


def _load_file(path_no_ext_candidates):
    for p in path_no_ext_candidates:
        if os.path.exists(p):
            if p.endswith(".safetensors"):
                from safetensors.torch import load_file
                return load_file(p)
            return torch.load(p, map_location="cpu", weights_only=True)
    return None


def _read_json(path):
    if not os.path.exists(path):
        return None
    with open(path) as f:
        return json.load(f)


def _unet_config(d):
    """`unet/config.json` (diffusers `UNet3DConditionModel.config`) -> UNet3DConfig; the Zeroscope values where the file
    or a key is absent.  Refuses what the kernels are not built for instead of loading it wrong."""
    j = _read_json(f"{d}/unet/config.json")
    if j is None:
        return _unet.UNet3DConfig.zeroscope()
    hd = j.get("attention_head_dim", 64)
    hd = set(hd) if isinstance(hd, (list, tuple)) else {hd}
    if hd != {64}:
        raise ValueError(f"unet/config.json: attention_head_dim {sorted(hd)} — the attention kernels are built for 64")
    if j.get("num_attention_heads") not in (None, 64) or j.get("act_fn", "silu") != "silu":
        raise ValueError("unet/config.json: only the ModelScope / Zeroscope UNet3D variant (silu, head dim 64) is implemented")
    z = _unet.UNet3DConfig.zeroscope()
    return _unet.UNet3DConfig(
        in_channels=j.get("in_channels", z.in_channels), out_channels=j.get("out_channels", z.out_channels),
        block_out_channels=tuple(j.get("block_out_channels", z.block_out_channels)),
        layers_per_block=j.get("layers_per_block", z.layers_per_block), attention_head_dim=64,
        cross_attention_dim=j.get("cross_attention_dim", z.cross_attention_dim),
        norm_num_groups=j.get("norm_num_groups", z.norm_num_groups), norm_eps=j.get("norm_eps", z.norm_eps),
        transformer_in_heads=z.transformer_in_heads,        # (fixed at 8 in diffusers' UNet3DConditionModel)
        down_block_types=tuple(j.get("down_block_types", z.down_block_types)),
        up_block_types=tuple(j.get("up_block_types", z.up_block_types)))


def _vae_config(d):
    j = _read_json(f"{d}/vae/config.json")
    z = _vae.VaeConfig.sd()
    if j is None:
        return z
    return _vae.VaeConfig(latent_channels=j.get("latent_channels", z.latent_channels), out_channels=j.get("out_channels", z.out_channels),
                          block_out_channels=tuple(j.get("block_out_channels", z.block_out_channels)),
                          layers_per_block=j.get("layers_per_block", z.layers_per_block),
                          norm_num_groups=j.get("norm_num_groups", z.norm_num_groups),
                          scaling_factor=j.get("scaling_factor", z.scaling_factor))


def _clip_config(d):
    j = _read_json(f"{d}/text_encoder/config.json")
    z = _clip.CLIPTextConfig.sd2()
    if j is None:
        return z
    if j.get("hidden_act", "gelu") != "gelu":
        raise ValueError(f"text_encoder/config.json: hidden_act {j.get('hidden_act')!r} — only the OpenCLIP tower of SD-2.x (gelu) is implemented")
    return _clip.CLIPTextConfig(**{k: j.get(k, getattr(z, k)) for k in (
        "vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads", "max_position_embeddings",
        "layer_norm_eps")})


class DiffusionPipeline:
    def __init__(self, unet, text_encoder, vae, tokenizer, scheduler, synthetic):
        self.unet, self.text_encoder, self.vae = unet, text_encoder, vae
        self.tokenizer, self.scheduler = tokenizer, scheduler
        self.synthetic_weights = synthetic

    @classmethod
    def from_pretrained(cls, model_id, torch_dtype=None, low_cpu_mem_usage=True, use_safetensors=None, device_map=None,
                        **_unused):
        if torch_dtype not in (None, torch.float16):
            raise ValueError("the HIP path computes in fp16 (the reference loads with torch_dtype=torch.float16)")
        if str(model_id) == "synthetic:tiny":       # test hook: the same topology at narrow widths (seconds to build)
            unet = UNet3DConditionModel(_unet.UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128,
                                                           transformer_in_heads=2))
            text = CLIPTextModel(_clip.CLIPTextConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=2,
                                                      num_attention_heads=2))
            vae = AutoencoderKL(_vae.VaeConfig(block_out_channels=(64, 64, 128, 128)))
        elif os.path.isdir(str(model_id)):      # a checkpoint in diffusers layout: its own config.json files give the widths
            d = str(model_id)
            unet = UNet3DConditionModel(_unet_config(d))
            text = CLIPTextModel(_clip_config(d))
            vae = AutoencoderKL(_vae_config(d))
        else:
            unet = UNet3DConditionModel(_unet.UNet3DConfig.zeroscope())
            text = CLIPTextModel(_clip.CLIPTextConfig.sd2())
            vae = AutoencoderKL(_vae.VaeConfig.sd())
        sched_kw, tokenizer, synthetic = {}, None, True
        if os.path.isdir(str(model_id)):
            d = str(model_id)
            usd = _load_file([f"{d}/unet/diffusion_pytorch_model.safetensors", f"{d}/unet/diffusion_pytorch_model.bin"])
            tsd = _load_file([f"{d}/text_encoder/model.safetensors", f"{d}/text_encoder/pytorch_model.bin"])
            vsd = _load_file([f"{d}/vae/diffusion_pytorch_model.safetensors", f"{d}/vae/diffusion_pytorch_model.bin"])
            if usd is None or tsd is None or vsd is None:
                raise FileNotFoundError(f"{d}: expected unet/, text_encoder/ and vae/ weights in diffusers layout")
            unet.load_diffusers_state_dict(usd, device="cpu")
            text.load_transformers_state_dict(tsd, device="cpu")
            vae.load_diffusers_state_dict(vsd, device="cpu")
            cfgp = f"{d}/scheduler/scheduler_config.json"
            if os.path.exists(cfgp):
                known = ("num_train_timesteps", "beta_start", "beta_end", "beta_schedule", "steps_offset",
                         "set_alpha_to_one", "clip_sample", "prediction_type", "timestep_spacing")
                sched_kw = {k: v for k, v in json.load(open(cfgp)).items() if k in known}
            if os.path.isdir(f"{d}/tokenizer"):
                import transformers
                tokenizer = transformers.CLIPTokenizer.from_pretrained(f"{d}/tokenizer")
            synthetic = False
        else:
            if str(model_id) != "synthetic:tiny":
                warnings.warn(f"diffusers shim: {model_id!r} is not a local directory and nothing can be downloaded here — "
                              "using seeded SYNTHETIC weights of the Zeroscope architecture and a hash tokenizer")
            unet.load_diffusers_state_dict(_weights.synthetic_state_dict(unet.cfg, 1234, "cpu"), device="cpu")
            text.load_transformers_state_dict(_weights.synthetic_clip_state_dict(text.cfg, 11, "cpu"), device="cpu")
            vae.load_diffusers_state_dict(_weights.synthetic_vae_state_dict(vae.cfg, 7, "cpu"), device="cpu")
        return cls(unet, text, vae, tokenizer or HashTokenizer(), DDIMScheduler(**sched_kw), synthetic)

    def to(self, device):
        for m in (self.unet, self.text_encoder, self.vae):
            m.to(device)
        return self

    def decode_latents(self, latents):
        """`fsdp.py:172` relies on this pipeline helper of older diffusers releases: latents (1,C,F,h,w) -> video
        (1,3,F,H,W) float32 in [-1, 1] scale of the decoder (the caller maps to uint8)."""
        z = latents / self.vae.config.scaling_factor
        b, c, f, h, w = z.shape
        img = self.vae.decode(z.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)).sample
        return img.reshape(b, f, *img.shape[1:]).permute(0, 2, 1, 3, 4).float()

"""Test infrastructure: writes a checkpoint directory in diffusers layout (unet/, vae/, text_encoder/, scheduler/ with
their config.json files and .safetensors weights) at NARROW widths, the way a user's local copy of
`cerspense/zeroscope_v2_XL` is laid out (`fsdp_chunked_coherent.py:55-61` loads such a directory).  The weights are the
seeded synthetic tables of the oracle / vdx.weights; some 1x1 projections are stored as Conv2d weights (C,C,1,1), as
older diffusers checkpoints do."""
import json
import os

import torch
from safetensors.torch import save_file

UNET_CH, CROSS = (64, 128, 128, 128), 128
VAE_CH = (64, 64, 128, 128)
CLIP = dict(vocab_size=49408, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
            max_position_embeddings=77, layer_norm_eps=1e-5)
SCHED = dict(_class_name="DDIMScheduler", num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
             beta_schedule="scaled_linear", clip_sample=False, set_alpha_to_one=False, steps_offset=1,
             prediction_type="epsilon", timestep_spacing="leading", skip_prk_steps=True)


def unet_state_dict():
    from oracle.unet3d_ref import UNet3DConfig as RefCfg, synthetic_state_dict
    return synthetic_state_dict(RefCfg.tiny(ch=UNET_CH, cross=CROSS, in_heads=8), seed=4321)      # transformer_in: 8 heads, as diffusers builds it


def write(root, conv_proj=True):
    import vdx  # noqa: F401
    from vdx import weights
    from vdx.clip_text import CLIPTextConfig
    from vdx.vae import VaeConfig
    for sub in ("unet", "vae", "text_encoder", "scheduler"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    usd = {k: v.half().contiguous() for k, v in unet_state_dict().items()}
    if conv_proj:        # spatial transformers' proj_in / proj_out as 1x1 convolutions
        for k in list(usd):
            if ".attentions." in k and k.endswith(("proj_in.weight", "proj_out.weight")):
                usd[k] = usd[k][:, :, None, None].contiguous()
    save_file(usd, os.path.join(root, "unet", "diffusion_pytorch_model.safetensors"))
    json.dump(dict(_class_name="UNet3DConditionModel", in_channels=4, out_channels=4, block_out_channels=list(UNET_CH),
                   layers_per_block=2, attention_head_dim=64, cross_attention_dim=CROSS, norm_num_groups=32, norm_eps=1e-5,
                   act_fn="silu", sample_size=32,
                   down_block_types=["CrossAttnDownBlock3D"] * 3 + ["DownBlock3D"],
                   up_block_types=["UpBlock3D"] + ["CrossAttnUpBlock3D"] * 3),
              open(os.path.join(root, "unet", "config.json"), "w"))
    vsd = {k: v.half().contiguous() for k, v in weights.synthetic_vae_state_dict(VaeConfig(block_out_channels=VAE_CH), 7, "cpu").items()}
    save_file(vsd, os.path.join(root, "vae", "diffusion_pytorch_model.safetensors"))
    json.dump(dict(_class_name="AutoencoderKL", latent_channels=4, out_channels=3, block_out_channels=list(VAE_CH),
                   layers_per_block=2, norm_num_groups=32, scaling_factor=0.18215),
              open(os.path.join(root, "vae", "config.json"), "w"))
    tsd = {"text_model." + k: v.half().contiguous() for k, v in weights.synthetic_clip_state_dict(CLIPTextConfig(**CLIP), 11, "cpu").items()}
    save_file(tsd, os.path.join(root, "text_encoder", "model.safetensors"))
    json.dump(dict(CLIP, hidden_act="gelu", model_type="clip_text_model"), open(os.path.join(root, "text_encoder", "config.json"), "w"))
    json.dump(SCHED, open(os.path.join(root, "scheduler", "scheduler_config.json"), "w"))
    return usd

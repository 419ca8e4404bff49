"""CPU suite: the AutoencoderKL-decode oracle (structural pins — the diffusers boundary is parity-unpinned,
oracle/__init__.py) and the product's weight ingest for it."""
import numpy as np
import pytest
import torch

import vdx  # noqa: F401
from vdx._lib import VdxError
from vdx.vae import AutoencoderKL, VaeConfig

from oracle import vae_ref


def test_vae_oracle_structure_matches_published_decoder():
    with torch.device("meta"):
        m = vae_ref.AutoencoderKLRef(vae_ref.VaeConfig.sd())
    # Stable-Diffusion AutoencoderKL: 83 653 863 parameters = encoder 34 163 592 + quant_conv 72
    # + post_quant_conv 20 + decoder 49 490 179
    assert sum(p.numel() for p in m.decoder.parameters()) == 49_490_179
    assert sum(p.numel() for p in m.post_quant_conv.parameters()) == 20
    sd = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert len(sd) == 140
    assert sd["post_quant_conv.weight"] == (4, 4, 1, 1)
    assert sd["decoder.conv_in.weight"] == (512, 4, 3, 3)
    assert sd["decoder.mid_block.attentions.0.to_q.weight"] == (512, 512)
    assert sd["decoder.mid_block.attentions.0.group_norm.weight"] == (512,)
    assert sd["decoder.up_blocks.2.resnets.0.conv_shortcut.weight"] == (256, 512, 1, 1)
    assert sd["decoder.up_blocks.3.resnets.0.conv_shortcut.weight"] == (128, 256, 1, 1)
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in sd
    assert sd["decoder.up_blocks.2.upsamplers.0.conv.weight"] == (256, 256, 3, 3)
    assert sd["decoder.conv_out.weight"] == (3, 128, 3, 3)


def test_vae_oracle_decodes_8x_and_maps_frames_like_the_reference():
    cfg = vae_ref.VaeConfig.tiny()
    m = vae_ref.AutoencoderKLRef(cfg).eval()
    m.load_state_dict(vae_ref.synthetic_state_dict(cfg))
    lat = torch.randn(1, 4, 2, 4, 8, generator=torch.Generator().manual_seed(3))
    frames = vae_ref.frames_from_latents(m, lat)
    assert len(frames) == 2 and frames[0].shape == (32, 64, 3) and frames[0].dtype == np.uint8
    with torch.no_grad():
        x = m.decode(lat[:, :, 1] / 0.18215).sample[0].permute(1, 2, 0)
    want = ((x * 0.5 + 0.5).clamp(0, 1) * 255).byte().numpy()      # .byte() truncates
    assert np.array_equal(frames[1], want)
    assert frames[1].min() >= 0 and frames[1].max() <= 255 and frames[1].std() > 1


def test_vae_weight_ingest_covers_every_decoder_key():
    with torch.device("meta"):
        ref = vae_ref.AutoencoderKLRef(vae_ref.VaeConfig.sd())
    sd = dict(ref.state_dict())
    m = AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd, device="meta")
    # packed = decoder + post_quant_conv folded into conv_in, conv_in K 45 -> 64, conv_out rows 3 -> 64, no value bias
    assert m.num_parameters() == 49_490_179 - 512 * 36 + 512 * 64 - 512 + 61 * (9 * 128 + 1)
    assert m.config.scaling_factor == 0.18215 and m.config.latent_channels == 4
    # a full AutoencoderKL checkpoint also carries the encoder half: accepted and dropped
    sd["encoder.conv_in.weight"] = torch.empty(128, 3, 3, 3, device="meta")
    sd["quant_conv.weight"] = torch.empty(8, 8, 1, 1, device="meta")
    AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd, device="meta")
    sd["decoder.bogus.weight"] = torch.empty(1, device="meta")
    with pytest.raises(VdxError):
        AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd, device="meta")


def test_vae_decode_has_no_cpu_path():
    cfg = vae_ref.VaeConfig.tiny()
    m = AutoencoderKL(VaeConfig(block_out_channels=cfg.block_out_channels))
    m.load_diffusers_state_dict({k: v.half() for k, v in vae_ref.synthetic_state_dict(cfg).items()})
    with pytest.raises(VdxError):
        m.decode(torch.zeros(1, 4, 8, 8, dtype=torch.float16))

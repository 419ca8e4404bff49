"""-m gpu: AutoencoderKL decode (fsdp_chunked_coherent.py:219-225) on libvdx_hip.so against
the CPU oracle (oracle/vae_ref.py) — committed golden output, live oracle on another shape, the reference's uint8
frame mapping bit-exact, and size-independent properties at the full Stable-Diffusion widths."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def tiny_vae(gpu):
    import vdx  # noqa: F401
    from vdx.vae import AutoencoderKL, VaeConfig
    from oracle import vae_ref
    cfg = vae_ref.VaeConfig.tiny()
    sd16 = {k: v.half() for k, v in vae_ref.synthetic_state_dict(cfg, seed=4321).items()}
    m = AutoencoderKL(VaeConfig(block_out_channels=cfg.block_out_channels)).load_diffusers_state_dict(sd16, device=gpu)
    return m, sd16, cfg


def test_softmax_rows_matches_torch(gpu):
    import vdx  # noqa: F401
    from vdx import ops
    g = torch.Generator().manual_seed(5)
    for rows, cols, scale in ((7, 64, 0.5), (130, 9216, 1 / 512 ** 0.5), (3, 2056, 1.0)):
        x = (torch.randn(rows, cols + 8, generator=g) * 6).half()
        ref = torch.softmax(x[:, :cols].float() * scale, dim=-1)
        d = x.to(gpu)
        ops.softmax_rows(d, rows=rows, cols=cols, scale=scale)
        out = d.cpu()
        assert torch.equal(out[:, cols:], x[:, cols:])                       # columns past `cols` untouched
        assert (out[:, :cols].float() - ref).abs().max() <= 1e-3 * ref.max() + 2e-7
        assert (out[:, :cols].float().sum(1) - 1).abs().max() < 2e-3


def test_frame_mapping_is_bit_exact_with_the_reference_ops(gpu):
    import vdx  # noqa: F401
    from vdx import ops
    g = torch.Generator().manual_seed(9)
    n, H, W = 2, 5, 7
    rows = (torch.randn(n * H * W, 64, generator=g) * 1.5).half()
    rows[0, :3] = torch.tensor([1.0, -1.0, 0.9961])                        # edges: 255, 0, just below 255
    rows[1, :3] = torch.tensor([5.0, -7.0, float("0.003")])
    got = ops.rows_to_u8_frames(rows.to(gpu), n, H, W).cpu()
    sample = rows[:, :3].reshape(n, H, W, 3)                                # == img_lat[i].permute(1,2,0)
    want = ((sample * 0.5 + 0.5).clamp(0, 1) * 255).byte()                  # fp16 arithmetic, as the reference runs it
    assert torch.equal(got, want)


def test_vae_tiny_matches_golden(gpu):
    gold = np.load(os.path.join(GOLD, "vae_tiny.npz"))
    m, _, _ = tiny_vae(gpu)
    z = torch.randn(3, 4, 8, 16, generator=torch.Generator().manual_seed(11)).half()
    out = m.decode(z.to(gpu)).sample
    assert out.shape == (3, 3, 64, 128) and out.dtype == torch.float16
    err, floor = rel_l2(out.float().cpu(), torch.from_numpy(gold["out"]).float()), float(gold["floor"])
    print(f"vae_tiny: rel-L2 {err:.3e} (fp16-CPU floor {floor:.3e})")
    assert err <= 2 * floor + 3e-4          # + the fp16 rounding of the stored vector


def test_vae_tiny_live_oracle_other_shape_and_frames(gpu):
    """Oracle run live on another shape (one frame, 16x8 latent); the uint8 frames agree with the oracle's to one
    grey level almost everywhere (the oracle maps an fp32 sample, the product its own fp16 sample)."""
    from oracle import vae_ref
    m, sd16, cfg = tiny_vae(gpu)
    ref = vae_ref.AutoencoderKLRef(cfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd16.items()})
    lat = torch.randn(1, 4, 2, 16, 8, generator=torch.Generator().manual_seed(23)) * 0.18215 * 0.8
    want = vae_ref.frames_from_latents(ref, lat)
    z = (lat[0].permute(1, 0, 2, 3) / 0.18215).half()
    with torch.no_grad():
        o32 = ref.decode(z.float()).sample
    out = m.decode(z.to(gpu)).sample
    err = rel_l2(out.float().cpu(), o32)
    print(f"vae live: rel-L2 {err:.3e}")
    assert err <= 4e-3
    got = m.decode_frames_u8(z.to(gpu)).cpu().numpy()
    assert got.shape == (2, 128, 64, 3)
    diff = np.abs(got.astype(np.int32) - np.stack(want).astype(np.int32))
    assert diff.max() <= 3 and (diff <= 1).mean() > 0.995
    # and the product's frames are exactly the reference mapping of the product's own sample
    mine = ((out.permute(0, 2, 3, 1) * 0.5 + 0.5).clamp(0, 1) * 255).byte().cpu().numpy()
    assert np.array_equal(got, mine)


def test_vae_full_width_matches_oracle_live(gpu):
    """Stable-Diffusion widths (49.5 M seeded parameters) on two frames of a 16x32 latent against the fp32 oracle."""
    import vdx  # noqa: F401
    from vdx.vae import AutoencoderKL, VaeConfig
    from oracle import vae_ref
    cfg = vae_ref.VaeConfig.sd()
    sd16 = {k: v.half() for k, v in vae_ref.synthetic_state_dict(cfg, seed=2).items()}
    ref = vae_ref.AutoencoderKLRef(cfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd16.items()})
    m = AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd16, device=gpu)
    z = torch.randn(2, 4, 16, 32, generator=torch.Generator().manual_seed(6)).half()
    with torch.no_grad():
        want = ref.decode(z.float()).sample
    got = m.decode(z.to(gpu)).sample
    err = rel_l2(got.float().cpu(), want)
    print(f"vae SD widths, 2 x 16x32: rel-L2 {err:.3e}, out std {float(want.std()):.3f}")
    assert got.shape == (2, 3, 128, 256) and err <= 4e-3


def test_vae_full_width_properties(gpu):
    """Stable-Diffusion widths (512/512/256/128) at a 24x32 latent: finite, deterministic, frames independent of
    their batch neighbours (to rounding), pipeline.decode_frames == decode of the same batch."""
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser
    from vdx.vae import AutoencoderKL, VaeConfig
    from oracle import vae_ref
    sd = {k: v.half() for k, v in vae_ref.synthetic_state_dict(vae_ref.VaeConfig.sd(), seed=2).items()}
    m = AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd, device=gpu)
    z = torch.randn(3, 4, 24, 32, generator=torch.Generator().manual_seed(4)).half().to(gpu)
    a = m.decode(z).sample
    assert a.shape == (3, 3, 192, 256) and torch.isfinite(a).all() and float(a.float().std()) > 0.05
    assert torch.equal(a, m.decode(z).sample)
    # frames do not mix, and the GroupNorm statistics are reduced with a fixed row-slab partition: a frame decoded
    # alone (how the reference decodes, fsdp_chunked_coherent.py:219-225) has the BITS of the same frame in a batch
    assert torch.equal(a[1:2], m.decode(z[1:2]).sample)
    assert torch.equal(a[:2], m.decode(z[:2]).sample)
    frames = DistributedVideoDiffuser.decode_frames(
        type("P", (), {"cfg": DiffuserConfig(device=gpu)})(), (z.permute(1, 0, 2, 3)[None].float() * 0.18215), m, batch=2)
    assert len(frames) == 3 and frames[0].shape == (192, 256, 3)
    one = m.decode_frames_u8(((z[2:3].float() * 0.18215) / 0.18215).half()).cpu().numpy()[0]
    assert np.array_equal(frames[2], one)

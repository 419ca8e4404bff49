"""-m gpu: the HIP path against fp32 references at the FULL spatial extent of the headline configuration
(BASELINE cfg2: Zeroscope-XL widths, 576x1024 video = 72x128 latent, `fsdp_chunked_coherent.py:140` at `--height 576
--width 1024`, :288-289).  VERDICT r5 weak point 1: the XL-width goldens stop at a 32x64 latent and the per-op flash test
at 4 096 keys, so the 9 216-key flash tiles, K1's border / slab logic on 9 216-row images, K3's image staging at 9 216
pixels and the GroupNorm slab partition at 9 216 rows per frame were only ever checked against themselves.

 (a) the whole UNet, CFG batch 2 on (2, 4, F, 72, 128) for F = 2 and 3 (F = 3: temporal padding, an odd frame count in
     every 5-D GroupNorm), LIVE fp32 oracle on the box's host cores (~15-25 s per forward), both the shared-prefix and the
     duplicated forward: rel-L2 <= 4e-3 as `test_unet_full_width_matches_golden`;
 (b) per-op at the real shapes against plain fp32 torch: flash attention at 9 216 keys x 5 heads (uniform and peaked
     scores), K1 on 72x128 images (320 -> 320 and the 640 + 320 concat form), K3 at 9 216 pixels, GroupNorm at
     9 216 x F rows per sample.
Tolerances are the per-op ones of tests/test_ops_gpu.py (|err| <= tol * max|ref| + tol * |ref|)."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

H, W = 72, 128
S = H * W


def h(x):
    return x.half().float()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def close(out, ref, tol):
    out = out.float().cpu()
    assert out.shape == ref.shape, (out.shape, ref.shape)
    assert torch.isfinite(out).all()
    scale = ref.abs().max().item() + 1e-6
    err = (out - ref).abs()
    bad = err > tol * scale + tol * ref.abs()
    assert not bad.any(), f"max err {err.max().item():.4g} (scale {scale:.4g}), {int(bad.sum())} / {bad.numel()} bad"


def _ops():
    import vdx  # noqa: F401
    from vdx import ops, packing
    return ops, packing


# ---------------------------------------------------------------------------------------------------------------
# (a) the whole UNet at 72x128
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def xl_pair(gpu):
    """(HIP UNet, fp32 oracle UNet) on the SAME seeded table (the golden generator's, rounded to fp16 — what the HIP
    model stores), built once for the module: 1.41 B parameters each."""
    import vdx  # noqa: F401
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig as RefCfg, synthetic_state_dict
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    sd = synthetic_state_dict(RefCfg.zeroscope(), seed=1234, dtype=torch.float16)
    m = UNet3DConditionModel(UNet3DConfig.zeroscope()).load_diffusers_state_dict(sd, device=gpu)
    with torch.device("meta"):
        ref = UNet3DConditionModelRef(RefCfg.zeroscope())
    ref = ref.to_empty(device="cpu").eval()
    ref.load_state_dict({k: v.float() for k, v in sd.items()})
    del sd
    yield m, ref
    del m, ref
    torch.cuda.empty_cache()


@pytest.mark.parametrize("Fr", [2, 3])
def test_unet_full_extent_matches_live_oracle(gpu, xl_pair, Fr):
    """The headline configuration's spatial extent through the whole network against the fp32 oracle, live."""
    from vdx import ops
    m, ref = xl_pair
    g = torch.Generator().manual_seed(500 + Fr)
    lat = torch.randn(1, 4, Fr, H, W, generator=g).half()
    ehs = torch.randn(2, 77, 1024, generator=g).half()
    t = 981
    with torch.no_grad():
        want = ref(torch.cat([lat, lat]).float(), torch.tensor(t), ehs.float()).sample
    x = ops.cfg_input(lat.to(gpu), None, 0.0)
    shared = m(x, t, encoder_hidden_states=ehs.to(gpu)).sample
    assert m.last_forward_shared_prefix
    dup = m(x.clone(), t, encoder_hidden_states=ehs.to(gpu)).sample
    assert not m.last_forward_shared_prefix
    e_s, e_d = rel_l2(shared.float().cpu(), want), rel_l2(dup.float().cpu(), want)
    print(f"unet XL widths, {Fr}f@{H}x{W}, CFG batch 2: rel-L2 shared-prefix {e_s:.3e}, duplicated {e_d:.3e}; "
          f"oracle out std {float(want.std()):.3f}")
    assert shared.shape == want.shape and torch.isfinite(shared.float()).all()
    assert e_s <= 4e-3 and e_d <= 4e-3
    assert torch.equal(shared, dup)
    # the two CFG items really differ (the text does): the comparison is not of two copies of one thing
    assert rel_l2(want[0], want[1]) > 1e-2


# ---------------------------------------------------------------------------------------------------------------
# (b) per-op at the real shapes
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("peak", [1.0, 4.0], ids=["uniform", "peaked"])
def test_flash_attention_at_9216_keys(gpu, peak):
    """The dominant kernel at its level-0 shape: 9 216 queries x 9 216 keys, 5 heads, two sequences, V as rows of the
    q|k|v projection.  `peaked`: q and k x4 each (scores x16, row maxima ~ +60 nat) so the lazy softmax offset has to move."""
    ops, _ = _ops()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    n_seq, heads = 2, 5
    inner = heads * 64
    g = torch.Generator().manual_seed(9216 + int(peak))
    qkv = torch.randn(n_seq * S, 3 * inner, generator=g)
    qkv[:, :2 * inner] *= peak
    qkv = h(qkv)
    q, k, v = (qkv[:, i * inner:(i + 1) * inner].reshape(n_seq, S, heads, 64).transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(n_seq * S, inner)
    d = qkv.half().to(gpu)
    out = ops.flash_attn(d[:, :inner], d[:, inner:2 * inner], d[:, 2 * inner:], n_seq=n_seq, sq=S, skv=S, skv_pad=S,
                         heads=heads, seq_per_kv=1, scale=0.125, v_rows=True)
    close(out, ref, tol=4e-3)
    assert torch.equal(out, ops.flash_attn(d[:, :inner], d[:, inner:2 * inner], d[:, 2 * inner:], n_seq=n_seq, sq=S, skv=S,
                                           skv_pad=S, heads=heads, seq_per_kv=1, scale=0.125, v_rows=True))


@pytest.mark.parametrize("c1,c2,temb,resid", [(320, 0, True, False), (320, 0, False, True), (640, 320, True, False)],
                         ids=["conv1-320", "conv2-320-res", "conv1-concat-960"])
def test_conv3x3_gn_on_72x128_images(gpu, c1, c2, temb, resid):
    """K1 on the headline image: 12 patch rows x 4 patch columns per image, every border case of the 8 x 34 staged patch,
    two images, the up blocks' 640 + 320 concat; against fp32 conv(SiLU(GroupNorm(cat))) and the un-fused kernels."""
    ops, packing = _ops()
    n, cout = 2, 320
    C_ = c1 + c2
    g = torch.Generator().manual_seed(C_ + int(temb))
    x = h(torch.randn(n, C_, H, W, generator=g) * 1.5 + 0.4 * torch.randn(1, C_, 1, 1, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C_, generator=g)), h(0.3 * torch.randn(C_, generator=g))
    w = h(torch.randn(cout, C_, 3, 3, generator=g) / math.sqrt(9 * C_))
    b = h(torch.randn(cout, generator=g) * 0.1)
    te = h(torch.randn(1, cout, generator=g) * 0.3) if temb else None       # one time-embedding row for both frames
    M = n * S
    res = h(torch.randn(M, cout, generator=g)) if resid else None
    act = h(F.silu(h(F.group_norm(x, 32, gamma, beta, 1e-5))))
    ref4 = F.conv2d(act, w, b, padding=1)
    if temb:
        ref4 = ref4 + te[:, :, None, None]
    ref = packing.nchw_to_rows(ref4)
    if resid:
        ref = ref + res
    dv = lambda t: None if t is None else t.half().to(gpu)   # noqa: E731
    rows = packing.nchw_to_rows(x).half().to(gpu)
    xa = rows[:, :c1].contiguous()
    xb = rows[:, c1:].contiguous() if c2 else None
    wp = packing.pack_conv3x3(w.half()).to(gpu)
    assert ops.conv3x3_gn_supported(c1, c2, cout)
    kw = dict(x2=xb, bias=dv(b), bias2=dv(te), rows_per_bias2=n * S, residual=dv(res), groups=32, n_img=n, h=H, wd=W, eps=1e-5)
    out = ops.conv3x3_gn(xa, dv(gamma), dv(beta), wp, **kw)
    close(out, ref, tol=4e-3)
    nrm = ops.groupnorm(xa, dv(gamma), dv(beta), groups=32, n_samples=n, rows_per_sample=S, eps=1e-5, silu_act=True, x2=xb)
    unf = ops.gemm(nrm, wp, M=M, mode=ops.CONV3X3, bias=dv(b), bias2=dv(te), rows_per_bias2=n * S, residual=dv(res),
                   conv=(n, H, W, H, W, 1, False))
    close(unf, ref, tol=4e-3)
    close(out, unf.float().cpu(), tol=2e-3)
    assert torch.equal(out, ops.conv3x3_gn(xa, dv(gamma), dv(beta), wp, **kw))


@pytest.mark.parametrize("Fr", [3, 12])
def test_tconv_gn_at_9216_pixels(gpu, Fr):
    """One link of the TemporalConvLayer chain on level-0 rows of the headline latent: 9 216 pixels per frame (576 pixel
    blocks of 16).  F = 12 takes K3 (image staged and normalised in LDS); F = 3 has no K3 tile (F must be a multiple of
    8 / 12 / 16) and takes GroupNorm apply + the temporal-conv GEMM, as the F = 3 UNet test above does."""
    ops, packing = _ops()
    B, C, Co = 1, 320, 320
    g = torch.Generator().manual_seed(40 + Fr)
    x5 = h(torch.randn(B, C, Fr, S, 1, generator=g) * 1.5 + 0.3 * torch.randn(1, C, 1, 1, 1, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    w = h(torch.randn(Co, C, 3, 1, 1, generator=g) / math.sqrt(3 * C))
    b = h(torch.randn(Co, generator=g) * 0.1)
    M = B * Fr * S
    res = h(torch.randn(M, Co, generator=g))
    a = h(F.silu(h(F.group_norm(x5, 32, gamma, beta, 1e-5))))
    y = F.conv3d(a, w, b, padding=(1, 0, 0))
    ref = y[..., 0].permute(0, 2, 3, 1).reshape(M, Co) + res
    rows = x5[..., 0].permute(0, 2, 3, 1).reshape(M, C).contiguous().half().to(gpu)
    dv = lambda t: t.half().to(gpu)   # noqa: E731
    wp = packing.pack_tconv3(w).half().to(gpu)
    n = ops.groupnorm(rows, dv(gamma), dv(beta), groups=32, n_samples=B, rows_per_sample=Fr * S, eps=1e-5, silu_act=True)
    unfused = ops.gemm(n, wp, M=M, mode=ops.TCONV3, bias=dv(b), residual=dv(res), tconv=(Fr, S))
    close(unfused, ref, tol=4e-3)
    assert ops.tconv_gn_supported(C, Co, Fr) == (Fr == 12)
    if Fr == 12:
        out = ops.tconv_gn(rows, dv(gamma), dv(beta), wp, bias=dv(b), residual=dv(res), groups=32, B=B, F=Fr, S=S, eps=1e-5)
        close(out, ref, tol=4e-3)
        close(out, unfused.float().cpu(), tol=2e-3)


@pytest.mark.parametrize("ns,Fr,c1,c2", [(2, 24, 320, 0), (2, 3, 320, 0), (1, 12, 640, 320), (48, 1, 320, 0)],
                         ids=["5d-24f", "5d-3f", "5d-12f-concat", "4d-48-images"])
def test_groupnorm_at_9216_rows_per_frame(gpu, ns, Fr, c1, c2):
    """GroupNorm (+SiLU) with the headline's slab partitions: 5-D samples of 9 216 x F rows (221 184 at 24 frames: 1 728
    slabs of 128 rows per sample), the concat form, and the 4-D form (one 9 216-row sample per image); against fp64."""
    ops, _ = _ops()
    C = c1 + c2
    rps = S * Fr
    g = torch.Generator().manual_seed(ns + Fr + C)
    x = h(torch.randn(ns * rps, C, generator=g) * 2 + torch.randn(1, C, generator=g))
    gamma, beta = h(1 + 0.2 * torch.randn(C, generator=g)), h(0.3 * torch.randn(C, generator=g))
    xd = x.half().to(gpu)
    ref = torch.empty_like(x)
    for s_ in range(ns):                                         # per sample: bounded host memory
        x3 = x[s_ * rps:(s_ + 1) * rps].double().t().unsqueeze(0)
        ref[s_ * rps:(s_ + 1) * rps] = F.silu(F.group_norm(x3, 32, gamma.double(), beta.double(), 1e-5))[0].t().float()
    x1 = xd[:, :c1].contiguous()
    x2 = xd[:, c1:].contiguous() if c2 else None
    out = ops.groupnorm(x1, gamma.half().to(gpu), beta.half().to(gpu), groups=32, n_samples=ns, rows_per_sample=rps,
                        eps=1e-5, silu_act=True, x2=x2)
    close(out, ref, tol=4e-3)


# ---------------------------------------------------------------------------------------------------------------
# (c) the step right after the path at its full extent: one 576x1024 frame through the VAE decoder (:219-225)
# ---------------------------------------------------------------------------------------------------------------
def test_vae_decode_one_full_size_frame_matches_oracle_live(gpu):
    """`vae.decode(z / 0.18215).sample` for ONE frame of the headline video — a 72x128 latent to 576x1024 RGB, Stable-Diffusion
    widths, the mid block's single attention head over 9 216 tokens — against the fp32 oracle on the host cores, and the uint8
    frame the reference would write (:224-225) against the oracle's to one grey level."""
    import numpy as np
    import vdx  # noqa: F401
    from vdx.vae import AutoencoderKL, VaeConfig
    from oracle import vae_ref
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    cfg = vae_ref.VaeConfig.sd()
    sd16 = {k: v.half() for k, v in vae_ref.synthetic_state_dict(cfg, seed=2).items()}
    ref = vae_ref.AutoencoderKLRef(cfg).eval()
    ref.load_state_dict({k: v.float() for k, v in sd16.items()})
    m = AutoencoderKL(VaeConfig.sd()).load_diffusers_state_dict(sd16, device=gpu)
    z = torch.randn(1, 4, H, W, generator=torch.Generator().manual_seed(72)).half()
    with torch.no_grad():
        want = ref.decode(z.float()).sample
    got = m.decode(z.to(gpu)).sample
    err = rel_l2(got.float().cpu(), want)
    print(f"vae SD widths, one 72x128 latent -> 576x1024: rel-L2 {err:.3e}, out std {float(want.std()):.3f}")
    assert got.shape == (1, 3, 8 * H, 8 * W) and err <= 4e-3
    u8 = m.decode_frames_u8(z.to(gpu)).cpu().numpy()[0]
    want_u8 = ((want[0].permute(1, 2, 0) * 0.5 + 0.5).clamp(0, 1) * 255).byte().numpy()
    diff = np.abs(u8.astype(np.int32) - want_u8.astype(np.int32))
    assert u8.shape == (8 * H, 8 * W, 3) and diff.max() <= 3 and (diff <= 1).mean() > 0.995

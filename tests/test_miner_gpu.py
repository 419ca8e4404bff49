"""-m gpu: the InferNet miner's denoising loop (InferNet/neurons/miner.py:516-586) on the
same UNet / DDIM kernels: recorded trajectory against the CPU oracle, and bit-stable fp16 bytes (they are what the
Merkle leaves commit to, miner.py:196-203)."""
import hashlib

import pytest
import torch

pytestmark = pytest.mark.gpu

TINY = dict(ch=(64, 128, 128, 128), cross=128, in_heads=2)


@pytest.mark.parametrize("shape", [(1, 4, 4, 16, 32),
                                   (1, 4, 3, 16, 16)])    # the miner's default: 128x128 px, 3 frames (miner.py:491-494,550)
def test_miner_loop_trace_is_bit_stable_and_matches_oracle(gpu, shape):
    import vdx  # noqa: F401
    from vdx.miner import denoise_with_trace, leaf_hash
    from vdx.scheduler import DDIMScheduler
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from oracle.ddim_ref import DDIMSchedulerRef
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig as RefCfg, synthetic_state_dict
    sd = synthetic_state_dict(RefCfg.tiny(**TINY), seed=1234)
    cfg = UNet3DConfig(block_out_channels=TINY["ch"], cross_attention_dim=TINY["cross"], transformer_in_heads=TINY["in_heads"])
    unet = UNet3DConditionModel(cfg).load_diffusers_state_dict(sd, device=gpu)
    g = torch.Generator().manual_seed(3)
    z0 = torch.randn(*shape, generator=g).half()
    emb = torch.randn(1, 77, TINY["cross"], generator=g).half()
    steps = 3
    runs = [denoise_with_trace(unet, DDIMScheduler(), z0.to(gpu), emb.to(gpu), steps) for _ in range(2)]
    a, b = runs
    assert a["timesteps"] == [667, 334, 1] and len(a["latents"]) == len(a["noise_preds"]) == steps
    assert torch.equal(a["latents"][0].cpu(), z0)
    # same inputs -> same bytes -> same leaves, run after run
    leaves = [[leaf_hash(t, z, e) for t, z, e in zip(r["timesteps"], r["latents"], r["noise_preds"])] for r in runs]
    assert leaves[0] == leaves[1] and torch.equal(a["z"], b["z"])
    t0, zb, eb = a["timesteps"][1], a["latents"][1].cpu().numpy().tobytes(), a["noise_preds"][1].cpu().numpy().tobytes()
    assert leaves[0][1] == hashlib.sha256(t0.to_bytes(2, "big") + zb + eb).digest() and len(zb) == 2 * z0.numel()
    # the chain is the scheduler's: z_{i+1} = step(eps_i, t_i, z_i)
    s = DDIMScheduler(); s.set_timesteps(steps)
    assert torch.equal(s.step(a["noise_preds"][0], a["timesteps"][0], a["latents"][0]).prev_sample, a["latents"][1])
    # oracle: fp32 UNet on fp16 I/O + DDIM reference, same loop
    ref = UNet3DConditionModelRef(RefCfg.tiny(**TINY)).eval()
    ref.load_state_dict({k: v.half().float() for k, v in sd.items()})
    rs = DDIMSchedulerRef(); rs.set_timesteps(steps)
    z = z0.clone()
    with torch.no_grad():
        for i, t in enumerate(rs.timesteps):
            eps = ref(z.float(), t, emb.float()).sample.half()
            if i == 0:
                e0 = float((a["noise_preds"][0].float().cpu() - eps.float()).norm() / eps.float().norm())
            z = rs.step(eps, t, z).prev_sample
    err = float((a["z"].float().cpu() - z.float()).norm() / z.float().norm())
    print(f"miner loop: eps_0 rel-L2 {e0:.3e}, final latent rel-L2 {err:.3e}")
    assert [float(x) for x in a["alphas"]] == [float(rs.alphas_cumprod[t]) for t in a["timesteps"]]
    assert e0 < 4e-3 and err < 1e-2

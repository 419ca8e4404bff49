"""-m gpu: WHOLE DDIM schedules on the HIP path against the fp32 oracle's stored trajectories — the
north star's sentence is about *denoised latents*, and one forward says nothing about how the fp16 error grows over a
schedule.  Tolerance: calibrated on the trajectory itself — the golden files carry, per step, the rel-L2 distance between
the SAME loop run in fp16 on the CPU (tiny widths: the real fp16 model; XL widths: fp16-storage emulation, see
tests/golden/make_golden.py) and the fp32 oracle; the HIP path must stay within 2x that at every recorded step.

  * sched50_tiny.npz   — the reference's default 50 steps (fsdp_chunked_coherent.py:284) on a `hybrid_ctx` job, tiny widths;
  * cfg1_xl_full.npz   — BASELINE cfg1 (8 frames @256x256, chunk_only planner) at Zeroscope-XL widths, all 10 steps."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TINY = dict(ch=(64, 128, 128, 128), cross=128, in_heads=2)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def denoise_logged(d, lat):
    """`DistributedVideoDiffuser.denoise` (pipeline.py) with the latent after every step kept."""
    from vdx import ops
    emb = torch.cat([d.uncond_emb, d.cond_emb], dim=0)
    lat, log = lat.contiguous(), []
    for t in d.scheduler._host_timesteps:
        x = ops.cfg_input(lat, d.ctx, d.cfg.context_weight)
        noise = d.unet(x, t, encoder_hidden_states=emb).sample
        lat = d.scheduler.step_cfg(noise, t, lat, d.cfg.guidance_scale)
        log.append(lat)
    return lat, log


def test_fifty_step_schedule_tiny_hybrid_ctx(gpu):
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser, seeded_noise
    from vdx.scheduler import DDIMScheduler
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from oracle.unet3d_ref import UNet3DConfig as RefCfg, synthetic_state_dict
    g = np.load(os.path.join(GOLD, "sched50_tiny.npz"))
    T, H, W, steps = 12, 16, 16, 50
    sd = synthetic_state_dict(RefCfg.tiny(**TINY), seed=1234)
    unet = UNet3DConditionModel(UNet3DConfig(block_out_channels=TINY["ch"], cross_attention_dim=TINY["cross"],
                                             transformer_in_heads=TINY["in_heads"])).load_diffusers_state_dict(sd, device=gpu)
    emb = torch.randn(2, 77, TINY["cross"], generator=torch.Generator().manual_seed(1)).half().to(gpu)
    cfg = DiffuserConfig(num_frames=T, steps=steps, chunk_size=6, overlap=2, height=H * 8, width=W * 8, mode="hybrid_ctx",
                         device="cuda", noise_device="cpu")
    d = DistributedVideoDiffuser(cfg, unet, DDIMScheduler(), emb[1:], emb[:1])
    cp = d.plan()
    assert [list(r) for r in cp.ranges] == g["ranges"].tolist() and (cp.chunk, cp.overlap) == (int(g["cs"]), int(g["ov"]))
    base = seeded_noise((1, 4, T, H, W), 1.0, gpu, "cpu")
    chunks = []
    for i, (s, e) in enumerate(cp.ranges):
        lat, log = denoise_logged(d, base[:, :, s:e].clone())
        chunks.append((s, e, lat))
        if i == 0:
            for k, step in enumerate(g["snap_steps"].tolist()):
                err, floor = rel_l2(log[step][0].cpu().float(), torch.from_numpy(g["lat_snaps_w0"][k]).float()), float(g["floor_snaps_w0"][k])
                print(f"step {step + 1:2d}: HIP rel-L2 {err:.3e}  fp16-CPU floor {floor:.3e}")
                assert err <= 2.0 * floor, (step, err, floor)
    out = d.blend(chunks, base, cp.overlap).cpu()
    err, floor = rel_l2(out, torch.from_numpy(g["lat"])), float(g["floor_blend"])
    print(f"blended latent after 50 steps: HIP rel-L2 {err:.3e}  fp16-CPU floor {floor:.3e}")
    assert err <= 2.0 * floor


def test_cfg1_full_ten_step_schedule_xl_widths(gpu):
    import importlib.util
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser, seeded_noise
    from vdx.scheduler import DDIMScheduler
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from oracle.unet3d_ref import UNet3DConfig as RefCfg, synthetic_state_dict
    path = os.path.join(GOLD, "cfg1_xl_full.npz")
    g = np.load(path)
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = synthetic_state_dict(RefCfg.zeroscope(), seed=1234, dtype=torch.float16)
    unet = UNet3DConditionModel(UNet3DConfig.zeroscope()).load_diffusers_state_dict(sd, device=gpu)
    del sd
    uncond, cond = mg.cfg1_embeddings()
    T, H, W, steps = 8, 32, 32, 10
    cfg = DiffuserConfig(num_frames=T, steps=steps, chunk_size=0, overlap=4, height=H * 8, width=W * 8, mode="chunk",
                         device="cuda", noise_device="cpu", overlap_rule="third")
    d = DistributedVideoDiffuser(cfg, unet, DDIMScheduler(), uncond.to(gpu), cond.to(gpu))
    cp = d.plan()
    assert [list(r) for r in cp.ranges] == g["ranges"].tolist() == [[0, 8], [6, 8]] and cp.overlap == int(g["ov"]) == 2
    base = seeded_noise((1, 4, T, H, W), 1.0, gpu, "cpu")
    chunks = []
    for i, (s, e) in enumerate(cp.ranges):
        lat, log = denoise_logged(d, base[:, :, s:e].clone())
        chunks.append((s, e, lat))
        if i == 0:
            for k in range(steps):
                err, floor = rel_l2(log[k][0].cpu().float(), torch.from_numpy(g["lat_steps_w0"][k]).float()), float(g["floor_steps_w0"][k])
                print(f"step {k + 1:2d}: HIP rel-L2 {err:.3e}  fp16-emulation floor {floor:.3e}")
                assert err <= max(2.0 * floor, 4e-3), (k, err, floor)
    out = d.blend(chunks, base, cp.overlap).cpu()
    err, floor = rel_l2(out, torch.from_numpy(g["lat"])), float(g["floor_blend"])
    print(f"cfg1 blended latent after 10 steps: HIP rel-L2 {err:.3e}  fp16-emulation floor {floor:.3e}")
    assert err <= max(2.0 * floor, 4e-3)

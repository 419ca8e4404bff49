"""Generates tests/golden/*.npz from the CPU oracle (oracle/), run in the build container.

The reference's own Python cannot be imported here (no diffusers/cv2/pynvml) and
holds no fixtures for this path, so the golden outputs come from the oracle restatement with
seeded synthetic weights ("parity unpinned" at the diffusers boundary — see oracle/__init__.py).
Inputs are regenerated from seeds by the tests; only expected outputs (+ the measured fp16 noise
floor of the same computation on CPU) are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.ddim_ref import DDIMSchedulerRef  # noqa: E402
from oracle.pipeline_ref import run_video  # noqa: E402
from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig, synthetic_state_dict  # noqa: E402
from oracle import vae_ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
TINY = dict(ch=(64, 128, 128, 128), cross=128, in_heads=2)


def tiny_inputs(F, H, W, seed=7):
    g = torch.Generator().manual_seed(seed)
    sample = torch.randn(2, 4, F, H, W, generator=g).half()
    ehs = torch.randn(2, 77, TINY["cross"], generator=g).half()
    return sample, ehs


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


class FP32UNetOnHalfIO(torch.nn.Module):
    """fp32 oracle UNet behind the reference's fp16 tensor interface."""

    def __init__(self, m):
        super().__init__()
        self.m = m
        self.config = m.config

    def forward(self, x, t, encoder_hidden_states):
        out = self.m(x.float(), t, encoder_hidden_states.float())
        out.sample = out.sample.half()
        return out


def main():
    torch.set_num_threads(8)
    cfg = UNet3DConfig.tiny(**TINY)
    sd = synthetic_state_dict(cfg, seed=1234)
    sd16 = {k: v.half() for k, v in sd.items()}
    m32 = UNet3DConditionModelRef(cfg).eval()
    m32.load_state_dict({k: v.float() for k, v in sd16.items()})       # fp16-rounded weights, fp32 math
    m16 = UNet3DConditionModelRef(cfg).eval().half()
    m16.load_state_dict(sd16)
    for tag, (F, H, W, t) in {"a": (4, 32, 32, 501), "b": (3, 32, 48, 981)}.items():
        sample, ehs = tiny_inputs(F, H, W)
        with torch.no_grad():
            o32 = m32(sample.float(), torch.tensor(t), ehs.float()).sample
            o16 = m16(sample, torch.tensor(t), ehs).sample
        floor = rel_l2(o16.float(), o32)
        print(f"unet_tiny_{tag}: out std {o32.std():.4f} max {o32.abs().max():.3f}  fp16-CPU floor rel-L2 {floor:.3e}")
        np.savez_compressed(os.path.join(HERE, f"unet_tiny_{tag}.npz"), out=o32.numpy(), floor=np.float64(floor),
                            shape=np.array([F, H, W, t]))
    # end-to-end: planner -> shared noise -> ctx -> 3 CFG/DDIM steps per chunk -> ramp blend
    T, H, W, steps = 10, 32, 32, 3
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(2, 77, TINY["cross"], generator=g).half()
    cond, uncond = emb[:1], emb[1:]
    lat, (cs, ov, ranges) = run_video(FP32UNetOnHalfIO(m32), DDIMSchedulerRef(), T, 4, H, W, world=1, steps=steps,
                                      uncond_emb=uncond, cond_emb=cond, chunk_size=6, overlap=2, mode="hybrid_ctx")
    print("denoise_tiny:", cs, ov, ranges, "lat std", float(lat.std()))
    np.savez_compressed(os.path.join(HERE, "denoise_tiny.npz"), lat=lat.numpy(), cs=cs, ov=ov,
                        ranges=np.array(ranges), T=T, H=H, W=W, steps=steps)


def xl_inputs():
    g = torch.Generator().manual_seed(31)
    sample = torch.randn(2, 4, 4, 32, 64, generator=g).half()          # CFG batch 2, 4 frames, 32x64 latent
    ehs = torch.randn(2, 77, 1024, generator=g).half()
    return sample, ehs


def unet_xl_main():
    """The FULL-WIDTH UNet (Zeroscope-XL configuration, 1.41 B seeded parameters) at a reduced extent: 4 frames of a
    32x64 latent give 16384 rows at level 0 — enough for the product to take its full-size kernel choices
    (weights-stationary GEMMs, 64-query flash blocks).  fp32 oracle output, stored as fp16."""
    torch.set_num_threads(8)
    cfg = UNet3DConfig.zeroscope()
    sd = {k: v.half().float() for k, v in synthetic_state_dict(cfg, seed=1234).items()}
    m = UNet3DConditionModelRef(cfg).eval()
    m.load_state_dict(sd)
    del sd
    sample, ehs = xl_inputs()
    with torch.no_grad():
        out = m(sample.float(), torch.tensor(501), ehs.float()).sample
    print(f"unet_xl_small: out std {out.std():.4f} max {out.abs().max():.3f}")
    np.savez_compressed(os.path.join(HERE, "unet_xl_small.npz"), out=out.numpy().astype(np.float16))


XL_CHUNK = dict(T=28, H=32, W=64, steps=50, windows=((0, 16), (16, 28)))   # a 16- and a 12-frame chunk (BASELINE cfg4 / cfg5)


def xl_chunk_embeddings():
    g = torch.Generator().manual_seed(41)
    emb = torch.randn(2, 77, 1024, generator=g).half()
    return emb[1:], emb[:1]                                             # uncond, cond


def unet_xl_chunks_main():
    """The per-rank workload of BASELINE configs 4 and 5 at a reduced extent: the FULL-WIDTH UNet on a 16-frame and a
    12-frame chunk of one shared-noise video (32x64 latent), with global-context injection, ONE step of
    `_denoise` (fsdp_chunked_coherent.py:129-143: cat, + 0.35 ctx, UNet, CFG combine, DDIM step at t = 981).
    Stored: eps (UNet output on the CFG batch) and the latent after the step, fp16."""
    from oracle.pipeline_ref import base_noise, global_context, denoise
    torch.set_num_threads(8)
    cfg = UNet3DConfig.zeroscope()
    sd = {k: v.half().float() for k, v in synthetic_state_dict(cfg, seed=1234).items()}
    m = UNet3DConditionModelRef(cfg).eval()
    m.load_state_dict(sd)
    del sd
    c = XL_CHUNK
    base = base_noise(c["T"], 4, c["H"], c["W"])
    ctx = global_context(c["T"], 4, c["H"], c["W"])
    uncond, cond = xl_chunk_embeddings()
    out = {}

    class Rec(FP32UNetOnHalfIO):
        def forward(self, x, t, encoder_hidden_states):
            o = super().forward(x, t, encoder_hidden_states)
            self.last = o.sample
            return o

    u = Rec(m)
    for s, e in c["windows"]:
        sched = DDIMSchedulerRef()
        sched.set_timesteps(c["steps"])
        sched.timesteps = sched.timesteps[:1]                           # one step, at the first timestep of the 50
        lat = denoise(u, sched, base[:, :, s:e].clone(), uncond, cond, 7.5, ctx, 0.35)
        print(f"xl chunk {e - s} f: eps std {u.last.float().std():.4f}  lat std {lat.float().std():.4f}")
        out[f"eps_{e - s}"] = u.last.numpy()
        out[f"lat_{e - s}"] = lat.numpy()
    np.savez_compressed(os.path.join(HERE, "unet_xl_chunks.npz"), **out)


CFG1 = dict(T=8, H=32, W=32, steps=3, of_steps=10)     # BASELINE config 1 at its real widths; 3 of its 10 DDIM steps


def cfg1_embeddings():
    g = torch.Generator().manual_seed(51)
    emb = torch.randn(2, 77, 1024, generator=g).half()
    return emb[1:], emb[:1]                                             # uncond, cond


def cfg1_main():
    """BASELINE config 1 (Zeroscope 576w shape: 8 frames @256x256 = 32x32 latent, chunk_only strategy) at the FULL
    width of the network: planner with the chunk_only overlap rule (`chunk_only.py:86`: cs 8, ov 2, windows (0,8),(6,8)),
    shared noise, the first 3 of the 10-step DDIM schedule per window, gather, ramp blend (`chunk_only.py:65-74,80-150`).
    The whole flow through `oracle.pipeline_ref.run_video` with the fp32 UNet; stored: the blended latent (fp32)."""
    from oracle.pipeline_ref import run_video
    torch.set_num_threads(8)
    cfg = UNet3DConfig.zeroscope()
    sd = {k: v.half().float() for k, v in synthetic_state_dict(cfg, seed=1234).items()}
    m = UNet3DConditionModelRef(cfg).eval()
    m.load_state_dict(sd)
    del sd
    c = CFG1
    uncond, cond = cfg1_embeddings()

    class FirstSteps(DDIMSchedulerRef):                                  # 10-step schedule, only its first `steps` entries
        def set_timesteps(self, n, device=None):
            super().set_timesteps(c["of_steps"], device)
            self.timesteps = self.timesteps[:c["steps"]]

    lat, (cs, ov, ranges) = run_video(FP32UNetOnHalfIO(m), FirstSteps(), c["T"], 4, c["H"], c["W"], world=1, steps=c["steps"],
                                      uncond_emb=uncond, cond_emb=cond, chunk_size=0, overlap=4, mode="chunk", rule="third")
    print("cfg1:", cs, ov, ranges, "lat std", float(lat.std()))
    np.savez_compressed(os.path.join(HERE, "cfg1_xl.npz"), lat=lat.numpy(), cs=cs, ov=ov, ranges=np.array(ranges))


# ---------------------------------------------------------------------------------------------------------------
# Full-schedule trajectories: error growth over a whole DDIM schedule, not over one forward
# ---------------------------------------------------------------------------------------------------------------
def round_leaf_outputs_to_fp16(model):
    """fp16-STORAGE emulation at fp32 speed: the output of every leaf module (convolution, Linear, GroupNorm, LayerNorm, ...)
    is rounded to fp16 and widened again, sums inside a module stay fp32 — what a GPU fp16 pipeline does between kernels.
    (The real fp16-CPU model is used where it finishes in minutes — tiny widths; torch's fp16 CPU kernels at the XL widths
    would take hours.)  Returns the hook handles."""
    def hook(_m, _inp, out):
        return out.half().float() if torch.is_tensor(out) and out.dtype == torch.float32 else out
    return [m.register_forward_hook(hook) for m in model.modules() if not list(m.children())]


class Trajectory(FP32UNetOnHalfIO):
    """Records the latent the loop feeds in at every step (x[:1] of the CFG batch, before the ctx term: ctx is None here or
    the caller subtracts nothing — only used with ctx-free modes or via `lat_log`)."""
    def __init__(self, m):
        super().__init__(m)
        self.eps_log = []

    def forward(self, x, t, encoder_hidden_states):
        o = super().forward(x, t, encoder_hidden_states)
        self.eps_log.append(o.sample.clone())
        return o


def denoise_logged(unet, sched, lat, uncond, cond, gs=7.5, ctx=None, cw=0.35):
    """`oracle.pipeline_ref.denoise` with the latent after every step kept (same statements, same order)."""
    log = []
    for t in sched.timesteps:
        x = sched.scale_model_input(torch.cat([lat] * 2), t)
        if ctx is not None:
            x = x + cw * ctx.repeat(1, 1, lat.shape[2], 1, 1)
        emb = torch.cat([uncond, cond], dim=0)
        with torch.no_grad():
            noise = unet(x, t, encoder_hidden_states=emb).sample
        u, c = noise.chunk(2)
        lat = sched.step(u + gs * (c - u), t, lat).prev_sample
        log.append(lat.clone())
    return lat, log


CFG1_FULL = dict(T=8, H=32, W=32, steps=10)     # BASELINE config 1 at the real widths, ALL 10 DDIM steps


def cfg1_full_main():
    """BASELINE config 1 with its whole schedule: Zeroscope-XL widths, 8 frames @ 256x256 (32x32 latent), chunk_only planner
    (windows (0,8),(6,8), ov 2), 10 DDIM steps per window, gather, ramp blend.  Two trajectories: the fp32 oracle (stored:
    the latent of window (0,8) after EVERY step, fp16; the blended latent, fp32) and the fp16-storage emulation of the same
    loop (stored: its rel-L2 distance to the fp32 trajectory after every step — the noise floor the HIP path is held to)."""
    from oracle.pipeline_ref import base_noise, plan_chunks, ramp_blend
    torch.set_num_threads(8)
    cfg = UNet3DConfig.zeroscope()
    sd = {k: v.half().float() for k, v in synthetic_state_dict(cfg, seed=1234).items()}
    m = UNet3DConditionModelRef(cfg).eval()
    m.load_state_dict(sd)
    del sd
    c = CFG1_FULL
    uncond, cond = cfg1_embeddings()
    cs, ov, ranges = plan_chunks(c["T"], 1, 0, 4, False, "third")
    base = base_noise(c["T"], 4, c["H"], c["W"])
    out = {"cs": cs, "ov": ov, "ranges": np.array(ranges)}
    finals = {}
    for tag in ("fp32", "fp16emu"):
        hooks = round_leaf_outputs_to_fp16(m) if tag == "fp16emu" else []
        u = FP32UNetOnHalfIO(m)
        chunks = []
        for s, e in ranges:
            sched = DDIMSchedulerRef()
            sched.set_timesteps(c["steps"])
            lat, log = denoise_logged(u, sched, base[:, :, s:e].clone(), uncond, cond)
            chunks.append((s, e, lat))
            finals[(tag, s, e)] = log
            print(f"cfg1_full {tag} window ({s},{e}): lat std {float(lat.float().std()):.4f}", flush=True)
        finals[(tag, "blend")] = ramp_blend(chunks, c["T"], ov, base)
        for h in hooks:
            h.remove()
    s0, e0 = ranges[0]
    ref_log, emu_log = finals[("fp32", s0, e0)], finals[("fp16emu", s0, e0)]
    out["lat_steps_w0"] = torch.stack([x[0] for x in ref_log]).numpy()                      # (steps, 4, 8, 32, 32) fp16
    out["floor_steps_w0"] = np.array([rel_l2(a.float(), b.float()) for a, b in zip(emu_log, ref_log)])
    out["lat"] = finals[("fp32", "blend")].numpy()
    out["floor_blend"] = np.float64(rel_l2(finals[("fp16emu", "blend")], finals[("fp32", "blend")]))
    print("cfg1_full: per-step fp16-emulation floor", np.array2string(out["floor_steps_w0"], precision=2), "blend", out["floor_blend"])
    np.savez_compressed(os.path.join(HERE, "cfg1_xl_full.npz"), **out)


SCHED50 = dict(T=12, H=16, W=16, steps=50, chunk=6, ov=2, every=5)


def sched50_main():
    """The reference's DEFAULT schedule length (50 steps, fsdp_chunked_coherent.py:284) on a `hybrid_ctx` job at tiny widths:
    12 frames of a 16x16 latent, windows of 6 / overlap 2, global-context injection.  fp32 oracle trajectory (stored: window
    0's latent after every 5th step and the blended latent) and the REAL fp16-CPU model's trajectory (stored: its rel-L2
    distance to the fp32 one at the same steps)."""
    from oracle.pipeline_ref import base_noise, global_context, plan_chunks, ramp_blend
    torch.set_num_threads(8)
    cfg = UNet3DConfig.tiny(**TINY)
    sd16 = {k: v.half() for k, v in synthetic_state_dict(cfg, seed=1234).items()}
    m32 = UNet3DConditionModelRef(cfg).eval()
    m32.load_state_dict({k: v.float() for k, v in sd16.items()})
    m16 = UNet3DConditionModelRef(cfg).eval().half()
    m16.load_state_dict(sd16)

    class Half(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m, self.config = m, m.config

        def forward(self, x, t, encoder_hidden_states):
            return self.m(x, t, encoder_hidden_states)

    c = SCHED50
    g = torch.Generator().manual_seed(1)
    emb = torch.randn(2, 77, TINY["cross"], generator=g).half()
    cond, uncond = emb[:1], emb[1:]
    cs, ov, ranges = plan_chunks(c["T"], 1, c["chunk"], c["ov"], False, "coherent")
    base = base_noise(c["T"], 4, c["H"], c["W"])
    ctx = global_context(c["T"], 4, c["H"], c["W"])
    res = {}
    for tag, u in (("fp32", FP32UNetOnHalfIO(m32)), ("fp16", Half(m16))):
        chunks = []
        for s, e in ranges:
            sched = DDIMSchedulerRef()
            sched.set_timesteps(c["steps"])
            lat, log = denoise_logged(u, sched, base[:, :, s:e].clone(), uncond, cond, 7.5, ctx, 0.35)
            chunks.append((s, e, lat))
            res[(tag, s, e)] = log
        res[(tag, "blend")] = ramp_blend(chunks, c["T"], ov, base)
    s0, e0 = ranges[0]
    idx = list(range(c["every"] - 1, c["steps"], c["every"]))
    out = {"cs": cs, "ov": ov, "ranges": np.array(ranges), "snap_steps": np.array(idx),
           "lat_snaps_w0": torch.stack([res[("fp32", s0, e0)][i][0] for i in idx]).numpy(),
           "floor_snaps_w0": np.array([rel_l2(res[("fp16", s0, e0)][i].float(), res[("fp32", s0, e0)][i].float()) for i in idx]),
           "lat": res[("fp32", "blend")].numpy(), "floor_blend": np.float64(rel_l2(res[("fp16", "blend")], res[("fp32", "blend")]))}
    print("sched50:", cs, ov, ranges, "fp16-CPU floor at steps", idx, np.array2string(out["floor_snaps_w0"], precision=2), "blend", out["floor_blend"])
    np.savez_compressed(os.path.join(HERE, "sched50_tiny.npz"), **out)


def vae_main():
    """AutoencoderKL decode (tiny widths, same topology): fp32 oracle output of seeded latents + the fp16-CPU
    noise floor of the same computation."""
    torch.set_num_threads(8)
    cfg = vae_ref.VaeConfig.tiny()
    sd16 = {k: v.half() for k, v in vae_ref.synthetic_state_dict(cfg, seed=4321).items()}
    m32 = vae_ref.AutoencoderKLRef(cfg).eval()
    m32.load_state_dict({k: v.float() for k, v in sd16.items()})
    m16 = vae_ref.AutoencoderKLRef(cfg).eval().half()
    m16.load_state_dict(sd16)
    g = torch.Generator().manual_seed(11)
    z = torch.randn(3, 4, 8, 16, generator=g).half()          # 3 frames, 8x16 latent -> 64x128 pixels
    with torch.no_grad():
        o32 = m32.decode(z.float()).sample
        o16 = m16.decode(z).sample
    floor = rel_l2(o16.float(), o32)
    print(f"vae_tiny: out std {o32.std():.4f} max {o32.abs().max():.3f}  fp16-CPU floor rel-L2 {floor:.3e}")
    np.savez_compressed(os.path.join(HERE, "vae_tiny.npz"), out=o32.numpy().astype(np.float16), floor=np.float64(floor))


def clip_tiny():
    """Tiny CLIP text tower (same topology as the SD-2.x one: causal, exact GELU, 64-wide heads) built from the
    REAL dependency, `transformers.CLIPTextModel`, with seeded weights — shared by the generator and the tests."""
    import transformers
    cfg = transformers.CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=512, num_hidden_layers=3,
                                      num_attention_heads=2, max_position_embeddings=77, hidden_act="gelu",
                                      layer_norm_eps=1e-5, projection_dim=64, bos_token_id=0, eos_token_id=2, pad_token_id=1)
    torch.manual_seed(77)
    m = transformers.CLIPTextModel(cfg).eval()
    with torch.no_grad():
        for k, v in m.state_dict().items():           # transformers' init is tiny (std 0.02): make activations O(1)
            if v.dtype.is_floating_point and v.dim() == 2 and "embedding" not in k:
                v.mul_(4.0)
            if k.endswith("bias"):
                v.add_(0.05 * torch.randn(v.shape))
        m.load_state_dict({k: (v.half().float() if v.dtype.is_floating_point else v) for k, v in m.state_dict().items()})
    ids = torch.randint(3, 1000, (2, 77), generator=torch.Generator().manual_seed(5))
    ids[:, 0] = 0
    ids[0, 20:] = 2                                    # a short prompt padded with EOS, like the tokenizer does
    return m, ids


def clip_main():
    m, ids = clip_tiny()
    with torch.no_grad():
        o32 = m(ids)[0]
        o16 = m.half()(ids)[0]
    floor = rel_l2(o16.float(), o32)
    print(f"clip_tiny: out std {o32.std():.4f} max {o32.abs().max():.3f}  fp16-CPU floor rel-L2 {floor:.3e}")
    np.savez_compressed(os.path.join(HERE, "clip_tiny.npz"), out=o32.numpy(), floor=np.float64(floor))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("unet", "all"):
        main()
    if what in ("unet_xl", "all"):
        unet_xl_main()
    if what in ("unet_xl_chunks", "all"):
        unet_xl_chunks_main()
    if what in ("cfg1", "all"):
        cfg1_main()
    if what in ("cfg1_full", "all"):
        cfg1_full_main()
    if what in ("sched50", "all"):
        sched50_main()
    if what in ("vae", "all"):
        vae_main()
    if what in ("clip", "all"):
        clip_main()

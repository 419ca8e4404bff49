"""Stand-in model objects shared by tests/golden/make_ref_fixtures.py (which runs the reference's own scripts on them in the
build container) and by the tests that replay the oracle / the HIP path on the same objects.  Not a test; no reference
code.  `diffusers` is absent from every machine this build sees: these objects have the attributes the reference's
scripts touch and nothing more."""
import types

import torch

TINY = dict(ch=(64, 128, 128, 128), cross=128, in_heads=2)


def text_table(cross=TINY["cross"]):
    """What the stand-in text encoder returns for [prompt, ""]: row 0 conditional, row 1 unconditional (reference :96-103)."""
    g = torch.Generator().manual_seed(1)
    return torch.randn(2, 77, cross, generator=g).half()


class ExactUNet(torch.nn.Module):
    """A UNet stand-in made of ELEMENTWISE fp16 tensor operations only (each one IEEE-rounded: the same bits on every CPU,
    with any thread count), so that everything downstream of it in the reference's loop — CFG combine, DDIM update, gather,
    blend — can be compared bit for bit.  It depends on everything the real model would see: the sample (mixing neighbouring
    frames, so a window's result depends on where the window was cut), the timestep, and WHICH text row sits in which batch
    slot (the [uncond, cond] order of :138)."""

    def __init__(self):
        super().__init__()
        self.config = types.SimpleNamespace(in_channels=4)
        self.p = torch.nn.Parameter(torch.zeros(1))
        self.calls = []

    def forward(self, x, t, encoder_hidden_states=None):
        self.calls.append((int(t), x.detach().clone()))
        w = encoder_hidden_states[:, 0, 0].to(x.dtype).view(-1, 1, 1, 1, 1)
        # (the timestep term is formed on the host in Python floats: torch divides by a scalar differently on CPU and GPU)
        tt = torch.tensor(float(int(t)) / 1000.0, dtype=torch.float32).to(device=x.device, dtype=x.dtype)
        y = x * 0.75 + torch.roll(x, 1, dims=2) * 0.125 + w * 0.0625 + tt * 0.03125
        return types.SimpleNamespace(sample=y)


def oracle_unet(record=None):
    """The fp32 oracle UNet (tiny widths, the seeded table of the other goldens) behind the reference's fp16 tensors.
    Its float arithmetic is NOT bit-reproducible across thread counts or CPUs: results through it carry a tolerance."""
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig, synthetic_state_dict
    cfg = UNet3DConfig.tiny(**TINY)
    m = UNet3DConditionModelRef(cfg).eval()
    m.load_state_dict({k: v.half().float() for k, v in synthetic_state_dict(cfg, seed=1234).items()})

    class HalfIO(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.m, self.config, self.calls = m, m.config, []

        def forward(self, x, t, encoder_hidden_states=None):
            self.calls.append((int(t), x.detach().clone()))
            with torch.no_grad():
                out = self.m(x.float(), t, encoder_hidden_states.float())
            return types.SimpleNamespace(sample=out.sample.half())
    return HalfIO()


def standin_decode(z):
    """The stand-in `vae.decode(z).sample`: a cheap deterministic image (1,3,8h,8w) of the latent frame it is handed."""
    img = torch.tanh(z[:, :3].float() * 0.18215)
    return img.repeat_interleave(8, dim=-1).repeat_interleave(8, dim=-2)


def frames_u8(z_frames):
    """reference :224-225 on the stand-in decoder's output -> uint8 (H,W,3) frames."""
    out = []
    for i in range(z_frames.shape[0]):
        img = standin_decode(z_frames[i:i + 1])
        img = (img[0].permute(1, 2, 0) * 0.5 + 0.5).clamp(0, 1)
        out.append((img * 255).byte().cpu().numpy())
    return out

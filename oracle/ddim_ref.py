"""ORACLE (test infrastructure): restatement of diffusers `DDIMScheduler` as the
reference uses it (`fsdp_chunked_coherent.py:95,115,132,133,142,182`).

[DEP] diffusers is absent (see oracle/__init__.py) — semantics restated from
SURVEY.md §8 a6 / Appendix B with the Zeroscope `scheduler_config.json` values:
num_train_timesteps 1000, beta_start 0.00085, beta_end 0.012, "scaled_linear",
clip_sample False, set_alpha_to_one False, steps_offset 1, epsilon prediction,
"leading" spacing, eta 0.  A second in-repo statement of the x0 half of the
update is `InferNet/template/validator/proof.py:377-380`.

Dtype behaviour mirrored: alphas are fp32 0-d tensors; the sample math runs in
the sample dtype (fp16 in the reference), one rounding per tensor op, exactly
as torch does when a 0-d fp32 tensor meets an fp16 tensor.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch


class DDIMSchedulerRef:
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 steps_offset=1, set_alpha_to_one=False):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                               dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, x, t=None):
        return x

    def coefficients(self, t: int):
        """(sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)) as fp32 0-d tensors."""
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (1 - a_t) ** 0.5, a_t ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5

    def step(self, eps, t, sample):
        s1, sa, sp, s1p = self.coefficients(int(t))
        x0 = (sample - s1 * eps) / sa
        prev = sp * x0 + s1p * eps
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

    def step_gpu_rules(self, eps, t, sample):
        """The same expression evaluated with torch's GPU type rules for fp16 tensors (what the
        reference executes: its tensors are on `cuda`).  torch-CPU differs in two places, measured
        in this container: (1) a 0-d fp32 tensor as the LEFT operand of `*` is first cast to fp16 on
        CPU, kept fp32 (opmath) on GPU; (2) `tensor / cpu_scalar` is a true division on CPU and
        `tensor * (1/scalar)` on GPU (`div_true_kernel_cuda`).  Every tensor op still rounds its
        result to fp16."""
        s1, sa, sp, s1p = (c.float() for c in self.coefficients(int(t)))
        r = lambda x: x.half()  # noqa: E731
        e, x = eps.float(), sample.float()
        a2 = r(x - r(s1 * e).float())
        x0 = r(a2.float() * (torch.tensor(1.0) / sa))
        prev = r(r(sp * x0.float()).float() + r(s1p * e).float())
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)

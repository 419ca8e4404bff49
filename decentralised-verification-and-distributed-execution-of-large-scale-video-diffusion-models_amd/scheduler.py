"""`DDIMScheduler` with the diffusers call surface the reference uses
(`fsdp_chunked_coherent.py:95,115,132,133,142,182`): `set_timesteps(n, device=)`, `.timesteps`,
`.scale_model_input(x, t)`, `.step(eps, t, x).prev_sample`, `.init_noise_sigma`.

Schedule tables (betas, cumulative alphas, timesteps) are host logic; the sample update runs in
the fused HIP kernel (`vdx_ddim_step_f16` / `vdx_cfg_ddim_step_f16`) — there is no CPU step.
Config = Zeroscope `scheduler_config.json`: 1000 train steps,
scaled-linear betas 0.00085..0.012, steps_offset 1, set_alpha_to_one False, epsilon, eta 0,
leading spacing, no clipping.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops


class DDIMScheduler:
    init_noise_sigma = 1.0
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012,
                 beta_schedule="scaled_linear", steps_offset=1, set_alpha_to_one=False,
                 clip_sample=False, prediction_type="epsilon", timestep_spacing="leading"):
        if beta_schedule != "scaled_linear" or clip_sample or prediction_type != "epsilon" \
                or timestep_spacing != "leading":
            raise NotImplementedError("only the Zeroscope DDIM configuration is implemented")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule,
                                      steps_offset=steps_offset, set_alpha_to_one=set_alpha_to_one,
                                      clip_sample=clip_sample, prediction_type=prediction_type,
                                      timestep_spacing=timestep_spacing)
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None
        self._host_timesteps = None

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError("num_inference_steps exceeds num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self._host_timesteps = [int(t) for t in ts]
        self.timesteps = torch.from_numpy(ts).to(device)

    def scale_model_input(self, sample, timestep=None):
        return sample

    def coefficients(self, t: int):
        """(sqrt(1-a_t), sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)) evaluated in fp32 like diffusers."""
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        return (float((1 - a_t) ** 0.5), float(a_t ** 0.5), float(a_prev ** 0.5), float((1 - a_prev) ** 0.5))

    def _host_timestep(self, timestep) -> int:
        """The step's timestep as a host integer, by VALUE (diffusers semantics) and — for the tensors the reference
        hands in — without reading device memory.  The reference passes `step` the 0-d device tensors it iterates
        over (`for t in scheduler.timesteps`, :132,142); `int(t)` on those is a device sync per step.  Such an element
        is a view of `self.timesteps`' storage, so its index is its address: (data_ptr - base) / itemsize, checked
        against the storage's extent.  Any other device tensor (a clone, arithmetic on a timestep, a foreign schedule)
        is read with `int(t)` — one sync, never a guess.  Python numbers and host tensors are used as given."""
        if torch.is_tensor(timestep) and timestep.is_cuda:
            ts = self.timesteps
            if ts is not None and ts.is_cuda and timestep.numel() == 1 and timestep.dtype == ts.dtype \
                    and timestep.device == ts.device \
                    and timestep.untyped_storage().data_ptr() == ts.untyped_storage().data_ptr():
                off = timestep.data_ptr() - ts.data_ptr()
                idx, rem = divmod(off, ts.element_size())
                if rem == 0 and 0 <= idx < ts.numel():
                    return self._host_timesteps[idx]
            return int(timestep)
        return int(timestep)

    def step(self, model_output, timestep, sample, eta: float = 0.0, **_unused):
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is not used by the reference path")
        if self.num_inference_steps is None:
            raise ValueError("call set_timesteps first")
        prev = ops.ddim_step(model_output.contiguous(), sample.contiguous(),
                             self.coefficients(self._host_timestep(timestep)))
        return SimpleNamespace(prev_sample=prev)

    def step_cfg(self, noise2, timestep, sample, guidance_scale: float):
        """Fused `u + gs*(c-u)` + step (fsdp_chunked_coherent.py:141-142) in one kernel."""
        return ops.cfg_ddim_step(noise2, sample, guidance_scale, self.coefficients(self._host_timestep(timestep)))

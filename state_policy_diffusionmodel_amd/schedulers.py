"""Host-side scheduler objects with the surface the reference's callers use on
diffusers' schedulers: construction kwargs (models/diffusion_ddpm.py:65-70,
generate.py:28-33), ``set_timesteps(n)``, ``.timesteps``,
``.config.num_train_timesteps`` and -- for callers that drive the loop themselves --
``step(eps, t, x).prev_sample`` / ``add_noise``.

They hold NO device code: they only produce the per-step coefficient table that
``spdm_set_schedule_tables`` (include/spdm.h) consumes, computed with torch fp32
scalars in the operation order of diffusers 0.17.1 so that the table is what the
reference's scheduler would use (the library can also build the table itself,
``spdm_set_schedule``; tests compare the two).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from ._lib import SPDM_DDIM, SPDM_DDPM


class _LinearBetaScheduler:
    kind = -1

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 1e-4, beta_end: float = 0.02,
                 beta_schedule: str = "linear", clip_sample: bool = False, prediction_type: str = "epsilon", **kw):
        if beta_schedule != "linear":
            raise NotImplementedError("the reference path only ever builds beta_schedule='linear' "
                                      "(models/diffusion_ddpm.py:67)")
        if clip_sample or prediction_type != "epsilon":
            raise NotImplementedError("reference path: clip_sample=False, prediction_type='epsilon'")
        self.config = SimpleNamespace(num_train_timesteps=int(num_train_timesteps), beta_start=beta_start,
                                      beta_end=beta_end, beta_schedule=beta_schedule, clip_sample=clip_sample,
                                      prediction_type=prediction_type, **kw)
        T = self.config.num_train_timesteps
        self.betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.final_alpha_cumprod = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = T
        self.timesteps = torch.from_numpy(np.arange(0, T)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        T = self.config.num_train_timesteps
        if num_inference_steps > T:
            raise ValueError(f"num_inference_steps ({num_inference_steps}) > num_train_timesteps ({T})")
        self.num_inference_steps = int(num_inference_steps)
        ratio = T // self.num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        self.timesteps = torch.from_numpy(ts)

    def _prev(self, t: int) -> int:
        return t - self.config.num_train_timesteps // self.num_inference_steps

    def add_noise(self, original, noise, timesteps):
        acp = self.alphas_cumprod.to(original.device)
        sa = acp[timesteps] ** 0.5
        sb = (1 - acp[timesteps]) ** 0.5
        while sa.dim() < original.dim():
            sa, sb = sa.unsqueeze(-1), sb.unsqueeze(-1)
        return sa * original + sb * noise

    def coefficient_table(self) -> np.ndarray:
        """(n, 6) fp32: [sqrt(1-abar_t), sqrt(abar_t), k_x0, k_x, k_eps, k_noise] per loop iteration."""
        rows = [self._coef(int(t)) for t in self.timesteps.tolist()]
        return np.asarray(rows, dtype=np.float32)

    def step(self, model_output, timestep, sample, noise: Optional[torch.Tensor] = None, generator=None):
        c = [torch.tensor(v) for v in self._coef(int(timestep))]
        x0 = (sample - c[0] * model_output) / c[1]
        if self.kind == SPDM_DDPM:
            prev = c[2] * x0 + c[3] * sample
            if int(timestep) > 0:
                if noise is None:
                    noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype,
                                        device=model_output.device)
                prev = prev + c[5] * noise
        else:
            prev = c[2] * x0 + c[4] * model_output
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


class DDPMScheduler(_LinearBetaScheduler):
    """Stand-in for diffusers.schedulers.scheduling_ddpm.DDPMScheduler (0.17.1),
    epsilon prediction, fixed_small variance."""
    kind = SPDM_DDPM

    def _coef(self, t: int):
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        k_x0 = (a_prev ** 0.5 * cur_b) / b_t
        k_x = cur_a ** 0.5 * b_prev / b_t
        k_noise = torch.tensor(0.0)
        if t > 0:
            var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
            k_noise = var ** 0.5
        return [float(b_t ** 0.5), float(a_t ** 0.5), float(k_x0), float(k_x), 0.0, float(k_noise)]


class DDIMScheduler(_LinearBetaScheduler):
    """Stand-in for diffusers.schedulers.scheduling_ddim.DDIMScheduler (0.17.1), eta = 0."""
    kind = SPDM_DDIM

    def _coef(self, t: int):
        prev_t = self._prev(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        std = torch.tensor(0.0)
        k_eps = (1 - a_prev - std ** 2) ** 0.5
        return [float(b_t ** 0.5), float(a_t ** 0.5), float(a_prev ** 0.5), 0.0, float(k_eps), 0.0]

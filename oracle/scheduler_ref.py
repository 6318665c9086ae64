"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the scheduler arithmetic and of the sampling loop that the
reference runs per denoise step.

* The loop: ``Diffusion_DDPM.sample`` (``/root/reference/models/diffusion_ddpm.py:223-277``),
  its byte-identical DDIM copy (``models/diffusion_ddim.py:23-74``) and the
  inpainting overwrite ``add_constraints`` (``models/diffusion_ddpm.py:216-219``).
* The scheduler: third-party, NOT vendored in the reference and NOT installed
  here -- ``diffusers==0.17.1`` (pin: ``/root/reference/requirements.txt:31``),
  classes ``schedulers.scheduling_ddpm.DDPMScheduler`` and
  ``schedulers.scheduling_ddim.DDIMScheduler``, constructed at
  ``models/diffusion_ddpm.py:65-70`` / ``generate.py:28-33`` with
  ``beta_schedule='linear', clip_sample=False, prediction_type='epsilon'`` and
  library defaults otherwise (beta 1e-4..0.02, ``variance_type='fixed_small'``,
  DDIM ``eta=0``, ``set_alpha_to_one=True``, ``steps_offset=0``).  What follows
  restates the published algorithm (Ho et al. 2020 Eq. 6-7/15; Song et al. 2021
  Eq. 12) in the operation order of that release, in torch fp32 scalars/tensors.

PARITY UNPINNED for the scheduler: the reference holds no test, fixture or
golden vector at this boundary and the library itself is unavailable, so this
restatement is guarded only by the closed-form property tests in
``tests/test_scheduler.py``.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch


class LinearBetaSchedule:
    """betas/alphas_cumprod as the scheduler constructors build them: fp32
    ``linspace`` -> ``1 - betas`` -> ``cumprod``; ``set_timesteps(n)`` gives
    ``(arange(n) * (T // n))[::-1]`` as int64."""

    def __init__(self, num_train_timesteps: int, beta_start: float = 1e-4, beta_end: float = 0.02):
        self.T = int(num_train_timesteps)
        self.betas = torch.linspace(beta_start, beta_end, self.T, dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.num_inference_steps = self.T
        self.timesteps = torch.arange(self.T - 1, -1, -1, dtype=torch.int64)

    def set_timesteps(self, n: int):
        self.num_inference_steps = int(n)
        ratio = self.T // self.num_inference_steps
        self.timesteps = (torch.arange(0, n, dtype=torch.int64) * ratio).flip(0)

    def prev_timestep(self, t: int) -> int:
        return t - self.T // self.num_inference_steps


def ddpm_step(s: LinearBetaSchedule, eps: torch.Tensor, t: int, x: torch.Tensor,
              noise: Optional[torch.Tensor]) -> torch.Tensor:
    """DDPMScheduler.step(...).prev_sample, epsilon prediction, fixed_small
    variance, no clipping.  ``noise`` is the N(0,1) draw the library would take
    from the global RNG when t > 0 (ignored at t == 0)."""
    t = int(t)
    prev_t = s.prev_timestep(t)
    a_t = s.alphas_cumprod[t]
    a_prev = s.alphas_cumprod[prev_t] if prev_t >= 0 else s.one
    b_t = 1 - a_t
    b_prev = 1 - a_prev
    cur_a = a_t / a_prev
    cur_b = 1 - cur_a
    x0 = (x - b_t ** 0.5 * eps) / a_t ** 0.5
    c_x0 = (a_prev ** 0.5 * cur_b) / b_t
    c_x = cur_a ** 0.5 * b_prev / b_t
    prev = c_x0 * x0 + c_x * x
    if t > 0:
        var = torch.clamp((1 - a_prev) / (1 - a_t) * cur_b, min=1e-20)
        prev = prev + (var ** 0.5) * noise
    return prev


def ddim_step(s: LinearBetaSchedule, eps: torch.Tensor, t: int, x: torch.Tensor) -> torch.Tensor:
    """DDIMScheduler.step(...).prev_sample with eta = 0 (deterministic)."""
    t = int(t)
    prev_t = s.prev_timestep(t)
    a_t = s.alphas_cumprod[t]
    a_prev = s.alphas_cumprod[prev_t] if prev_t >= 0 else s.one
    b_t = 1 - a_t
    x0 = (x - b_t ** 0.5 * eps) / a_t ** 0.5
    std = torch.tensor(0.0)
    direction = (1 - a_prev - std ** 2) ** 0.5 * eps
    return a_prev ** 0.5 * x0 + direction


@torch.no_grad()
def sample_loop(unet: Callable[[torch.Tensor, torch.Tensor, torch.Tensor], torch.Tensor],
                kind: str, num_train_timesteps: int, num_inference_steps: int,
                cond: torch.Tensor, x_T: torch.Tensor, noise: Optional[torch.Tensor],
                inpaint: Optional[torch.Tensor], history: bool = False):
    """The reference loop (models/diffusion_ddpm.py:252-277), batched: the
    reference hard-wires B = 1 (``obs_cond[0]``, ``:246``); the U-Net is
    batch-agnostic, so B independent trajectories are B reference runs.

    cond (B,1,obs_h,obs_dim); x_T (B,1,H,D) -- the reference draws it with
    ``torch.rand`` (uniform, ``:252``); noise (N,B,1,H,D), row i used by loop
    iteration i when its t > 0; inpaint (B,1,inp_h,D) or (1,1,inp_h,D) or None.
    Returns x_0 (B,1,H,D), or the list [x_T, x_{T-1}, ..., x_0] when history."""
    s = LinearBetaSchedule(num_train_timesteps)
    s.set_timesteps(num_inference_steps)
    x = x_T.clone().to(torch.float32)
    hist: List[torch.Tensor] = [x.clone()]
    for i, t in enumerate(s.timesteps.tolist()):
        eps = unet(x, torch.tensor([t]), cond)
        if kind == "ddpm":
            x = ddpm_step(s, eps, t, x, None if noise is None else noise[i])
        elif kind == "ddim":
            x = ddim_step(s, eps, t, x)
        else:
            raise ValueError(kind)
        if inpaint is not None:
            x[:, :, :inpaint.shape[2], :] = inpaint          # add_constraints, :216-219
        if history:
            hist.append(x.clone())
    return hist if history else x

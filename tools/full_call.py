#!/usr/bin/env python3
"""One whole `sample()` call at the headline geometry, from raw observation frames to the final trajectories:
frame encoder (once) + 1000 DDPM steps at B = 4096, H = 32, D = 3 through the facade.  Run on the GPU box."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM  # noqa: E402
from state_policy_diffusionmodel_amd.vision import ENCODER_SHAPES  # noqa: E402

B, oh, N = int(os.environ.get("FULL_B", 4096)), 10, int(os.environ.get("FULL_N", 1000))
g = torch.Generator().manual_seed(0)
enc_sd = {k: (torch.rand(s, generator=g) - 0.5) * 0.1 for k, s in ENCODER_SHAPES.items()}
m = Diffusion_DDPM(noise_steps=N, obs_horizon=oh, pred_horizon=31, observation_dim=135, prediction_dim=3,
                   model="UNet_Film", inpaint_horizon=1, weight_seed=0, max_batch=B, vision_encoder_state_dict=enc_sd)
dev = m.device
gd = torch.Generator(device=dev).manual_seed(1)
batch = {"image": torch.rand(B, oh, 3, 96, 96, device=dev, generator=gd),
         "position": torch.rand(B, oh, 2, device=dev, generator=gd) * 2 - 1,
         "velocity": torch.rand(B, oh, 2, device=dev, generator=gd),
         "action": torch.rand(B, oh, 3, device=dev, generator=gd)}
inpaint = torch.rand(B, 1, 3, device=dev, generator=gd) * 2 - 1
x_T = torch.rand(B, 1, 32, 3, device=dev, generator=gd)


def call(n_steps):
    m.noise_steps = n_steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ob = m.prepare_observation_batch(batch)
    cond = m.prepare_obs_cond_vectors(ob)                 # frame encoder + concat (models/diffusion_ddpm.py:317-330)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = m.sample({"obs_cond": cond, "inpaint": inpaint}, x_T=x_T.clone(), batched=True, seed=5)
    torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t1, out


call(4)                                                  # warm-up: engine, weights, tables, encoder
t_enc, t_loop, out = call(N)
print(json.dumps({"workload": f"whole sample() call: {B} trajectories, {B * oh} frames of 3x96x96 -> obs_cond, {N}-step DDPM, "
                              "H=32, D=3, UNet_Film attention on, device Philox noise",
                  "front_end_s": t_enc, "loop_s": t_loop, "ms_per_step": t_loop / N * 1e3,
                  "trajectory_steps_per_s": B * N / t_loop, "front_end_share": t_enc / (t_enc + t_loop),
                  "finite": bool(torch.isfinite(out).all()), "absmax": float(out.abs().max())}))

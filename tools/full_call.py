#!/usr/bin/env python3
"""Whole `sample()` calls through the facade, from raw observation frames to the final trajectories (run on the GPU box).

    FULL_B=4096 FULL_N=1000 FULL_KIND=ddpm python tools/full_call.py      # the headline geometry (default)
    FULL_B=1 FULL_KIND=ddim FULL_N=100 python tools/full_call.py           # generate.py:23's default call
    FULL_B=1 FULL_KIND=ddpm FULL_N=1000 python tools/full_call.py          # run_predictions.py:151-156's call

Reports the front end (frame encoder + concat), the loop, and -- from calls of two lengths -- the per-call overhead of the
loop entry (FiLM hoist, time-embedding lookup, schedule install, graph capture on the first call only): a straight line
t(N) = overhead + N * step through the medians of repeated calls at N and N / 2."""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from state_policy_diffusionmodel_amd.diffusion import load_model  # noqa: E402
from state_policy_diffusionmodel_amd.vision import ENCODER_SHAPES  # noqa: E402

B, oh, N = int(os.environ.get("FULL_B", 4096)), 10, int(os.environ.get("FULL_N", 1000))
KIND = os.environ.get("FULL_KIND", "ddpm").lower()
REPS = int(os.environ.get("FULL_REPS", 5 if B <= 64 else 1))
g = torch.Generator().manual_seed(0)
enc_sd = {k: (torch.rand(s, generator=g) - 0.5) * 0.1 for k, s in ENCODER_SHAPES.items()}
hp = dict(noise_steps=1000, obs_horizon=oh, pred_horizon=31, observation_dim=135, prediction_dim=3, model="UNet_Film",
          inpaint_horizon=1, weight_seed=0, max_batch=B, vision_encoder_state_dict=enc_sd)
m = load_model("DDIM", num_of_ddim_steps=N, **hp) if KIND == "ddim" else load_model("DDPM", **hp)
dev = m.device
gd = torch.Generator(device=dev).manual_seed(1)
batch = {"image": torch.rand(B, oh, 3, 96, 96, device=dev, generator=gd),
         "position": torch.rand(B, oh, 2, device=dev, generator=gd) * 2 - 1,
         "velocity": torch.rand(B, oh, 2, device=dev, generator=gd),
         "action": torch.rand(B, oh, 3, device=dev, generator=gd)}


def call(n_steps):
    """one caller-side iteration: fresh tensors every time, as run_predictions.py builds them"""
    if KIND == "ddim":
        from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler
        m.noise_scheduler = DDIMScheduler(num_train_timesteps=n_steps)         # generate.py:28-35
    m.noise_steps = n_steps
    inpaint = torch.rand(B, 1, 3, device=dev, generator=gd) * 2 - 1
    x_T = torch.rand(B, 1, 32, 3, device=dev, generator=gd)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ob = m.prepare_observation_batch(batch)
    cond = m.prepare_obs_cond_vectors(ob)                 # frame encoder + concat (models/diffusion_ddpm.py:317-330)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out = m.sample({"obs_cond": cond, "inpaint": inpaint}, x_T=x_T, batched=True, seed=5)
    torch.cuda.synchronize()
    return t1 - t0, time.perf_counter() - t1, out


call(max(4, min(N, 8)))                                  # warm-up: engine, weights, tables, encoder, first capture
full = [call(N) for _ in range(REPS)]
half = [call(max(N // 2, 3)) for _ in range(REPS)]
t_enc = statistics.median(c[0] for c in full)
t_full = statistics.median(c[1] for c in full)
t_half = statistics.median(c[1] for c in half)
n_half = max(N // 2, 3)
step = (t_full - t_half) / (N - n_half)
overhead = t_full - N * step
out = full[-1][2]
print(json.dumps({"workload": f"whole sample() call through the facade: {B} trajectories, {B * oh} frames of 3x96x96 -> obs_cond, "
                              f"{N}-step {KIND.upper()}, H=32, D=3, UNet_Film attention on, device Philox noise; medians of {REPS} calls",
                  "front_end_s": t_enc, "loop_s": t_full, "ms_per_step_whole_call": t_full / N * 1e3,
                  "ms_per_step_marginal": step * 1e3, "per_call_overhead_ms": overhead * 1e3,
                  "per_call_overhead_share": overhead / t_full,
                  "trajectory_steps_per_s": B * N / t_full, "front_end_share": t_enc / (t_enc + t_full),
                  "graph_captures": m._engine.graph_captures,
                  "finite": bool(torch.isfinite(out).all()), "absmax": float(out.abs().max())}))

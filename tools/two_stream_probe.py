#!/usr/bin/env python3
"""Probe: does running two half-batches concurrently on two HIP streams beat one full batch?  (GPU box)
usage: two_stream_probe.py [B] -- times K denoise steps of one engine at batch B against two engines at B/2 on two streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from state_policy_diffusionmodel_amd.engine import SpdmEngine
from state_policy_diffusionmodel_amd.schedulers import DDPMScheduler
from state_policy_diffusionmodel_amd.weights import random_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, W = 40, 5
sd = random_state_dict(1350, seed=0)


def make(b):
    e = SpdmEngine(32, 3, 1350, max_batch=b)
    e.load_state_dict(sd)
    s = DDPMScheduler(num_train_timesteps=1000)
    s.set_timesteps(1000)
    e.set_scheduler(s)
    g = torch.Generator().manual_seed(b)
    e.sample_begin(torch.randn(b, 1, 10, 135, generator=g).cuda(), torch.rand(b, 1, 32, 3, generator=g).cuda(), seed=3)
    return e


full = make(B)
full.sample_run(0, W)
torch.cuda.synchronize()
t0 = time.perf_counter()
full.sample_run(W, W + K)
torch.cuda.synchronize()
t_full = (time.perf_counter() - t0) / K * 1e3

for parts in (2, 4):
    engs = [make(B // parts) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.sample_run(0, W)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for e, s in zip(engs, streams):
        with torch.cuda.stream(s):
            e.sample_run(W, W + K)
    torch.cuda.synchronize()
    t_split = (time.perf_counter() - t0) / K * 1e3
    print(f"B={B}: one engine {t_full:.3f} ms/step; {parts} engines of {B // parts} on {parts} streams {t_split:.3f} ms/step", flush=True)
    for e in engs:
        e.close()

#!/usr/bin/env python3
"""Do two builds of the library produce the same bits?  (SPDM_LIB selects the build: run once per build, compare the digests.)
usage: [SPDM_LIB=...] python tools/probes/same_bits.py [batch ...]"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from state_policy_diffusionmodel_amd.engine import SpdmEngine
from state_policy_diffusionmodel_amd.weights import random_state_dict

H, D, cd = 32, 3, 1350
sd = random_state_dict(cd, seed=0, attention=True)
for B in [int(a) for a in sys.argv[1:]] or [1, 64, 512, 4096]:
    eng = SpdmEngine(H, D, cd, max_batch=B, attention=True, num_train_timesteps=1000)
    eng.load_state_dict(sd)
    eng.set_builtin_schedule(0, 1000, 1000)
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(B, 1, 10, 135, generator=g).cuda()
    x_T = torch.rand(B, 1, H, D, generator=g).cuda()
    eng.sample_begin(cond, x_T, noise=None, inpaint=None, seed=7)
    eng.sample_run(0, 3)
    out = eng.sample_result().cpu().numpy()
    print(B, hashlib.sha256(out.tobytes()).hexdigest()[:16], float(abs(out).max()))
    if os.environ.get("BITS_DUMP"):
        import numpy as np
        np.save(f"{os.environ['BITS_DUMP']}_b{B}.npy", out)
    eng.close()

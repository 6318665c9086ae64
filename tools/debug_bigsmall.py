#!/usr/bin/env python3
"""Debug helper: eps of a big batch vs the same trajectories alone vs the oracle (GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.unet_film_ref import unet_film_forward
from state_policy_diffusionmodel_amd.engine import SpdmEngine
from state_policy_diffusionmodel_amd.weights import random_state_dict
B, H, D = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = random_state_dict(1350, seed=0)
g = torch.Generator().manual_seed(2)
cond = torch.randn(B, 1, 10, 135, generator=g)
x = torch.rand(B, 1, H, D, generator=g)
eng = SpdmEngine(H, D, 1350, max_batch=B)
eng.load_state_dict(sd)
for t in (49, 3):
    big = eng.unet_forward(x.cuda(), [t], cond.cuda()).cpu()
    for i in (0, 1, B // 2 - 1, B - 1):
        small = eng.unet_forward(x[i:i+1].cuda(), [t], cond[i:i+1].cuda()).cpu()
        ref = unet_film_forward(sd, x[i:i+1], torch.tensor([t]), cond[i:i+1])
        print(f"t={t} i={i}: |big-ref|={float((big[i]-ref[0]).abs().max()):.2e} |small-ref|={float((small[0]-ref[0]).abs().max()):.2e} |big-small|={float((big[i]-small[0]).abs().max()):.2e}", flush=True)
eng.close()

#!/usr/bin/env python3
"""Actual deviation of the HIP path from the reference's golden vectors (tests assert <= 1e-4; this prints the value).
Run on the GPU box: python tools/golden_error.py"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_parity import make_engine, weights  # noqa: E402

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "unet_*.npz"))):
    g = np.load(path)
    H, D, B = int(g["H"]), int(g["D"]), int(g["B"])
    cond_dim = int(g["obs_h"]) * int(g["obs_dim"])
    attention = bool(int(g["attention"]))
    sd = weights(cond_dim, int(g["wseed"]), attention, str(g["weights_sha256"]))
    for exact in (False, True):
        eng = make_engine(H, D, cond_dim, B, sd, attention, exact_fp32=exact)
        x, cond = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["cond"]).cuda()
        worst = 0.0
        for t, want in zip(g["t"], g["eps"]):
            got = eng.unet_forward(x, np.atleast_1d(t), cond).cpu().numpy()
            worst = max(worst, float(np.abs(got - want).max()))
        eng.close()
        print(f"{os.path.basename(path):34s} {'exact fp32 MFMA' if exact else 'split-fp16 MFMA'}: max|eps - reference| = {worst:.2e}"
              f"   (max|eps| = {float(np.abs(g['eps']).max()):.2f})", flush=True)

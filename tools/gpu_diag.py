#!/usr/bin/env python3
"""First-light diagnostic on the GPU box: every U-Net golden, per-block taps, max abs error.
Writes gpurun_out/diag.log.  (Checker = committed golden vectors from the imported reference.)"""
import glob
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from state_policy_diffusionmodel_amd.engine import SpdmEngine
from state_policy_diffusionmodel_amd.weights import random_state_dict

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
log = open(os.path.join(OUT, "diag.log"), "w")


def say(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    log.write(s + "\n")
    log.flush()


TAPS = {"inc": "x1", "down1": "d1", "sa1": "x2", "down2": "d2", "sa2": "x3", "down3": "d3", "sa3": "x4",
        "bot3": "x5", "up1": "u1", "sa4": "a4", "up2": "u2", "sa5": "a5", "up3": "u3", "sa6": "a6"}

say("device:", torch.cuda.get_device_name(0))
only = sys.argv[1:]
for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "unet_*.npz"))):
    if only and not any(o in path for o in only):
        continue
    g = np.load(path)
    name = os.path.basename(path)
    try:
        H, D, B = int(g["H"]), int(g["D"]), int(g["B"])
        cond_dim = int(g["obs_h"]) * int(g["obs_dim"])
        attention = bool(int(g["attention"]))
        sd = random_state_dict(cond_dim, seed=int(g["wseed"]), attention=attention)
        t0 = time.time()
        eng = SpdmEngine(H, D, cond_dim, max_batch=B, attention=attention, debug=True,
                         exact_fp32=os.environ.get('DIAG_EXACT') == '1')
        eng.load_state_dict(sd)
        x = torch.from_numpy(g["x"]).cuda()
        cond = torch.from_numpy(g["cond"]).cuda()
        for t, want in zip(g["t"], g["eps"]):
            got = eng.unet_forward(x, np.atleast_1d(t), cond).cpu().numpy()
            err = np.abs(got - want).max()
            say(f"{name} t={np.atleast_1d(t).tolist()} eps max|d|={err:.3e} (|want|max={np.abs(want).max():.3f}) "
                f"{'OK' if err <= 1e-4 else 'FAIL'}")
        if any(k.startswith("tap_") for k in g.files):
            eng.unet_forward(x, np.atleast_1d(g["t"][0]), cond)
            for ref_name, mine in TAPS.items():
                if "tap_" + ref_name not in g.files:
                    continue
                try:
                    got = eng.debug_tensor(mine).cpu().numpy()
                    want = g["tap_" + ref_name]
                    say(f"   tap {ref_name:6s} shape {got.shape} max|d|={np.abs(got - want).max():.3e} "
                        f"(|want|max={np.abs(want).max():.3f})")
                except Exception as e:
                    say(f"   tap {ref_name}: {e}")
        say(f"   ({time.time() - t0:.1f}s, device bytes {eng.device_bytes / 1e6:.1f} MB)")
        eng.close()
    except Exception:
        say(name, "EXCEPTION\n" + traceback.format_exc())
say("done")

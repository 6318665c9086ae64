#!/usr/bin/env python3
"""Phase timeline of sa_fused64_kernel's workgroup 0 in a real step (diagnostic build only:
   SPDM_EXTRA_FLAGS=-DSPDM_DIAG_SAF SPDM_BUILD_OUT=.../libspdm_saf.so python -m state_policy_diffusionmodel_amd.build).
usage: SPDM_LIB=.../libspdm_saf.so python tools/probes/saf_stamps.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from state_policy_diffusionmodel_amd import _lib
from state_policy_diffusionmodel_amd.engine import SpdmEngine
from state_policy_diffusionmodel_amd.weights import random_state_dict

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, D, cd = 32, 3, 1350
eng = SpdmEngine(H, D, cd, max_batch=B, attention=True, num_train_timesteps=1000)
eng.load_state_dict(random_state_dict(cd, seed=0, attention=True))
eng.set_builtin_schedule(0, 1000, 1000)
g = torch.Generator().manual_seed(1)
cond = torch.randn(B, 1, 10, 135, generator=g).cuda()
x_T = torch.rand(B, 1, H, D, generator=g).cuda()
eng.sample_begin(cond, x_T, noise=None, inpaint=None, seed=7)
eng.sample_run(0, 12)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 64)()
assert lib.spdm_debug_saf_stamps(out) == 0
names = ["weights (qkv, out-proj) -> LDS, FiLM coefficients"]
for p in range(2):
    names.append(f"pair {p}: x load, LayerNorm 1, q k v products, bias")
    for sub in range(2):
        names += [f"  head {2 * p + sub}: split, K / V^T -> LDS, barriers", f"  head {2 * p + sub}: attention loop", f"  head {2 * p + sub}: out-proj"]
names += ["bias + residual", "weights (ff1, ff2) -> LDS", "LayerNorm 2, ff1, GELU", "ff2", "store"]
for base, what in ((0, "sa6: 256 tokens"), (32, "sa5: 64 tokens")):
    st = [out[base + i] for i in range(len(names) + 1)]
    print(f"sa_fused64_kernel, {what} (B = {B}), workgroup 0 thread 0: us per phase")
    for i, n in enumerate(names):
        print(f"  {n:56s} +{(st[i + 1] - st[i]) * 0.01:6.2f}   = {(st[i + 1] - st[0]) * 0.01:7.2f}")

#!/usr/bin/env python3
"""Phase timeline of sa_tail_kernel's workgroup 0 in a real step (diagnostic build only:
   SPDM_EXTRA_FLAGS=-DSPDM_DIAG_TAIL SPDM_BUILD_OUT=.../libspdm_tail.so python -m state_policy_diffusionmodel_amd.build).
usage: SPDM_LIB=.../libspdm_tail.so python tools/probes/tail_stamps.py [batch]"""
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
g = torch.Generator().manual_seed(1)
cond = torch.randn(B, 1, 10, 135, generator=g).cuda()
x_T = torch.rand(B, 1, H, D, generator=g).cuda()
eng.set_builtin_schedule(0, 1000, 1000)
eng.sample_begin(cond, x_T, noise=None, inpaint=None, seed=7)
eng.sample_run(0, 30)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 64)()
assert lib.spdm_debug_tail_stamps(out) == 0
names = ["entry", "o/x loaded, slab written", "product 1 (out-proj)", "tile -> LDS", "LayerNorm, slab", "product 2 (ff1)",
         "GELU, slab", "product 3 (ff2)", "bias + residual + store"]
for base, C in ((0, 128), (16, 256)):
    st = [out[base + i] for i in range(9)]
    print(f"sa_tail_kernel<{C}> (the LAST launch of it in the step), workgroup 0, thread 0: us since entry")
    for i in range(1, 9):
        print(f"  {names[i]:32s} +{(st[i] - st[i - 1]) * 0.01:6.2f}   = {(st[i] - st[0]) * 0.01:6.2f}")

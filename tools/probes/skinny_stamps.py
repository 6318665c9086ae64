#!/usr/bin/env python3
"""Phase timeline of conv_skinny_kernel's workgroup 0 for chosen layers of a real step (diagnostic build only:
   SPDM_EXTRA_FLAGS=-DSPDM_DIAG_SKINNY SPDM_BUILD_OUT=.../libspdm_skinny.so python -m state_policy_diffusionmodel_amd.build).
usage: SPDM_LIB=.../libspdm_skinny.so python tools/probes/skinny_stamps.py [batch]"""
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
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["statistics", "slab staged", "items (MFMA loop)", "barrier after the loop", "partials -> LDS", "sum over waves, store",
         "per-sample totals"]
# (Cin, Cout): the LAST launch of that shape in the step is the one recorded
for K, N, what in ((64, 64, "level 0/1: 64 -> 64"), (128, 128, "up3.dc1 / down2.dc1: 128 -> 128"), (256, 256, "level 1/3: 256 -> 256"),
                   (512, 512, "up1.dc1 / bot: 512 -> 512"), (128, 64, "up3.dc2a: 128 -> 64")):
    assert lib.spdm_debug_skinny_select(K, N) == 0
    eng.sample_begin(cond, x_T, noise=None, inpaint=None, seed=7)
    eng.sample_run(0, 12)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    assert lib.spdm_debug_skinny_stamps(out) == 0
    st = [out[i] for i in range(8)]
    print(f"conv_skinny_kernel, {what} (B = {B}): us per phase, workgroup 0 thread 0")
    for i in range(1, 8):
        print(f"  {names[i - 1]:26s} +{(st[i] - st[i - 1]) * 0.01:6.2f}   = {(st[i] - st[0]) * 0.01:6.2f}")

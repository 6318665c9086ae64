#!/usr/bin/env python3
"""Phase timeline of conv_reg64_kernel (workgroup 0, waves 0 and 4, first three tiles) on inc.b at batch BG_B (diagnostic build only:
   SPDM_EXTRA_FLAGS=-DSPDM_DIAG_REG SPDM_BUILD_OUT=.../libspdm_reg.so python -m state_policy_diffusionmodel_amd.build).
usage: SPDM_LIB=.../libspdm_reg.so BG_B=4096 python tools/probes/reg_stamps.py
(The stamps order memory waits exactly -- each ends an s_waitcnt vmcnt(0) lgkmcnt(0) -- but not VALU work: hipcc moves register
arithmetic across them, so the prologue shows up inside the MFMA phases.  What they showed: the second wave of a SIMD runs 2-3 x
slower than the first while both are resident, and loads behind the previous tile's stores wait for the write acknowledgements.)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from state_policy_diffusionmodel_amd import _lib

lib = _lib.load()
B = int(os.environ.get("BG_B", "4096"))
ms = (ctypes.c_double * 3)()
_lib.check(lib.spdm_bench_gemm(0, B, 32, 8, 64, 64, 9, 2, 0, 1, 3, 0, ms), "spdm_bench_gemm")
print(f"inc.b at B = {B}: {ms[0] * 1e3:.1f} us per launch (diagnostic build: stamps force waits)")
out = (ctypes.c_ulonglong * 48)()
assert lib.spdm_debug_reg_stamps(out) == 0
names = ["raw loads back", "statistics", "prologue k-step 0", "MFMA k-step 0", "prologue k-step 1", "MFMA k-step 1", "epilogue"]
for w in range(2):
    for t in range(3):
        st = [out[(w * 3 + t) * 8 + i] for i in range(8)]
        print(f"wave {4 * w} tile {t}: start at {(st[0] - out[0]) * 0.01:7.2f} us; " +
              "  ".join(f"{n} +{(st[i + 1] - st[i]) * 0.01:.2f}" for i, n in enumerate(names)) + f"  = {(st[7] - st[0]) * 0.01:.2f}")

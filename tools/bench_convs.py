#!/usr/bin/env python3
"""Per-layer micro-benchmark of the 31 3x3 / 3x1 convolutions of one U-Net evaluation (horizon 32, state_dim 3) at batch
BG_B (GPU box; kernel tuning).  Each launch is checked against the exact fp32 kernel on the same data (max |diff|, and the
GroupNorm statistics against totals recomputed from its own output).  usage: BG_B=512 python tools/bench_convs.py [--csv]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from state_policy_diffusionmodel_amd import _lib

lib = _lib.load()
B = int(os.environ.get("BG_B", "512"))
H0 = int(os.environ.get("BG_H", "32"))
LV = [(H0, 8), (H0 // 2, 4), (H0 // 4, 2), (H0 // 8, 1)]


def block(name, lvl, cin, cout, first_pro=0):
    # DoubleConv(cin, cin) then DoubleConv(cin, cout): conv a (input materialised or pending GN), b (GN+GELU), c (GN), d (GN+GELU)
    return [(f"{name}.dc1a", lvl, cin, cin, first_pro), (f"{name}.dc1b", lvl, cin, cin, 2),
            (f"{name}.dc2a", lvl, cin, cout, 1), (f"{name}.dc2b", lvl, cout, cout, 2)]


LAYERS = [("inc.b", 0, 64, 64, 2)]
LAYERS += block("down1", 1, 64, 128) + block("down2", 2, 128, 256) + block("down3", 3, 256, 256)
LAYERS += [("bot1.a", 3, 256, 512, 0), ("bot1.b", 3, 512, 512, 2), ("bot2.a", 3, 512, 512, 1), ("bot2.b", 3, 512, 512, 2),
           ("bot3.a", 3, 512, 256, 1), ("bot3.b", 3, 256, 256, 2)]
LAYERS += block("up1", 2, 512, 128) + block("up2", 1, 256, 64) + block("up3", 0, 128, 64)

only = [a for a in sys.argv[1:] if not a.startswith("--")]
tot = 0.0
for name, lvl, cin, cout, pro in LAYERS:
    if only and not any(name.startswith(o) for o in only):
        continue
    H, W = LV[lvl]
    taps = 3 if W == 1 else 9
    ms = (ctypes.c_double * 3)()
    try:
        _lib.check(lib.spdm_bench_gemm(0, B, H, W, cin, cout, taps, pro, 0, 1, 10, 0, ms), "spdm_bench_gemm")
    except RuntimeError:
        if os.environ.get("SPDM_BENCH_SKIP_ERRORS"):      # (autotune: a forced geometry this layer's kernels do not take)
            continue
        raise
    flops = 2.0 * B * H * W * cin * cout * taps
    tot += ms[0]
    print(f"{name:10s} M={B*H*W:7d} K={cin*taps:5d} N={cout:4d}  {ms[0]*1e3:7.1f} us  {flops/ms[0]/1e9:7.1f} TF  "
          f"max|split-f32|={ms[1]:.1e} stats {ms[2]:.1e}", flush=True)
print(f"total {tot*1e3:.0f} us over the listed layers (B={B})")

#!/usr/bin/env python3
"""Micro-benchmark / ablation of conv_gemm_kernel launch shapes (GPU box).
usage: bench_gemm.py [--f32] [name ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from state_policy_diffusionmodel_amd import _lib

lib = _lib.load()
B = int(os.environ.get("BG_B", "4096"))
SHAPES = {  # name: (H, W, Cin, Cout, taps, pro, epi)
    "up3.dc1a": (32, 8, 128, 128, 9, 0, 0), "up3.dc1b": (32, 8, 128, 128, 9, 2, 0),
    "up3.dc2a": (32, 8, 128, 64, 9, 1, 0), "up3.dc2b": (32, 8, 64, 64, 9, 2, 0),
    "inc.b": (32, 8, 64, 64, 9, 2, 0), "up2.dc2a": (16, 4, 256, 64, 9, 1, 0),
    "up2.dc1a": (16, 4, 256, 256, 9, 0, 0), "up1.dc1a": (8, 2, 512, 512, 9, 0, 0), "up1.dc1b": (8, 2, 512, 512, 9, 2, 0),
    "up1.dc2a": (8, 2, 512, 128, 9, 1, 0), "down2.b": (8, 2, 128, 128, 9, 2, 0), "down2.c": (8, 2, 128, 256, 9, 1, 0),
    "bot2a": (4, 1, 512, 512, 3, 1, 0), "down3": (4, 1, 256, 256, 3, 2, 0),
    "sa6.qkv": (32, 8, 64, 192, 1, 0, 1), "sa6.out": (32, 8, 64, 64, 1, 0, 3), "sa6.ff1": (32, 8, 64, 64, 1, 0, 2),
    "sa1.qkv": (16, 4, 128, 384, 1, 1, 1), "sa1.out": (16, 4, 128, 128, 1, 0, 3), "sa1.ff1": (16, 4, 128, 128, 1, 1, 2),
    "sa2.qkv": (8, 2, 256, 768, 1, 1, 1), "sa2.out": (8, 2, 256, 256, 1, 0, 3), "sa4.ff2": (8, 2, 128, 128, 1, 0, 3),
}
DBG = {"full": 0, "stamp": 128, "pp": 64, "pp+stamp": 192, "noMFMA": 1, "noWload": 2, "noGELU": 4, "noStore": 8, "noAload": 16, "noMFMA+noW": 3, "noW+stamp": 130, "noMFMA+noW+noA+noSt": 27}
split = 0 if "--f32" in sys.argv else 1
names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(SHAPES)
for n in names:
    H, W, Cin, Cout, taps, pro, epi = SHAPES[n]
    flops = 2.0 * B * H * W * Cin * Cout * taps
    out = []
    for dn, dv in DBG.items():
        if dn in ("stamp", "pp", "pp+stamp", "noW+stamp"):
            if "--stamp" not in sys.argv or (dn != "stamp" and os.environ.get("SPDM_STAMP_DUMP")):
                continue
        elif dn != "full" and "--ablate" not in sys.argv:
            continue
        ms = (ctypes.c_double * 3)()
        rc = lib.spdm_bench_gemm(0, B, H, W, Cin, Cout, taps, pro, epi, split, 5, dv, ms)
        _lib.check(rc, "spdm_bench_gemm")
        out.append(f"{dn}={ms[0]*1e3:.0f}us")
        if dn == "full":
            out[-1] += f" ({flops/ms[0]/1e9:.0f} TF, max|split-f32|={ms[1]:.2e}, stats rel {ms[2]:.1e})"
    print(f"{n:10s} M={B*H*W:8d} K={Cin*taps:5d} N={Cout:4d} " + "  ".join(out), flush=True)

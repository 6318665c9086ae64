#!/usr/bin/env python3
"""Exhaustive per-layer search over (m_tile, n_tile, ksplit) for the 31 convolutions of a step at batch BG_B (GPU box).
Every candidate runs in a fresh process (the SPDM_TUNE* overrides are read once per process) through tools/bench_convs.py's
machinery (spdm_bench_gemm: timing + check against the exact fp32 kernel).  Prints, per layer, the default choice's time and
the best candidate.  usage: BG_B=512 python tools/autotune_convs.py"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.environ.get("BG_B", "512")
pat = re.compile(r"(\S+)\s+M=\s*(\d+)\s+K=\s*(\d+)\s+N=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TF\s+max\|split-f32\|=(\S+) stats (\S+)")


def run(env_extra):
    env = dict(os.environ)
    env.update(env_extra)
    env["BG_B"] = B
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_convs.py")], capture_output=True, text=True, env=env, timeout=600)
    out = {}
    for ln in r.stdout.splitlines():
        m = pat.match(ln)
        if m and float(m.group(7)) <= 5e-5 and float(m.group(8)) <= 1e-5:
            out[m.group(1)] = float(m.group(5))
    return out, r.stdout


base, _ = run({})
cands = {}
for mt in (128, 256):
    for nt in (64, 128):
        for ks in (1, 2, 4, 8):
            res, raw = run({"SPDM_TUNE5": str(mt), "SPDM_TUNE6": str(nt), "SPDM_TUNE7": str(ks), "SPDM_BENCH_SKIP_ERRORS": "1"})
            cands[(mt, nt, ks)] = res
            print(f"# candidate m_tile={mt} n_tile={nt} ksplit={ks}: {len(res)} layers ran", flush=True)
tot_b = tot_best = 0.0
for name, t0 in base.items():
    best, cfg = t0, "default"
    for c, res in cands.items():
        if name in res and res[name] < best:
            best, cfg = res[name], c
    tot_b += t0
    tot_best += best
    print(f"{name:10s} default {t0:7.1f} us   best {best:7.1f} us  {cfg}", flush=True)
print(f"total default {tot_b:.0f} us, best-per-layer {tot_best:.0f} us (B={B})")

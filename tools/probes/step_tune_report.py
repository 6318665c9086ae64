#!/usr/bin/env python3
"""Tabulate tools/probes/step_tune.sh output: usage  step_tune_report.py <dir> <KNOB>"""
import glob, json, os, re, sys
d, knob = sys.argv[1], sys.argv[2]
rows = {}
for f in glob.glob(os.path.join(d, f"{knob}_b*_v*_*.json")):
    m = re.search(rf"{knob}_b(\d+)_v(-?\d+)_(\d+)\.json", f)
    try:
        ms = json.load(open(f))["ms_per_step"]
    except Exception:
        continue
    rows.setdefault(int(m.group(1)), {}).setdefault(int(m.group(2)), []).append(ms)
for B in sorted(rows):
    print(f"B={B:5d}: " + "  ".join(f"{v}: {min(t):.3f}" for v, t in sorted(rows[B].items())))

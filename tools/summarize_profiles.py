#!/usr/bin/env python3
"""Turn one profiling sweep (gpurun_out/<dir>/{trace,fetch,write,sq,sq2,grbm} from tools/profile_sweep.sh) into the committed
summaries under profiles/.
usage: summarize_profiles.py gpurun_out/p1 profiles/r03 <tag> [--batch B --horizon H --state-dim D --kind K --no-attention]"""
import argparse
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ap = argparse.ArgumentParser()
ap.add_argument("src"); ap.add_argument("prefix"); ap.add_argument("tag")
ap.add_argument("--batch", type=int, default=4096); ap.add_argument("--horizon", type=int, default=32)
ap.add_argument("--state-dim", type=int, default=3); ap.add_argument("--kind", default="ddpm")
ap.add_argument("--no-attention", action="store_true")
a = ap.parse_args()
src, prefix, tag = a.src, a.prefix, a.tag
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("spdm::", "").replace("void ", "").split("(")[0]


shutil.copy(os.path.join(src, "trace", "run_kernel_stats.csv"), f"{prefix}_bench_{tag}_kernel_stats.csv")
with open(f"{prefix}_step_timeline_{tag}.txt", "w") as fh:
    subprocess.run([sys.executable, os.path.join(here, "trace_step.py"), os.path.join(src, "trace", "run_kernel_trace.csv"), "-v"],
                   stdout=fh, check=True)
if os.path.exists(os.path.join(src, "fetch", "run_counter_collection.csv")):
    extra = ["--batch", str(a.batch), "--horizon", str(a.horizon), "--state-dim", str(a.state_dim), "--kind", a.kind]
    if a.no_attention:
        extra.append("--no-attention")
    subprocess.run([sys.executable, os.path.join(here, "pmc_traffic.py"), os.path.join(src, "fetch", "run_counter_collection.csv"),
                    os.path.join(src, "write", "run_counter_collection.csv"), prefix, tag] + extra, check=True, stdout=subprocess.DEVNULL)
if not os.path.exists(os.path.join(src, "sq", "run_counter_collection.csv")):
    sys.exit(0)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for sub in ("sq", "sq2"):
    path = os.path.join(src, sub, "run_counter_collection.csv")
    if not os.path.exists(path):
        continue
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if "conv" in k or "sa_" in k or "attention" in k:
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "SQ_WAVE_CYCLES":
                cnt[k] += 1
g, gd = collections.defaultdict(float), collections.defaultdict(float)
for r in csv.DictReader(open(os.path.join(src, "grbm", "run_counter_collection.csv"))):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        k = short(r["Kernel_Name"])
        g[k] += float(r["Counter_Value"])
        gd[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"_method": "rocprofv3 --pmc (SQ counters in two passes of 8, GRBM_GUI_ACTIVE in another) over `bench.py --steps 3 --warmup 1`; "
                  "per-launch averages per kernel; clock_GHz_from_GRBM = GRBM_GUI_ACTIVE / 8 XCDs / launch duration (reads high on "
                  "launches shorter than ~0.3 ms); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles); the SQ_* "
                  "cycle counters are in quad-cycles summed over waves; *_per_wave_cycle = counter / SQ_WAVE_CYCLES"}
for k, v in acc.items():
    if not cnt[k]:
        continue
    d = {c: v[c] / cnt[k] for c in v}
    d["launches"] = cnt[k]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_ratio"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM",
                  "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"):
            if c in d:
                d[c + "_per_wave_cycle"] = d[c] / wc
    if k in g:
        d["GRBM_GUI_ACTIVE_per_launch"] = g[k] / cnt[k]
        d["clock_GHz_from_GRBM"] = g[k] / 8 / gd[k]
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE_per_launch"] / 8)
    out[k] = d
json.dump(out, open(f"{prefix}_conv_pmc_summary_{tag}.json", "w"), indent=1)
for k, d in out.items():
    if k != "_method":
        print(f"{k:52s} n={d['launches']:3d} clk {d.get('clock_GHz_from_GRBM', 0):.2f} GHz  mfma busy {d.get('mfma_busy_frac', 0):.2f}  "
              f"wait_any {d.get('SQ_WAIT_ANY_per_wave_cycle', 0):.2f} wait_inst {d.get('SQ_WAIT_INST_ANY_per_wave_cycle', 0):.2f} "
              f"lds_act {d.get('SQ_ACTIVE_INST_LDS_per_wave_cycle', 0):.2f} vmem {d.get('SQ_INST_CYCLES_VMEM_per_wave_cycle', 0):.2f}")

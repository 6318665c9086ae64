#!/usr/bin/env python3
"""Turn one profiling sweep (gpurun_out/<dir>/{trace,fetch,write,sq,grbm} from rocprofv3 over bench.py) into the
committed summaries under profiles/.  usage: summarize_profiles.py gpurun_out/p11 profiles/r01"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

src, prefix = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("spdm::", "").replace("void ", "").split("(")[0]


shutil.copy(os.path.join(src, "trace", "run_kernel_stats.csv"), prefix + "_bench_b4096_kernel_stats_final.csv")
with open(prefix + "_step_timeline_final.txt", "w") as fh:
    subprocess.run([sys.executable, os.path.join(here, "trace_step.py"), os.path.join(src, "trace", "run_kernel_trace.csv"), "-v"],
                   stdout=fh, check=True)
subprocess.run([sys.executable, os.path.join(here, "pmc_traffic.py"), os.path.join(src, "fetch", "run_counter_collection.csv"),
                os.path.join(src, "write", "run_counter_collection.csv"), prefix], check=True, stdout=subprocess.DEVNULL)

acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(os.path.join(src, "sq", "run_counter_collection.csv"))):
    k = short(r["Kernel_Name"])
    if "conv3x3_wide" in k or "conv_gemm_kernel" in k or "sa_" in k or "attention" in k:
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            cnt[k] += 1
g, gd = collections.defaultdict(float), collections.defaultdict(float)
for r in csv.DictReader(open(os.path.join(src, "grbm", "run_counter_collection.csv"))):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        k = short(r["Kernel_Name"])
        g[k] += float(r["Counter_Value"])
        gd[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
out = {"_method": "rocprofv3 --pmc (SQ counters in one pass, GRBM_GUI_ACTIVE in another) over `bench.py --steps 3 --warmup 1`; "
                  "per-launch averages per kernel; clock_GHz_from_GRBM = GRBM_GUI_ACTIVE / 8 XCDs / launch duration (reads high on "
                  "launches shorter than ~0.3 ms); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x launch cycles)"}
for k, v in acc.items():
    d = {c: v[c] / cnt[k] for c in v}
    d["launches"] = cnt[k]
    if d.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_conflict_ratio"] = d.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["SQ_LDS_IDX_ACTIVE"]
    if k in g:
        d["GRBM_GUI_ACTIVE_per_launch"] = g[k] / cnt[k]
        d["clock_GHz_from_GRBM"] = g[k] / 8 / gd[k]
        d["mfma_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * d["GRBM_GUI_ACTIVE_per_launch"] / 8)
    out[k] = d
json.dump(out, open(prefix + "_conv_pmc_summary_final.json", "w"), indent=1)
for k, d in out.items():
    if k != "_method":
        print(f"{k:48s} n={d['launches']:3d} clk {d.get('clock_GHz_from_GRBM', 0):.2f} GHz  mfma busy {d.get('mfma_busy_frac', 0):.2f}")

#!/usr/bin/env python3
"""Print the per-launch timeline of the LAST denoise step in a rocprofv3 kernel_trace.csv."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "conv_in_kernel" in r["Kernel_Name"]]      # first kernel of a denoise step


def step_at(k):
    st = rows[idx[k]:]
    end = next((i for i, r in enumerate(st) if "out_step" in r["Kernel_Name"]), len(st) - 1)
    return st[: end + 1]


# the LAST step whose launches are back to back (a step interrupted by the profiler's own buffer flush shows a gap of a millisecond)
step = step_at(-1)
for k in range(len(idx) - 1, -1, -1):
    cand = step_at(k)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in cand)
    if int(cand[-1]["End_Timestamp"]) - int(cand[0]["Start_Timestamp"]) <= 1.1 * busy:
        step = cand
        break
tot = 0.0
agg = {}
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("spdm::", "").replace("void ", "").split("(")[0][:62]
    agg[n] = agg.get(n, 0.0) + d
    if "-v" in sys.argv:
        print(f"{n:64s} grid={int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']):7d} vgpr={r['VGPR_Count']:>4s}+{r['Accum_VGPR_Count']:>3s} {d:9.1f} us")
print(f"sum of kernels {tot/1e3:.2f} ms, wall {(int(step[-1]['End_Timestamp'])-int(step[0]['Start_Timestamp']))/1e6:.2f} ms")
for n, d in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"  {n:64s} {d/1e3:8.3f} ms {100*d/tot:5.1f}%")

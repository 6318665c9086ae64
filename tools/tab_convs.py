#!/usr/bin/env python3
"""Tabulate tools/bench_convs.py outputs side by side: tab_convs.py file1 file2 ... (us per layer)."""
import re
import sys
cols = []
for f in sys.argv[1:]:
    d = {}
    for ln in open(f):
        m = re.match(r"(\S+)\s+M=\s*(\d+)\s+K=\s*(\d+)\s+N=\s*(\d+)\s+([\d.]+) us\s+([\d.]+) TF\s+max\|split-f32\|=(\S+) stats (\S+)", ln)
        if m:
            d[m.group(1)] = (int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), float(m.group(7)), float(m.group(8)))
    cols.append(d)
names = list(cols[0].keys())
print(f"{'layer':10s} {'M':>7s} {'K':>5s} {'N':>4s} " + " ".join(f"{f.split('/')[-1].replace('convs_','').replace('.txt','')[:12]:>12s}" for f in sys.argv[1:]) + "   worst maxdiff / stats")
tot = [0.0] * len(cols)
for n in names:
    M, K, N = cols[0][n][:3]
    row = []
    for i, c in enumerate(cols):
        v = c.get(n)
        row.append(f"{v[3]:12.1f}" if v else " " * 12)
        if v:
            tot[i] += v[3]
    wd = max(c[n][4] for c in cols if n in c)
    ws = max(c[n][5] for c in cols if n in c)
    print(f"{n:10s} {M:7d} {K:5d} {N:4d} " + " ".join(row) + f"   {wd:.1e} {ws:.1e}")
print(f"{'total':29s} " + " ".join(f"{t:12.0f}" for t in tot))

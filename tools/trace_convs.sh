#!/bin/bash
# rocprofv3 kernel trace of tools/bench_convs.py (per-kernel durations of conv vs split-K combine)
out="$(realpath -m "$1")"; repo="$(pwd)"; shift
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$repo/tools/bench_convs.py" "$@" > "$out/log.txt" 2>&1
python3 - "$out/run_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("spdm::", "").replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{n:72s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY

#!/bin/bash
# kernel trace of one bench.py configuration + last-step timeline.  usage: bash tools/trace_bench.sh <outdir> [bench args]
out="$(realpath -m "$1")"; repo="$(pwd)"; shift
mkdir -p "$out"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o run -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/log.txt" 2>&1
python3 "$repo/tools/trace_step.py" "$out/run_kernel_trace.csv" -v > "$out/timeline.txt"
tail -30 "$out/timeline.txt"

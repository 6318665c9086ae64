#!/bin/bash
# One profiling sweep over bench.py on the GPU box: kernel trace + stats, then the PMC passes -- each in its own run, never
# combined with a trace domain.
#   usage (repo root, under gpurun):  bash tools/profile_sweep.sh <outdir> <all|trace|traffic> [bench.py args ...]
#   then, back in the container:      python tools/summarize_profiles.py <outdir> profiles/r03 <tag> [--batch .. --horizon .. --state-dim .. --kind ..]
# passes: trace = kernel trace only; traffic = trace + FETCH_SIZE + WRITE_SIZE; all = traffic + two SQ passes + GRBM.
set -e
out="$(realpath -m "$1")"; what="$2"; shift 2; repo="$(pwd)"; BENCH_ARGS="$*"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$out/trace.log" 2>&1
echo "pass trace done"
[ "$what" = "trace" ] && exit 0
passes=("fetch FETCH_SIZE" "write WRITE_SIZE")
if [ "$what" = "all" ]; then
  passes+=("sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
           "sq2 SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
           "grbm GRBM_GUI_ACTIVE")
fi
for pass in "${passes[@]}"; do
    set -- $pass; name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o run -- python3 "$repo/bench.py" --steps 3 --warmup 1 --no-cpu-baseline $BENCH_ARGS > "$out/$name.log" 2>&1 || echo "pass $name FAILED (see $out/$name.log)"
    echo "pass $name done"
done

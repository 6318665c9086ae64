#!/bin/bash
# One profiling sweep over bench.py on the GPU box: kernel trace + stats, then the PMC passes (each in its own run, never
# combined with a trace domain).  usage (from the repo root, under gpurun):  bash tools/profile_sweep.sh gpurun_out/p20 [trace-only]
# then, back in the container:  python tools/summarize_profiles.py gpurun_out/p20 profiles/r01
set -e
out="$(realpath -m "$1")"; repo="$(pwd)"
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 "$repo/bench.py" --steps 3 --warmup 1 > "$out/trace.log" 2>&1
[ "$2" = "trace-only" ] && exit 0
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "grbm GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -o run -- python3 "$repo/bench.py" --steps 3 --warmup 1 > "$out/$name.log" 2>&1
    echo "pass $name done"
done

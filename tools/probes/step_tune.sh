# whole-step timing (graph replay, bench.py) of one tuning knob: usage  step_tune.sh <out> <KNOB> "<values>" "<batches>"
set -e
out=gpurun_out/${1:-st}; knob=$2; mkdir -p $out
for rep in 1 2; do
for B in $4; do
  for v in $3; do
    env $knob=$v timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/${knob}_b${B}_v${v}_$rep.json 2>/dev/null
  done
done
done

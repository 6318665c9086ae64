# same-box A/B of one kernel-selection switch over a few batches (graph-replay step time), alternating: usage ab_switch.sh <out> <SWITCH=1 | assignment> "<batches>"
set -e
out=gpurun_out/${1:-absw}; mkdir -p $out
for rep in 1 2 3; do
  for B in $3; do
    timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/base_b${B}_$rep.json 2>/dev/null
    env $2 timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/sw_b${B}_$rep.json 2>/dev/null
  done
done
python3 - "$out" "$3" <<'PY'
import json, sys, glob
out, batches = sys.argv[1], sys.argv[2].split()
for B in batches:
    r = {}
    for k in ("base", "sw"):
        r[k] = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"{out}/{k}_b{B}_*.json"))]
    print(f"B={B:>5s}  default {min(r['base']):.4f} ms (runs {', '.join('%.4f' % v for v in r['base'])})   switch on {min(r['sw']):.4f} ms (runs {', '.join('%.4f' % v for v in r['sw'])})")
PY

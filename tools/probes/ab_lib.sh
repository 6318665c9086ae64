# same-box A/B of a second build of the library (state_policy_diffusionmodel_amd/libspdm_prev.so) against the in-tree one, alternating:
# usage ab_lib.sh <out> "<batches>"
set -e
out=gpurun_out/${1:-ablib}; mkdir -p $out
P=$PWD/state_policy_diffusionmodel_amd/libspdm_prev.so
for rep in 1 2 3; do
  for B in $2; do
    timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/base_b${B}_$rep.json 2>/dev/null
    SPDM_LIB=$P timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/sw_b${B}_$rep.json 2>/dev/null
  done
done
python3 - "$out" "$2" <<'PY'
import json, sys, glob
out, batches = sys.argv[1], sys.argv[2].split()
for B in batches:
    r = {}
    for k in ("base", "sw"):
        r[k] = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob(f"{out}/{k}_b{B}_*.json"))]
    print(f"B={B:>5s}  in-tree {min(r['base']):.4f} ms (runs {', '.join('%.4f' % v for v in r['base'])})   other build {min(r['sw']):.4f} ms (runs {', '.join('%.4f' % v for v in r['sw'])})")
PY

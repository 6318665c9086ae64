# whole-step timing (graph replay) with each kernel-selection switch flipped, one at a time: usage step_switches.sh <out> "<batches>"
set -e
out=gpurun_out/${1:-sw}; mkdir -p $out
for rep in 1 2; do
for B in $2; do
  timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/base_b${B}_$rep.json 2>/dev/null
  for sw in SPDM_NO_WIDE128 SPDM_NO_W2 SPDM_T3_BIG SPDM_NO_SMALL_TPI3 SPDM_WIDE_N64_2X2 SPDM_NO_WIDE_PIPE SPDM_DEEP SPDM_NO_SKINNY SPDM_NO_T512 SPDM_T512 SPDM_ATTN_VALU SPDM_SA_NO_WLDS SPDM_NO_SA_TAIL; do
    env $sw=1 timeout -k 10 120 python bench.py --batch $B --steps 40 --warmup 5 --no-cpu-baseline > $out/${sw}_b${B}_$rep.json 2>/dev/null
  done
done
done

# whole-step timing (graph replay, bench.py) of conv_skinny's column-tile thresholds -- the per-layer micro-benchmark re-reads hot
# weights and overstates what narrow tiles gain; SPDM_TUNE14 = 16-wide below, SPDM_TUNE11 = 32-wide below
set -e
out=gpurun_out/${1:-st}; mkdir -p $out
for rep in 1 2; do
for B in ${BATCHES:-8 32 64 128 256 512}; do
  for cfg in "0 128" "0 192" "0 256" "0 257"; do
    set -- $cfg
    SPDM_TUNE14=$1 SPDM_TUNE11=$2 timeout -k 10 120 python bench.py --batch $B --steps 50 --warmup 5 --no-cpu-baseline > $out/b${B}_t14_$1_t11_$2_$rep.json 2>/dev/null
  done
done
done

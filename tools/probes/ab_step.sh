# same-box A/B of two builds of the library over the batch sweep (graph-replay step time): SPDM_LIB=<prev> vs the in-tree build
set -e
out=gpurun_out/${1:-ab}; mkdir -p $out
P=$PWD/state_policy_diffusionmodel_amd/libspdm_prev.so
for rep in 1 2; do
  for B in ${BATCHES:-1 2 4 8 16 32 64 128 256 512}; do
    SPDM_LIB=$P timeout -k 10 120 python bench.py --batch $B --steps 50 --warmup 5 --no-cpu-baseline > $out/prev_b${B}_$rep.json 2>/dev/null
    timeout -k 10 120 python bench.py --batch $B --steps 50 --warmup 5 --no-cpu-baseline > $out/new_b${B}_$rep.json 2>/dev/null
  done
done

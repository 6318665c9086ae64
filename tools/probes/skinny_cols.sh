# conv_skinny 16-wide tiles (kernel tuning): all 31 layers per batch with the 16-column threshold (SPDM_TUNE14) moved
set -e
out=gpurun_out/${1:-sc}; mkdir -p $out
for B in 1 4 8 16 32 64 128; do
  for t in 0 64 96 128 192; do
    SPDM_TUNE14=$t SPDM_TUNE11=$(( t > 128 ? t : 128 )) BG_B=$B timeout -k 10 120 python tools/bench_convs.py > $out/b${B}_t$t.txt 2>&1
  done
done

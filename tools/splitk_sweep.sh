#!/bin/bash
# split-K tuning sweep on the GPU box: per-layer conv times at several batches, split-K off / on at several grid targets
out=gpurun_out/r2b; mkdir -p $out
for b in 1 8 256 512; do
  SPDM_NO_SPLITK=1 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_nosplit.txt 2>&1 || exit 1
  for t in 128 256 512; do
    SPDM_TUNE0=$t BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_t${t}.txt 2>&1 || exit 1
  done
  SPDM_TUNE0=256 SPDM_TUNE1=0 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_t256_n64.txt 2>&1 || exit 1
  echo "B=$b: $(tail -1 $out/convs_b${b}_nosplit.txt) | t128 $(tail -1 $out/convs_b${b}_t128.txt) | t256 $(tail -1 $out/convs_b${b}_t256.txt) | t512 $(tail -1 $out/convs_b${b}_t512.txt) | t256/n64 $(tail -1 $out/convs_b${b}_t256_n64.txt)"
done

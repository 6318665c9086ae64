#!/bin/bash
out=gpurun_out/r2c; mkdir -p $out
for b in 256 512 1024; do
  SPDM_TUNE1=0 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_n64.txt 2>&1 || exit 1
  SPDM_TUNE1=0 SPDM_TUNE2=1 SPDM_TUNE3=2 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_big_s2.txt 2>&1 || exit 1
  SPDM_TUNE1=0 SPDM_TUNE2=1 SPDM_TUNE3=4 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_big_s4.txt 2>&1 || exit 1
  SPDM_TUNE1=0 SPDM_TUNE2=1 SPDM_TUNE3=4 SPDM_TUNE0=512 BG_B=$b timeout -k 10 200 python tools/bench_convs.py > $out/convs_b${b}_big_s4_t512.txt 2>&1 || exit 1
  python tools/tab_convs.py $out/convs_b${b}_n64.txt $out/convs_b${b}_big_s2.txt $out/convs_b${b}_big_s4.txt $out/convs_b${b}_big_s4_t512.txt
done

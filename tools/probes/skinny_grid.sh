# conv_skinny grid limits (kernel tuning): all 31 layers per batch with the 16 / 32 / 64-row grid limits moved (SPDM_TUNE8 / 12 / 13)
set -e
out=gpurun_out/${1:-sg}; mkdir -p $out
for B in 32 64 128 256 512 1024; do
  BG_B=$B timeout -k 10 120 python tools/bench_convs.py > $out/b${B}_base.txt 2>&1
  for g in 384 512; do
    SPDM_TUNE8=$g BG_B=$B timeout -k 10 120 python tools/bench_convs.py > $out/b${B}_g16_$g.txt 2>&1
    SPDM_TUNE12=$g BG_B=$B timeout -k 10 120 python tools/bench_convs.py > $out/b${B}_g32_$g.txt 2>&1
    SPDM_TUNE13=$g SPDM_TUNE9=250 BG_B=$B timeout -k 10 120 python tools/bench_convs.py > $out/b${B}_g64_$g.txt 2>&1
  done
done

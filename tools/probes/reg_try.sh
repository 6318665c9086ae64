set -e
mkdir -p gpurun_out/r13
for B in 4096 1024 512 256 128; do
  for thr in 9999999 1; do
    echo "== B=$B threshold $thr" >> gpurun_out/r13/convs.txt
    SPDM_TUNE19=$thr BG_B=$B timeout -k 10 120 python tools/bench_convs.py up3.dc2a inc.b 2>&1 | grep -v amdgpu.ids >> gpurun_out/r13/convs.txt
  done
done
grep -v "^total" gpurun_out/r13/convs.txt

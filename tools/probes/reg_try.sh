set -e
mkdir -p gpurun_out/r6
for B in 4096 1024 512; do
  echo "== B=$B" >> gpurun_out/r6/convs.txt
  BG_B=$B timeout -k 10 120 python tools/bench_convs.py inc.b up3.dc2b 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6/convs.txt
done
echo "== B=4096 stagger 0" >> gpurun_out/r6/convs.txt
SPDM_TUNE18=0 BG_B=4096 timeout -k 10 120 python tools/bench_convs.py inc.b up3.dc2b 2>&1 | grep -v amdgpu.ids >> gpurun_out/r6/convs.txt
cat gpurun_out/r6/convs.txt

set -e
mkdir -p gpurun_out/r10
for B in 8 16 32; do
  for thr in 9999999 1; do
    echo "== B=$B threshold $thr" >> gpurun_out/r10/convs.txt
    SPDM_TUNE17=$thr BG_B=$B timeout -k 10 120 python tools/bench_convs.py inc.b up3.dc2b 2>&1 | grep -v amdgpu.ids >> gpurun_out/r10/convs.txt
  done
done
for B in 64 128 256 512 1024 2048; do
  for thr in 9999999 1; do
    echo "== B=$B threshold $thr" >> gpurun_out/r10/convs.txt
    SPDM_TUNE17=$thr BG_B=$B timeout -k 10 120 python tools/bench_convs.py down1.dc1 up2.dc2b 2>&1 | grep -v amdgpu.ids >> gpurun_out/r10/convs.txt
  done
done
grep -v "^total" gpurun_out/r10/convs.txt | awk '{ if ($1=="==") print; else print "   ", $1, $2, $6, $7 }'

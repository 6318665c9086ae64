# conv_skinny tile rules (kernel tuning): per-layer times under the shipped rule and with the 32-column threshold moved
# (SPDM_TUNE11: 32-wide tiles while m-tiles x N/64 is under it); earlier form of this probe forced the row tile (profiles/r02_skinny_rows.txt)
set -e
mkdir -p gpurun_out/r8
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r8/pytest.log 2>&1
for B in 1 8 64 256; do
  for t in 32 64 128 257; do
    SPDM_TUNE11=$t BG_B=$B timeout -k 10 120 python tools/bench_convs.py > gpurun_out/r8/b${B}_nt$t.txt 2>&1
  done
done
for B in 1 2 4 8 16 32 64 128 256 512 1024; do
  timeout -k 10 120 python bench.py --batch $B --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r8/bench_b$B.json 2>/dev/null
done
tail -2 gpurun_out/r8/pytest.log

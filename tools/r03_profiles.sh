#!/bin/bash
# Round-3 profiling passes (run on the GPU box; ~15 GPU-minutes).  usage: bash tools/r03_profiles.sh gpurun_out/p1
out="$1"; mkdir -p "$out"
rocprofv3 -L > "$out/counters.txt" 2>&1 || true
bash tools/profile_sweep.sh "$out/b4096_h32d3" all
bash tools/profile_sweep.sh "$out/c5_b4096_h64d6_ddim" traffic --kind ddim --train-steps 50 --horizon 64 --state-dim 6 --batch 4096
bash tools/profile_sweep.sh "$out/c3_b1024_ddim" trace --kind ddim --train-steps 50 --batch 1024
bash tools/profile_sweep.sh "$out/b512_h32d3" trace --batch 512
bash tools/profile_sweep.sh "$out/b256_h32d3" trace --batch 256
bash tools/profile_sweep.sh "$out/b64_h32d3" trace --batch 64
bash tools/profile_sweep.sh "$out/b1_h32d3" trace --batch 1
# rehearsal of the N > 1 launcher with the real engine on this one GPU (gloo rendezvous, both ranks on cuda:0)
python bench.py --gpus 2 --shared-device --backend gloo --steps 10 --warmup 3 --batch 512 --global-batch 1024 > "$out/rehearsal_2ranks_shared_device.json" 2> "$out/rehearsal.err" || echo "rehearsal FAILED"
ls "$out"

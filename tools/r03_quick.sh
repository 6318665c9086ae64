#!/bin/bash
# Round-3 bench lines: headline (with the CPU leg), configs 3 and 5, the batch sweep, whole-call latency of the closed-loop
# caller.  usage (repo root, under gpurun):  bash tools/r03_quick.sh gpurun_out/q1
out="$1"; mkdir -p "$out"
python bench.py > "$out/bench_line.json" 2> "$out/bench_line.err"
python bench.py --kind ddim --train-steps 50 --batch 1024 --no-cpu-baseline > "$out/config3_ddim_b1024.json" 2>> "$out/err.txt"
python bench.py --kind ddim --train-steps 50 --horizon 64 --state-dim 6 --batch 4096 --no-cpu-baseline > "$out/config5_ddim_b4096_h64d6.json" 2>> "$out/err.txt"
python bench.py --kind ddim --train-steps 50 --horizon 64 --state-dim 6 --batch 512 --no-cpu-baseline > "$out/config5_per_rank_ddim_b512_h64d6.json" 2>> "$out/err.txt"
python bench.py --batch 256 --no-cpu-baseline > "$out/config2_b256.json" 2>> "$out/err.txt"
: > "$out/batch_sweep.jsonl"
for b in 1 2 4 8 16 32 64 128 256 512 1024 2048 4096; do
  python bench.py --batch $b --no-cpu-baseline >> "$out/batch_sweep.jsonl" 2>> "$out/err.txt"
done
FULL_B=1 FULL_KIND=ddim FULL_N=100 python tools/full_call.py > "$out/full_call_b1_ddim100.json" 2>> "$out/err.txt"
FULL_B=1 FULL_KIND=ddpm FULL_N=1000 python tools/full_call.py > "$out/full_call_b1_ddpm1000.json" 2>> "$out/err.txt"
python - "$out" <<'PY'
import json, sys, glob, os
o = sys.argv[1]
for f in sorted(glob.glob(os.path.join(o, "*.json"))):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "UNREADABLE", e); continue
    if "ms_per_step" in d:
        print(os.path.basename(f), round(d["ms_per_step"], 4), round(d.get("graph_replay", {}).get("ms_per_step", 0), 4), round(d["value"]))
    else:
        print(os.path.basename(f), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d.items() if k != "workload"})
for ln in open(os.path.join(o, "batch_sweep.jsonl")):
    d = json.loads(ln); print(d["config"]["per_gpu_batch"], round(d["ms_per_step"], 4), round(d["graph_replay"]["ms_per_step"], 4))
PY

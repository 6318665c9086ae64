#!/bin/bash
# Turn one profiling sweep (tools/r03_profiles.sh -> gpurun_out/<p>) and one set of bench lines (tools/r03_quick.sh -> gpurun_out/<q>)
# into the committed summaries under profiles/.  usage (container, repo root): bash tools/r03_collect.sh gpurun_out/p6 gpurun_out/q10
set -e
S="$1"; Q="$2"
python tools/summarize_profiles.py $S/b4096_h32d3 profiles/r03 b4096_h32d3
python tools/summarize_profiles.py $S/c5_b4096_h64d6_ddim profiles/r03 c5_b4096_h64d6_ddim --batch 4096 --horizon 64 --state-dim 6 --kind ddim
python tools/summarize_profiles.py $S/c3_b1024_ddim profiles/r03 c3_b1024_h32d3_ddim --batch 1024 --kind ddim
for b in 512 256 64 1; do python tools/summarize_profiles.py $S/b${b}_h32d3 profiles/r03 b${b}_h32d3 --batch $b; done
cp $S/rehearsal_2ranks_shared_device.json profiles/r03_rehearsal_2ranks_shared_device.json
cp $Q/bench_line.json profiles/r03_bench_line.json
cp $Q/batch_sweep.jsonl profiles/r03_batch_sweep.jsonl
cat $Q/config2_b256.json $Q/config3_ddim_b1024.json $Q/config5_ddim_b4096_h64d6.json $Q/config5_per_rank_ddim_b512_h64d6.json > profiles/r03_other_configs.jsonl
python3 - "$Q" <<'PY'
import json, sys
q = sys.argv[1]
d = {"ddim_100_generate_py_default": json.load(open(f"{q}/full_call_b1_ddim100.json")),
     "ddpm_1000_run_predictions_py": json.load(open(f"{q}/full_call_b1_ddpm1000.json"))}
json.dump(d, open("profiles/r03_full_call_b1.json", "w"), indent=1)
PY

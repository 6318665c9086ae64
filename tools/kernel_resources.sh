#!/bin/bash
# registers / scratch / LDS of every kernel in one source file.  usage: tools/kernel_resources.sh conv_wide.hip [extra flags]
src="state_policy_diffusionmodel_amd/csrc/$1"; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o /tmp/kr.s "$src" "$@" 2>/dev/null
python3 - <<'PY'
import re
txt = open("/tmp/kr.s").read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    def g(k):
        r = re.search(r"\.amdhsa_" + k + r"\s+(\S+)", body)
        return r.group(1) if r else "?"
    import subprocess
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace("spdm::", "").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{dem[:72]:72s} next_free_vgpr {g('next_free_vgpr'):>4s} accum_offset {g('accum_offset'):>4s} scratch {g('private_segment_fixed_size'):>5s} sgpr {g('next_free_sgpr'):>4s}")
PY

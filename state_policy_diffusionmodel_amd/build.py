"""Build driver for libspdm_hip.so (hipcc, gfx950 only; cross-compiles without a GPU).

    python -m state_policy_diffusionmodel_amd.build [--force]

The shared library is built IN-TREE (state_policy_diffusionmodel_amd/libspdm_hip.so) so that it
travels with the repo snapshot to the GPU box; it is git-ignored (*.so).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("SPDM_BUILD_OUT") or os.path.join(HERE, "libspdm_hip.so")      # (SPDM_BUILD_OUT: a second build for A/B timing)
OBJ = os.path.join(HERE, "csrc", "_obj" + ("_" + os.path.basename(LIB).replace(".", "_") if os.environ.get("SPDM_BUILD_OUT") else ""))
SOURCES = ["conv_gemm.hip", "conv_wide.hip", "conv_skinny.hip", "conv_reg.hip", "elementwise.hip", "attention.hip", "sa_fused.hip", "sa_tail.hip", "encoder.hip", "spdm_api.hip"]
HEADERS = ["kernels.h", "device_utils.h", os.path.join("..", "..", "include", "spdm.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
if os.environ.get("SPDM_DIAG") == "1":      # diagnostic build: in-kernel s_memtime stamps + experimental schedules
    FLAGS.append("-DSPDM_DIAG")
FLAGS += os.environ.get("SPDM_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X build needs ROCm's hipcc")


def _digest() -> str:
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = False) -> str:
    stamp = LIB + ".stamp"
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # Link against the HIP runtime that PyTorch-ROCm itself loads (torch/lib/libamdhip64.so), NOT the
    # system /opt/rocm one hipcc would add: device pointers and hipStream_t handles cross the C ABI from
    # torch, so both sides must live in the same runtime instance (same rule torch's cpp_extension follows).
    import torch
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    if not os.path.exists(os.path.join(tlib, "libamdhip64.so")):
        raise RuntimeError(f"{tlib}/libamdhip64.so not found: this build expects PyTorch-ROCm")
    cxx = shutil.which("g++") or shutil.which("c++")
    cmd = [cxx, "-shared", "-fPIC", "-o", LIB, *objs, f"-L{tlib}", "-lamdhip64", f"-Wl,-rpath,{tlib}",
           "-Wl,--no-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

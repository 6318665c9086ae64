"""ctypes binding of libspdm_hip.so (include/spdm.h).  Fails loudly: there is no CPU
fallback and no alternative backend -- if the HIP library is missing or does not export
the ABI, importing the product path raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int32, c_int64, c_size_t, c_uint64, c_void_p

from .weights import TensorIndex

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPDM_LIB") or os.path.join(HERE, "libspdm_hip.so")     # (SPDM_LIB: A/B builds during kernel tuning)

SPDM_DDPM, SPDM_DDIM = 0, 1
SPDM_FLAG_DEBUG_KEEP = 1
SPDM_FLAG_EXACT_FP32 = 2
ABI_VERSION = 1


class SpdmConfig(ctypes.Structure):
    _fields_ = [(n, c_int32) for n in ("horizon", "state_dim", "cond_dim", "time_dim", "attention", "max_batch",
                                        "device", "num_train_timesteps", "flags")]


# every symbol include/spdm.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "spdm_abi_version": (c_int32, []),
    "spdm_last_error": (c_char_p, []),
    "spdm_create": (c_int32, [POINTER(SpdmConfig), POINTER(c_void_p)]),
    "spdm_destroy": (None, [c_void_p]),
    "spdm_load_weights": (c_int32, [c_void_p, c_void_p, c_size_t, POINTER(TensorIndex), c_int32]),
    "spdm_set_time_table": (c_int32, [c_void_p, c_void_p, c_int32]),
    "spdm_set_schedule": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_float, c_float]),
    "spdm_set_schedule_tables": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "spdm_schedule_tables": (c_int32, [c_int32, c_int32, c_int32, c_float, c_float, c_void_p, c_void_p]),
    "spdm_unet_forward": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "spdm_sample": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_uint64,
                              c_uint64, c_void_p, c_void_p, c_void_p]),
    "spdm_sample_begin": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p,
                                    c_uint64, c_uint64, c_void_p, c_void_p]),
    "spdm_sample_run": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "spdm_sample_result": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "spdm_graph_captures": (c_int64, [c_void_p]),
    "spdm_debug_tensor": (c_int32, [c_void_p, c_char_p, c_void_p, c_size_t, POINTER(c_int32 * 4)]),
    "spdm_uses_split_precision": (c_int32, [c_void_p]),
    "spdm_demoted_tensors": (c_int32, [c_void_p]),
    "spdm_nonfinite": (c_int32, [c_void_p, POINTER(c_int32), c_void_p]),
    "spdm_set_switch": (c_int32, [c_void_p, c_char_p, c_int32]),
    "spdm_device_bytes": (c_size_t, [c_void_p]),
    "spdm_profile_enable": (c_int32, [c_void_p, c_int32]),
    "spdm_profile_read": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_double), POINTER(c_double)]),
    "spdm_encoder_create": (c_int32, [c_int32, c_void_p, c_size_t, POINTER(TensorIndex), c_int32, POINTER(c_void_p)]),
    "spdm_encoder_forward": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "spdm_encoder_destroy": (None, [c_void_p]),
    "spdm_op_gelu": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "spdm_debug_geometry": (c_int32, [c_int32] * 6 + [ctypes.c_uint32, POINTER(c_int32 * 10)]),
    "spdm_bench_gemm": (c_int32, [c_int32] * 12 + [POINTER(c_double)]),   # ms_out[2]: {ms per launch, max|split - fp32|}
}

_lib = None


def load() -> ctypes.CDLL:
    """dlopen the library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run `python -m state_policy_diffusionmodel_amd.build`); there is no CPU fallback")
    import torch  # noqa: F401  -- FIRST: the library binds to the HIP runtime instance torch has loaded
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.spdm_abi_version() != ABI_VERSION:
        raise RuntimeError(f"ABI version mismatch: library {lib.spdm_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().spdm_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

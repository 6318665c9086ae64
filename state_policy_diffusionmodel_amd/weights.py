"""Parameter inventory of the FiLM U-Net noise predictor, a deterministic
random-init generator, and the packer that turns a ``state_dict`` into the
flat blob + tensor index that ``spdm_load_weights`` (include/spdm.h) takes.

The tensor NAMES and SHAPES are exactly those of the reference module's
``state_dict()`` (``models/Unet_FiLmLayer.py:240-264`` builds the module tree;
162 tensors / 24 823 297 parameters at ``global_cond_dim=1350``), so a real
checkpoint's ``noise_estimator.*`` entries can be fed to the same packer.
``tests/test_oracle_vs_reference.py`` loads a generated dict into the imported
reference module with ``strict=True`` to pin the inventory key-for-key.
"""
from __future__ import annotations

import ctypes
import math
from collections import OrderedDict
from typing import Dict, List, Tuple

import numpy as np

NAME_MAX = 64  # must match SPDM_NAME_MAX in include/spdm.h


def _double_conv(prefix: str, cin: int, cout: int) -> List[Tuple[str, Tuple[int, ...]]]:
    # reference: DoubleConvolution.__init__, models/Unet_FiLmLayer.py:93-105
    return [
        (f"{prefix}.first.weight", (cout, cin, 3, 3)),
        (f"{prefix}.second.weight", (cout, cout, 3, 3)),
        (f"{prefix}.norm.weight", (cout,)),
        (f"{prefix}.norm.bias", (cout,)),
    ]


def _resample_block(prefix: str, cin: int, cout: int, time_dim: int, cond_dim: int):
    # reference: DownSample/UpSample.__init__, models/Unet_FiLmLayer.py:129-154, 187-213
    out = _double_conv(f"{prefix}.doubleConv1", cin, cin)
    out += _double_conv(f"{prefix}.doubleConv2", cin, cout)
    out += [
        (f"{prefix}.emb_layer.1.weight", (cout, time_dim)),
        (f"{prefix}.emb_layer.1.bias", (cout,)),
        (f"{prefix}.cond_encoder.2.weight", (2 * cout, cond_dim)),
        (f"{prefix}.cond_encoder.2.bias", (2 * cout,)),
    ]
    return out


def _self_attention(prefix: str, c: int):
    # reference: SelfAttention.__init__, models/Unet_FiLmLayer.py:56-67
    return [
        (f"{prefix}.attention.in_proj_weight", (3 * c, c)),
        (f"{prefix}.attention.in_proj_bias", (3 * c,)),
        (f"{prefix}.attention.out_proj.weight", (c, c)),
        (f"{prefix}.attention.out_proj.bias", (c,)),
        (f"{prefix}.ln.weight", (c,)),
        (f"{prefix}.ln.bias", (c,)),
        (f"{prefix}.ff_self.0.weight", (c,)),
        (f"{prefix}.ff_self.0.bias", (c,)),
        (f"{prefix}.ff_self.1.weight", (c, c)),
        (f"{prefix}.ff_self.1.bias", (c,)),
        (f"{prefix}.ff_self.3.weight", (c, c)),
        (f"{prefix}.ff_self.3.bias", (c,)),
    ]


def unet_film_param_spec(global_cond_dim: int, time_dim: int = 256,
                         attention: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered name -> shape map, same order as the reference ``state_dict()``
    (``UNet_Film.__init__``, models/Unet_FiLmLayer.py:246-264; with
    ``attention=False`` the six ``sa*`` blocks drop out, which is the whole
    difference to ``models/Unet_FiLmLayer_noAttention.py:240-300``)."""
    spec: List[Tuple[str, Tuple[int, ...]]] = []
    spec += _double_conv("inc", 1, 64)
    spec += _resample_block("down1", 64, 128, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa1", 128)
    spec += _resample_block("down2", 128, 256, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa2", 256)
    spec += _resample_block("down3", 256, 256, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa3", 256)
    spec += _double_conv("bot1", 256, 512)
    spec += _double_conv("bot2", 512, 512)
    spec += _double_conv("bot3", 512, 256)
    spec += _resample_block("up1", 512, 128, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa4", 128)
    spec += _resample_block("up2", 256, 64, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa5", 64)
    spec += _resample_block("up3", 128, 64, time_dim, global_cond_dim)
    if attention:
        spec += _self_attention("sa6", 64)
    spec += [("outc.weight", (1, 64, 1, 1)), ("outc.bias", (1,))]
    return OrderedDict(spec)


def random_state_dict(global_cond_dim: int, seed: int = 0, time_dim: int = 256,
                      attention: bool = True) -> "OrderedDict[str, np.ndarray]":
    """Deterministic random-init weights (numpy PCG64, independent of torch's
    initialisers so the GPU box regenerates the identical 99 MB blob from the
    seed). Scale follows torch's defaults (U(+-1/sqrt(fan_in)) for conv/linear
    weights and biases); norm gains/offsets and the attention biases are made
    non-trivial on purpose so that a kernel which drops one of them fails
    parity instead of passing on ones/zeros."""
    rng = np.random.default_rng(seed)
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in unet_film_param_spec(global_cond_dim, time_dim, attention).items():
        is_norm = (".norm." in name) or (".ln." in name) or (".ff_self.0." in name)
        if is_norm and name.endswith("weight"):
            w = 1.0 + 0.1 * rng.uniform(-1.0, 1.0, size=shape)
        elif is_norm:
            w = 0.1 * rng.uniform(-1.0, 1.0, size=shape)
        elif len(shape) >= 2:
            fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-bound, bound, size=shape)
        else:  # bias of a conv / linear layer: fan_in of its weight
            n = shape[0]
            if name.endswith("in_proj_bias"):
                fan_in = n // 3
            elif name.startswith("outc"):
                fan_in = 64
            elif ".cond_encoder." in name:
                fan_in = global_cond_dim
            elif ".emb_layer." in name:
                fan_in = time_dim
            else:
                fan_in = n
            bound = 1.0 / math.sqrt(fan_in)
            w = rng.uniform(-bound, bound, size=shape)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def state_dict_to_numpy(sd) -> "OrderedDict[str, np.ndarray]":
    """Accepts a dict of numpy arrays or torch tensors; strips an optional
    ``noise_estimator.`` prefix (Lightning checkpoint layout,
    ``models/diffusion_ddpm.py:76``)."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, v in sd.items():
        if k.startswith("noise_estimator."):
            k = k[len("noise_estimator."):]
        elif "." in k and k.split(".")[0] in ("vision_encoder", "loss"):
            continue
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        out[k] = np.ascontiguousarray(v, dtype=np.float32)
    return out


class TensorIndex(ctypes.Structure):
    """Mirror of ``spdm_tensor_index`` (include/spdm.h)."""
    _fields_ = [
        ("name", ctypes.c_char * NAME_MAX),
        ("offset", ctypes.c_uint64),   # in floats, into the blob
        ("numel", ctypes.c_uint64),
        ("ndim", ctypes.c_int32),
        ("shape", ctypes.c_int32 * 4),
    ]


def pack_state_dict(sd) -> Tuple[np.ndarray, "ctypes.Array[TensorIndex]"]:
    """state_dict -> (flat fp32 blob, index array). The library does the
    kernel-friendly re-layout itself (tap-major, ci-contiguous conv weights);
    the blob keeps torch's native layouts so that any caller can produce it."""
    sd = state_dict_to_numpy(sd)
    total = sum(int(v.size) for v in sd.values())
    blob = np.empty(total, dtype=np.float32)
    idx = (TensorIndex * len(sd))()
    off = 0
    for i, (name, arr) in enumerate(sd.items()):
        if len(name.encode()) >= NAME_MAX:
            raise ValueError(f"tensor name too long for the C ABI: {name}")
        if arr.ndim > 4:
            raise ValueError(f"{name}: rank {arr.ndim} > 4")
        blob[off:off + arr.size] = arr.reshape(-1)
        idx[i].name = name.encode()
        idx[i].offset = off
        idx[i].numel = arr.size
        idx[i].ndim = arr.ndim
        for d in range(4):
            idx[i].shape[d] = arr.shape[d] if d < arr.ndim else 1
        off += arr.size
    return blob, idx


def blob_sha256(sd) -> str:
    import hashlib
    h = hashlib.sha256()
    for name, arr in state_dict_to_numpy(sd).items():
        h.update(name.encode())
        h.update(np.ascontiguousarray(arr, dtype=np.float32).tobytes())
    return h.hexdigest()


# ---- checkpoint ingestion ---------------------------------------------------------
def fetch_hyperparams_from_yaml(file_path: str) -> dict:
    """utils/data_utils.py:5-8 of the reference: the Lightning ``hparams.yaml`` is plain ``key: value`` YAML."""
    import yaml
    with open(file_path, "r") as fh:
        return yaml.safe_load(fh) or {}


def safe_load_state_dict(checkpoint_path: str):
    """The full ``state_dict`` of a Lightning ``.ckpt`` (or a bare state_dict file), read with
    ``torch.load(weights_only=True)`` only: a file the safe loader refuses is reported, never unpickled."""
    import torch
    try:
        ck = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001 - re-raised with the policy spelled out
        raise RuntimeError(f"{checkpoint_path}: torch.load(weights_only=True) refused this file ({e}); "
                           "re-save it as a plain state_dict (tensors only) -- it will not be unpickled here") from e
    sd = ck.get("state_dict", ck) if isinstance(ck, dict) else None
    if not isinstance(sd, dict) or not sd:
        raise RuntimeError(f"{checkpoint_path}: no state_dict found")
    return sd


def load_checkpoint_state_dict(checkpoint_path: str, prefix: str = "noise_estimator."):
    """The U-Net tensors of a reference checkpoint (Lightning ``.ckpt``: ``{'state_dict': {'noise_estimator.*': ...,
    'vision_encoder.*': ...}, ...}``; generate.py:23-26 reads it through ``load_from_checkpoint``), or of a bare
    ``state_dict`` file, with the ``noise_estimator.`` prefix removed.

    Only loaders that execute nothing from the file are used (``torch.load(weights_only=True)``); a checkpoint the
    safe loader refuses is reported, never unpickled.  Returns ``(unet_state_dict, other_keys)``."""
    sd = safe_load_state_dict(checkpoint_path)
    unet, other = OrderedDict(), []
    has_prefix = any(k.startswith(prefix) for k in sd)
    for k, v in sd.items():
        if has_prefix:
            if k.startswith(prefix):
                unet[k[len(prefix):]] = v
            else:
                other.append(k)
        else:
            unet[k] = v
    return unet, other


def check_state_dict(sd, global_cond_dim: int, time_dim: int = 256, attention: bool = True) -> None:
    """Key-for-key / shape-for-shape check against the UNet_Film parameter inventory (the same 162-tensor inventory
    the golden fixtures pin against the reference module, tools/make_golden.py)."""
    spec = unet_film_param_spec(global_cond_dim, time_dim=time_dim, attention=attention)
    want = dict(spec)
    missing = [k for k in want if k not in sd]
    extra = [k for k in sd if k not in want and not k.endswith("num_batches_tracked")]
    bad = [f"{k}: {tuple(sd[k].shape)} != {want[k]}" for k in want if k in sd and tuple(sd[k].shape) != tuple(want[k])]
    if missing or extra or bad:
        raise ValueError("state_dict does not match UNet_Film"
                         f"{'' if attention else '_noAttention'}(global_cond_dim={global_cond_dim}, time_dim={time_dim}): "
                         f"missing {missing[:4]}{'...' if len(missing) > 4 else ''}, unexpected {extra[:4]}"
                         f"{'...' if len(extra) > 4 else ''}, shape mismatches {bad[:4]}")


# ---- data (un)normalisation the reference applies around the sampler (utils/data_utils.py:10-40) ----------------
def normalize_data(data, stats):
    ndata = (data - stats["min"]) / (stats["max"] - stats["min"])
    return ndata * 2 - 1


def unnormalize_data(ndata, stats):
    ndata = (ndata + 1) / 2
    return ndata * (stats["max"] - stats["min"]) + stats["min"]


def normalize_position(sample, position_stats):
    sample_normalized = normalize_data(sample, position_stats)
    translation_vec = sample_normalized[0, :]
    return (sample_normalized - translation_vec) / 2.0, translation_vec


def unnormalize_position(nsample, translation_vec, position_stats):
    return unnormalize_data(np.asarray(nsample) * 2.0 + translation_vec, position_stats)

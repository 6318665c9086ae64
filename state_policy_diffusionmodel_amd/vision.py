"""Observation front end: the encoder of the reference's lightweight autoencoder on the GPU.

Mirror of ``self.vision_encoder`` in ``Diffusion_DDPM`` (models/diffusion_ddpm.py:84-88: ``vision.encoder`` of
models/encoder/autoencoder.py:11-20, an ``nn.Sequential`` whose state_dict keys are ``0.weight 0.bias 2.* 4.* 7.*``),
called by ``prepare_obs_cond_vectors`` (:317-321) on ``(B*obs_h, 3, 96, 96)`` frames.  All compute is in
libspdm_hip.so (``spdm_encoder_*``, csrc/encoder.hip + the product's GEMM); there is no CPU path here.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from .weights import pack_state_dict

ENCODER_KEYS = ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "7.weight", "7.bias")
ENCODER_SHAPES = {"0.weight": (16, 3, 2, 2), "0.bias": (16,), "2.weight": (32, 16, 2, 2), "2.bias": (32,),
                  "4.weight": (64, 32, 2, 2), "4.bias": (64,), "7.weight": (128, 9216), "7.bias": (128,)}


def encoder_state_dict_from(sd, prefix: str = "vision_encoder."):
    """Pick the encoder's tensors out of a diffusion checkpoint's state_dict (keys ``vision_encoder.N.*``) or an
    autoencoder checkpoint's (``encoder.N.*`` / ``model.encoder.N.*``).  Returns None when they are not there."""
    for pre in (prefix, "encoder.", "model.encoder.", ""):
        if all((pre + k) in sd for k in ENCODER_KEYS):
            return {k: sd[pre + k] for k in ENCODER_KEYS}
    return None


class VisionEncoder:
    """``VisionEncoder(state_dict)(images)`` == ``Autoencoder().encoder(images)`` (eval mode), images ``(N,3,96,96)``
    fp32 on the GPU -> ``(N,128)``."""

    def __init__(self, state_dict, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("VisionEncoder needs a visible MI355X (HIP device); there is no CPU fallback")
        sd = {k: state_dict[k] for k in ENCODER_KEYS}
        for k, shp in ENCODER_SHAPES.items():
            if tuple(sd[k].shape) != shp:
                raise ValueError(f"encoder tensor {k}: shape {tuple(sd[k].shape)}, expected {shp}")
        self.device = torch.device("cuda", device)
        blob, idx = pack_state_dict(sd)
        h = ctypes.c_void_p()
        _lib.check(self.lib.spdm_encoder_create(device, blob.ctypes.data_as(ctypes.c_void_p), blob.size, idx, len(idx),
                                                ctypes.byref(h)), "spdm_encoder_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.lib.spdm_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self):
        return self

    def __call__(self, images: torch.Tensor) -> torch.Tensor:
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, 96, 96):
            raise ValueError(f"expected (N,3,96,96) frames, got {tuple(images.shape)}")
        x = images.to(self.device, torch.float32).contiguous()
        out = torch.empty(x.shape[0], 128, device=self.device, dtype=torch.float32)
        if x.shape[0] == 0:
            return out
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(self.lib.spdm_encoder_forward(self._h, x.shape[0], ctypes.c_void_p(x.data_ptr()),
                                                 ctypes.c_void_p(out.data_ptr()), stream), "spdm_encoder_forward")
        return out

"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in torch-CPU fp32 functional ops, of the encoder half of the reference's lightweight autoencoder
(``/root/reference/models/encoder/autoencoder.py:11-20``), the ``self.vision_encoder`` that
``Diffusion_DDPM.prepare_obs_cond_vectors`` applies to every observed frame (``models/diffusion_ddpm.py:84-88,317-321``):

    Conv2d(3,16,2,stride=2,padding=1) ReLU  Conv2d(16,32,2,stride=2) ReLU  Conv2d(32,64,2,stride=2) ReLU
    Flatten  Linear(64*12*12, 128)

PINNED: ``tools/make_golden.py::encoder_case`` imports the reference ``Autoencoder`` class itself (stubbing the two
module-level imports the encoder never touches, ``pytorch_lightning`` and ``torchvision``), loads the tensors of
``make_encoder_state_dict`` into its ``.encoder`` with ``strict=True`` and commits its output on seeded frames
(``tests/golden/encoder_n5.npz``); ``tests/test_encoder.py`` checks this restatement -- and the HIP encoder -- against
that fixture.  Only ``tests/`` and ``__graft_entry__.smoke()`` may import this file.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

KEYS = ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias", "7.weight", "7.bias")


def make_encoder_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights with torch's default initialisers, in the nn.Sequential's own key names."""
    g = torch.Generator().manual_seed(seed)
    shapes = {"0": (16, 3, 2, 2), "2": (32, 16, 2, 2), "4": (64, 32, 2, 2), "7": (128, 9216)}
    sd = {}
    for k, shp in shapes.items():
        fan_in = 1
        for d in shp[1:]:
            fan_in *= d
        bound = 1.0 / fan_in ** 0.5                 # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
        sd[k + ".weight"] = (torch.rand(shp, generator=g) * 2 - 1) * bound
        sd[k + ".bias"] = (torch.rand(shp[0], generator=g) * 2 - 1) * bound
    return sd


def encoder_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor) -> torch.Tensor:
    """(N,3,96,96) -> (N,128); autoencoder.py:11-20."""
    x = images.float()
    x = F.relu(F.conv2d(x, sd["0.weight"], sd["0.bias"], stride=2, padding=1))     # :12-13  -> (N,16,49,49)
    x = F.relu(F.conv2d(x, sd["2.weight"], sd["2.bias"], stride=2))                # :14-15  -> (N,32,24,24)
    x = F.relu(F.conv2d(x, sd["4.weight"], sd["4.bias"], stride=2))                # :16-17  -> (N,64,12,12)
    x = x.flatten(1)                                                               # :18
    return F.linear(x, sd["7.weight"], sd["7.bias"])                               # :19

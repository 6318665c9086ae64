"""Live pin of the oracle against the imported reference module.  Runs only
where /root/reference exists (the build container); on the GPU box the
committed golden vectors (tests/test_oracle.py) stand in for it."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import REFERENCE
from oracle.unet_film_ref import unet_film_forward
from state_policy_diffusionmodel_amd.weights import random_state_dict, unet_film_param_spec

pytestmark = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not present")


def _import():
    sys.dont_write_bytecode = True
    sys.modules.setdefault("torchvision", types.ModuleType("torchvision"))
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    from models.Unet_FiLmLayer import UNet_Film
    from models.Unet_FiLmLayer_noAttention import UNet_Film_noAttention
    return UNet_Film, UNet_Film_noAttention


@pytest.mark.parametrize("attention", [True, False])
def test_inventory_matches_reference_state_dict(attention):
    cls = _import()[0 if attention else 1]
    m = cls(1, 1, 1000, time_dim=256, global_cond_dim=1350)
    ref = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    mine = dict(unet_film_param_spec(1350, attention=attention))
    assert list(ref.keys()) == list(mine.keys())
    assert ref == mine


@pytest.mark.parametrize("H,D,B,attention", [(24, 4, 3, True), (8, 1, 2, True), (48, 8, 1, False)])
def test_forward_matches_reference(H, D, B, attention):
    cls = _import()[0 if attention else 1]
    cond_dim = 3 * 11
    sd = random_state_dict(cond_dim, seed=11, attention=attention)
    m = cls(1, 1, 1000, time_dim=256, global_cond_dim=cond_dim)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    g = torch.Generator().manual_seed(H * 100 + D)
    x = torch.randn(B, 1, H, D, generator=g)
    y = torch.randn(B, 1, 3, 11, generator=g)
    for t in (torch.tensor([42]), torch.arange(B) * 300):
        with torch.no_grad():
            want = m(x, t, y).numpy()
        got = unet_film_forward(sd, x, t, y, attention=attention).numpy()
        assert np.abs(got - want).max() <= 2e-5

"""Observation front end: oracle vs torch.nn modules on CPU, HIP encoder vs oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle.encoder_ref import encoder_forward, make_encoder_state_dict

TOL = 1e-4


def _images(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, 96, 96, generator=g)           # frames are in [0,1] (the autoencoder ends in a Sigmoid)


def test_oracle_matches_torch_modules_built_like_the_reference():
    """autoencoder.py:11-20 as nn modules with the same constructor arguments, same weights."""
    sd = make_encoder_state_dict(3)
    enc = torch.nn.Sequential(torch.nn.Conv2d(3, 16, 2, stride=2, padding=1), torch.nn.ReLU(),
                              torch.nn.Conv2d(16, 32, 2, stride=2, padding=0), torch.nn.ReLU(),
                              torch.nn.Conv2d(32, 64, 2, stride=2, padding=0), torch.nn.ReLU(),
                              torch.nn.Flatten(), torch.nn.Linear(64 * 12 * 12, 128))
    enc.load_state_dict(sd, strict=True)
    x = _images(3, 1)
    with torch.no_grad():
        want = enc(x)
    got = encoder_forward(sd, x)
    assert got.shape == (3, 128)
    assert torch.equal(got, want)


def _golden():
    import os
    from conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, "encoder_n5.npz"))
    sd = make_encoder_state_dict(int(g["wseed"]))
    x = _images(int(g["n"]), int(g["iseed"]))
    # the fixture stores checksums of the regenerated inputs / weights instead of 550 KB of frames
    assert abs(float(x.double().sum()) - float(g["images_sum"])) <= 1e-6 * float(g["images_sum"])
    np.testing.assert_array_equal(x[0, 0, 0].numpy(), g["first_image_row"])
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["weights_sum"])) <= 1e-9 + 1e-9 * abs(float(g["weights_sum"]))
    return sd, x, g["latent"]


def test_oracle_matches_reference_golden():
    """tests/golden/encoder_n5.npz: output of the IMPORTED reference Autoencoder().encoder (tools/make_golden.py)."""
    sd, x, want = _golden()
    got = encoder_forward(sd, x).numpy()
    assert got.shape == want.shape == (5, 128)
    assert np.abs(got - want).max() <= 2e-5


@pytest.mark.gpu
def test_hip_encoder_matches_reference_golden():
    from state_policy_diffusionmodel_amd.vision import VisionEncoder
    sd, x, want = _golden()
    enc = VisionEncoder(sd)
    try:
        got = enc(x.cuda()).cpu().numpy()
        assert np.abs(got - want).max() <= TOL
    finally:
        enc.close()


def test_encoder_state_dict_extraction_from_checkpoint_key_styles():
    from state_policy_diffusionmodel_amd.vision import encoder_state_dict_from
    sd = make_encoder_state_dict(0)
    for pre in ("vision_encoder.", "encoder.", "model.encoder."):
        full = {pre + k: v for k, v in sd.items()}
        full["noise_estimator.inc.first.weight"] = torch.zeros(1)
        got = encoder_state_dict_from(full)
        assert got is not None and all(torch.equal(got[k], sd[k]) for k in sd)
    assert encoder_state_dict_from({"noise_estimator.x": torch.zeros(1)}) is None


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 7, 2500])
def test_hip_encoder_matches_oracle(n):
    """n = 2500 crosses the 2048-image chunk of spdm_encoder_forward."""
    from state_policy_diffusionmodel_amd.vision import VisionEncoder
    sd = make_encoder_state_dict(5)
    enc = VisionEncoder(sd)
    try:
        x = _images(n, n)
        x[0, :, :3, :3] = 0.0
        x[-1, :, -1, :] = 1.0                               # the last input row / column (read only through conv 1's window)
        got = enc(x.cuda()).cpu()
        want = encoder_forward(sd, x)
        assert got.shape == (n, 128)
        assert float((got - want).abs().max()) <= TOL
    finally:
        enc.close()


@pytest.mark.gpu
def test_facade_builds_obs_cond_from_raw_frames_with_the_native_encoder():
    """prepare_obs_cond_vectors (models/diffusion_ddpm.py:317-330): cat(position, action, velocity, encoder(frames))."""
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    from state_policy_diffusionmodel_amd.weights import random_state_dict
    enc_sd = make_encoder_state_dict(9)
    B, obs_h = 3, 2
    obs_dim = 2 + 3 + 2 + 128
    model = Diffusion_DDPM(noise_steps=20, obs_horizon=obs_h, pred_horizon=8, observation_dim=obs_dim, prediction_dim=5,
                           model="UNet_Film", state_dict=random_state_dict(obs_h * obs_dim, seed=1, attention=True),
                           vision_encoder_state_dict=enc_sd)
    g = torch.Generator().manual_seed(0)
    batch = {"image": torch.rand(B, obs_h, 3, 96, 96, generator=g), "position": torch.randn(B, obs_h, 2, generator=g),
             "action": torch.randn(B, obs_h, 3, generator=g), "velocity": torch.randn(B, obs_h, 2, generator=g)}
    ob = model.prepare_observation_batch(batch)
    got = model.prepare_obs_cond_vectors(ob).cpu()
    feats = encoder_forward(enc_sd, batch["image"].flatten(end_dim=1)).reshape(B, obs_h, 128)
    want = torch.cat([batch["position"], batch["action"], batch["velocity"], feats], dim=-1)
    assert got.shape == (B, obs_h, obs_dim)
    assert float((got - want).abs().max()) <= TOL

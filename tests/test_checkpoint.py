"""Checkpoint ingestion of the host facade: Lightning-style ``.ckpt`` + ``hparams.yaml``
-> sampler object, the way generate.py:23-37 / run_predictions.py of the reference obtain it.  CPU only: the sampler
creates its device engine lazily.  The checkpoint files are written by the test itself (the reference ships none,
.gitignore:12,21,23)."""
import numpy as np
import pytest
import torch
import yaml

from state_policy_diffusionmodel_amd import weights
from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDIM, Diffusion_DDPM, load_model
from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler

HP = dict(noise_steps=200, obs_horizon=5, pred_horizon=12, observation_dim=7, prediction_dim=5, learning_rate=1e-4,
          model="UNet_Film", vision_encoder="resnet18", noise_scheduler_type="linear", inpaint_horizon=3, step_size=1)


class Hook:
    """stands for the callbacks / loop state objects a Lightning checkpoint may carry"""


def _write(tmp_path, attention=True, drop=None, extra_obj=None):
    hp = dict(HP, model="UNet_Film" if attention else "UNet_FilmnoAttention")
    cond_dim = hp["observation_dim"] * hp["obs_horizon"]
    sd = weights.random_state_dict(cond_dim, seed=9, attention=attention)
    full = {"noise_estimator." + k: torch.from_numpy(np.array(v)) for k, v in weights.state_dict_to_numpy(sd).items()}
    full["vision_encoder.encoder.0.weight"] = torch.zeros(4, 3, 3, 3)
    if drop:
        del full["noise_estimator." + drop]
    ck = {"epoch": 4, "global_step": 100, "state_dict": full, "hyper_parameters": {k: v for k, v in hp.items()}}
    if extra_obj is not None:
        ck["callbacks"] = extra_obj
    cp, yp = tmp_path / "epoch=4.ckpt", tmp_path / "hparams.yaml"
    torch.save(ck, cp)
    yp.write_text(yaml.safe_dump(hp))
    return str(cp), str(yp), sd


def test_load_from_checkpoint_reads_hparams_and_unet_tensors(tmp_path):
    cp, yp, sd = _write(tmp_path)
    m = Diffusion_DDPM.load_from_checkpoint(cp, hparams_file=yp)
    assert (m.noise_steps, m.obs_horizon, m.pred_horizon, m.observation_dim, m.prediction_dim, m.inpaint_horizon) == (200, 5, 12, 7, 5, 3)
    assert m.attention and isinstance(m.noise_scheduler, DDPMScheduler)
    got, want = m.noise_estimator.state_dict(), weights.state_dict_to_numpy(sd)
    assert list(got) == list(want)
    assert all(np.array_equal(got[k].numpy(), want[k]) for k in want)


def test_load_model_keeps_the_reference_signature_and_ddim_swap(tmp_path):
    cp, yp, _ = _write(tmp_path, attention=False)
    m = load_model("DDIM", cp, yp, 50)
    assert isinstance(m, Diffusion_DDIM) and isinstance(m.noise_scheduler, DDIMScheduler)
    assert m.noise_steps == 50 and m.noise_scheduler.config.num_train_timesteps == 50 and not m.attention
    with pytest.raises(ValueError):
        load_model("PNDM", cp, yp)


def test_inventory_mismatch_and_unsafe_files_are_refused(tmp_path):
    cp, yp, _ = _write(tmp_path, drop="inc.first.weight")
    with pytest.raises(ValueError, match="missing"):
        Diffusion_DDPM.load_from_checkpoint(cp, hparams_file=yp)

    (tmp_path / "b").mkdir()
    cp2, yp2, _ = _write(tmp_path / "b", extra_obj=Hook())   # an arbitrary pickled object: refused, no fallback
    with pytest.raises(RuntimeError, match="weights_only"):
        Diffusion_DDPM.load_from_checkpoint(cp2, hparams_file=yp2)


def test_normalisation_helpers_follow_the_reference_formulas():
    rng = np.random.default_rng(0)
    data = rng.uniform(-3, 5, size=(40, 2))
    stats = {"min": data.min(0), "max": data.max(0)}
    nd = weights.normalize_data(data, stats)
    assert nd.min() == -1.0 and nd.max() == 1.0
    assert np.allclose(weights.unnormalize_data(nd, stats), data)
    ns, tv = weights.normalize_position(data, stats)
    assert np.allclose(ns[0], 0.0) and np.allclose(weights.unnormalize_position(ns, tv, stats), data)

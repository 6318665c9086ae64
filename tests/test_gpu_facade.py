"""GPU tests of the drop-in surface: Diffusion_DDPM / Diffusion_DDIM look-alikes driven the way
generate.py drives the reference (generate.py:23-37, 68-79)."""
import numpy as np
import pytest
import torch

from oracle.scheduler_ref import sample_loop
from oracle.unet_film_ref import unet_film_forward

pytestmark = pytest.mark.gpu


def _batch(B, obs_h, g):
    return {"position": torch.rand(B, obs_h, 2, generator=g) * 2 - 1, "velocity": torch.rand(B, obs_h, 2, generator=g),
            "action": torch.rand(B, obs_h, 3, generator=g), "image_features": torch.randn(B, obs_h, 4, generator=g)}


def test_ddpm_sample_matches_oracle_and_reference_shapes():
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    g = torch.Generator().manual_seed(0)
    m = Diffusion_DDPM(noise_steps=8, obs_horizon=3, pred_horizon=14, observation_dim=11, prediction_dim=5,
                       model="UNet_Film", inpaint_horizon=2, weight_seed=4)
    m.eval()
    obs = m.prepare_observation_batch(_batch(4, 5, g))
    assert obs["position"].shape == (4, 3, 2)
    H, D = 16, 5
    x_T = torch.rand(1, 1, H, D, generator=g)
    noise = torch.randn(8, 1, 1, H, D, generator=g)
    out = m.sample(dict(obs), x_T=x_T.cuda(), noise=noise.cuda())
    assert out.shape == (1, 1, H, D)                                  # reference: B forced to 1
    cond = m.prepare_obs_cond_vectors(obs)[0:1].unsqueeze(1).cpu()
    inp = m.prepare_inpaint_vectors(obs)[0:1].unsqueeze(1).cpu()
    assert cond.shape == (1, 1, 3, 11) and inp.shape == (1, 1, 2, 5)
    sd = m.noise_estimator._sd
    want = sample_loop(lambda x, t, y: unet_film_forward(sd, x, t, y), "ddpm", 8, 8, cond, x_T, noise, inp)
    assert float((out.cpu() - want).abs().max()) <= 1e-4
    hist = m.sample(dict(obs), option="sample_history", x_T=x_T.cuda(), noise=noise.cuda())
    assert isinstance(hist, list) and len(hist) == 9 and hist[0].shape == (1, 1, H, D)
    assert torch.equal(hist[-1], out)
    # the noise predictor is callable like the reference's module
    eps = m.noise_estimator(x_T.cuda(), torch.tensor([3]), cond.cuda())
    assert float((eps.cpu() - unet_film_forward(sd, x_T, torch.tensor([3]), cond)).abs().max()) <= 1e-4


def test_ddim_scheduler_swap_idiom():
    from state_policy_diffusionmodel_amd.diffusion import load_model
    g = torch.Generator().manual_seed(1)
    m = load_model("DDIM", num_of_ddim_steps=6, noise_steps=1000, obs_horizon=2, pred_horizon=7, observation_dim=11,
                   prediction_dim=5, model="UNet_FilmnoAttention", inpaint_horizon=1, weight_seed=6, max_batch=3)
    assert m.noise_steps == 6 and type(m.noise_scheduler).__name__ == "DDIMScheduler"
    obs = m.prepare_observation_batch(_batch(3, 2, g))
    x_T = torch.rand(3, 1, 8, 5, generator=g)
    out = m.sample(dict(obs), x_T=x_T.cuda(), batched=True)
    assert out.shape == (3, 1, 8, 5)
    cond = m.prepare_obs_cond_vectors(obs).unsqueeze(1).cpu()
    inp = m.prepare_inpaint_vectors(obs).unsqueeze(1).cpu()
    sd = m.noise_estimator._sd
    want = sample_loop(lambda x, t, y: unet_film_forward(sd, x, t, y, attention=False), "ddim", 6, 6, cond, x_T, None,
                       inp)
    assert float((out.cpu() - want).abs().max()) <= 1e-4


def test_training_step_forward_half_and_validate_against_oracle():
    """models/diffusion_ddpm.py:140-172 (forward only): add_noise at per-sample t -> inpaint -> U-Net(t:(B,)) -> MSE,
    checked against the torch-CPU oracle on the same t / noise; validate() = the first trajectory through the loop."""
    from oracle.scheduler_ref import LinearBetaSchedule
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    g = torch.Generator().manual_seed(5)
    obs_h, pred_h, inp_h, B = 3, 13, 3, 4
    m = Diffusion_DDPM(noise_steps=50, obs_horizon=obs_h, pred_horizon=pred_h, observation_dim=11, prediction_dim=5,
                       model="UNet_Film", inpaint_horizon=inp_h, weight_seed=8, max_batch=B)
    batch = _batch(B, obs_h + pred_h, g)
    t = torch.tensor([0, 7, 31, 49])
    noise = torch.randn(B, 1, pred_h + inp_h, 5, generator=g)
    loss, eps, x_noisy = m.training_step({k: v.clone() for k, v in batch.items()}, t=t, noise=noise, return_parts=True)
    assert eps.shape == (B, 1, pred_h + inp_h, 5) and loss.ndim == 0
    # oracle: the same forward process in torch on the CPU
    obs = {k: v[:, :obs_h].float() for k, v in batch.items()}
    cond = torch.cat([obs["position"], obs["action"], obs["velocity"], obs["image_features"]], -1).unsqueeze(1)
    x0 = torch.cat([batch["position"][:, obs_h:], batch["action"][:, obs_h:]], -1).unsqueeze(1).float()
    inp = torch.cat([obs["position"][:, -inp_h:], obs["action"][:, -inp_h:]], -1).unsqueeze(1)
    pv = torch.cat([inp, x0], 2)
    ac = LinearBetaSchedule(50).alphas_cumprod
    a = ac[t].sqrt().view(B, 1, 1, 1)
    b = (1 - ac[t]).sqrt().view(B, 1, 1, 1)
    xn = a * pv + b * noise
    xn[:, :, :inp_h, :] = inp
    assert float((x_noisy.cpu() - xn).abs().max()) <= 1e-6
    want = unet_film_forward(m.noise_estimator._sd, xn, t, cond)
    assert float((eps.cpu() - want).abs().max()) <= 1e-4
    assert abs(float(loss) - float(torch.mean((noise - want) ** 2))) <= 1e-5
    # validate(): first trajectory only, like the reference
    x_T = torch.rand(1, 1, pred_h + inp_h, 5, generator=g)
    nz = torch.randn(50, 1, 1, pred_h + inp_h, 5, generator=g)
    x0v, ob, iv = m.validate({k: v.clone() for k, v in batch.items()}, x_T=x_T.cuda(), noise=nz.cuda())
    assert x0v.shape == (1, 1, pred_h + inp_h, 5) and iv.shape == (1, 1, inp_h, 5)
    want_v = sample_loop(lambda x, tt, y: unet_film_forward(m.noise_estimator._sd, x, tt, y), "ddpm", 50, 50, cond[0:1],
                         x_T, nz, inp[0:1])
    assert float((x0v.cpu() - want_v).abs().max()) <= 1e-4


def test_default_seed_is_fresh_per_call_and_follows_manual_seed():
    """The reference's DDPM step draws torch.randn from the global generator on every step of every call (diffusers
    `step`, called at models/diffusion_ddpm.py:274): two sample() calls differ, torch.manual_seed reproduces them."""
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    g = torch.Generator().manual_seed(3)
    m = Diffusion_DDPM(noise_steps=12, obs_horizon=3, pred_horizon=14, observation_dim=11, prediction_dim=5,
                       model="UNet_Film", inpaint_horizon=2, weight_seed=4)
    obs = m.prepare_observation_batch(_batch(2, 5, g))
    x_T = torch.rand(1, 1, 16, 5, generator=g).cuda()
    a = m.sample(dict(obs), x_T=x_T).cpu()
    b = m.sample(dict(obs), x_T=x_T).cpu()
    assert float((a - b).abs().max()) > 1e-3                     # same x_T, fresh noise: different trajectories
    torch.manual_seed(1234)
    c = m.sample(dict(obs), x_T=x_T).cpu()
    torch.manual_seed(1234)
    d = m.sample(dict(obs), x_T=x_T).cpu()
    assert torch.equal(c, d)
    e = m.sample(dict(obs), x_T=x_T, seed=99).cpu()             # an explicit seed still pins the stream
    f = m.sample(dict(obs), x_T=x_T, seed=99).cpu()
    assert torch.equal(e, f) and not torch.equal(e, c)


def test_history_stream_yields_the_same_iterates_as_the_history_list():
    """option='sample_history_stream': iterates handed out while the loop runs (every k steps) == the list form."""
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    g = torch.Generator().manual_seed(4)
    m = Diffusion_DDPM(noise_steps=10, obs_horizon=3, pred_horizon=14, observation_dim=11, prediction_dim=5,
                       model="UNet_Film", inpaint_horizon=2, weight_seed=4)
    obs = m.prepare_observation_batch(_batch(2, 5, g))
    x_T = torch.rand(1, 1, 16, 5, generator=g).cuda()
    hist = m.sample(dict(obs), option="sample_history", x_T=x_T, seed=5)
    for every in (1, 3, 4):
        got = list(m.sample(dict(obs), option="sample_history_stream", x_T=x_T, seed=5, every=every))
        want_idx = list(range(0, 10, every)) + [10]
        assert [i for i, _ in got] == want_idx
        for i, x in got:
            assert not x.is_cuda and torch.equal(x, hist[i].cpu())


def test_closed_loop_calls_replay_one_captured_step():
    """run_predictions.py:151-156 calls model.sample(batch) once per control period with freshly built tensors.  The
    step graph is keyed by the session's shape, not by buffer addresses or the seed: the first call captures, every
    later one replays -- with or without a history buffer, whatever the allocator hands out."""
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    g = torch.Generator().manual_seed(3)
    m = Diffusion_DDPM(noise_steps=8, obs_horizon=3, pred_horizon=14, observation_dim=11, prediction_dim=5,
                       model="UNet_Film", inpaint_horizon=2, weight_seed=4)
    keep = []
    outs = []
    for call in range(4):
        obs = m.prepare_observation_batch(_batch(2, 5, g))
        keep.append(torch.empty(1000 + 333 * call, device="cuda"))          # perturb the allocator between calls
        x_T = torch.rand(1, 1, 16, 5, generator=g).cuda()
        if call == 2:
            hist = m.sample(dict(obs), option="sample_history", x_T=x_T, seed=call)
            outs.append(hist[-1])
        else:
            outs.append(m.sample(dict(obs), x_T=x_T, seed=call))
        assert m._engine.graph_captures == 1, call
    # and the replayed graph really reads the NEW buffers: same inputs again give the same result, different ones differ
    obs = m.prepare_observation_batch(_batch(2, 5, torch.Generator().manual_seed(77)))
    x_T = torch.rand(1, 1, 16, 5, generator=torch.Generator().manual_seed(78)).cuda()
    a = m.sample(dict(obs), x_T=x_T.clone(), seed=5)
    b = m.sample(dict(obs), x_T=x_T.clone(), seed=5)
    assert torch.equal(a, b) and not torch.equal(a, outs[0])
    assert torch.equal(a[:, :, :2, :].cpu(), m.prepare_inpaint_vectors(obs)[0:1].unsqueeze(1).cpu())   # the NEW inpaint rows
    assert m._engine.graph_captures == 1

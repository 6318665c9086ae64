"""Property tests for the scheduler restatement (diffusers is absent, so the
scheduler is PARITY-UNPINNED; these closed forms are its guard -- SURVEY.md
section 4 item 3)."""
import numpy as np
import pytest
import torch

from oracle.scheduler_ref import LinearBetaSchedule, ddim_step, ddpm_step


def _f64_tables(T):
    betas = np.linspace(1e-4, 0.02, T, dtype=np.float64)
    return betas, np.cumprod(1 - betas)


@pytest.mark.parametrize("T", [10, 100, 1000])
def test_timesteps_full_schedule(T):
    s = LinearBetaSchedule(T)
    s.set_timesteps(T)
    assert s.timesteps.dtype == torch.int64
    assert s.timesteps.tolist() == list(range(T - 1, -1, -1))


def test_timesteps_subsampled():
    s = LinearBetaSchedule(1000)
    s.set_timesteps(50)
    assert s.timesteps.tolist() == list(range(980, -1, -20))
    assert s.prev_timestep(980) == 960 and s.prev_timestep(0) == -20


@pytest.mark.parametrize("T", [20, 1000])
def test_ddpm_posterior_mean_closed_form(T):
    # with n == T the posterior mean is 1/sqrt(alpha_t) (x - beta_t/sqrt(1-abar_t) eps)   (Ho et al. Eq. 11)
    s = LinearBetaSchedule(T)
    s.set_timesteps(T)
    # closed form evaluated in f64 ON THE fp32 TABLES: isolates the step algebra from the
    # (reference-inherent) fp32 cumprod rounding, which test_tables_close_to_f64 bounds separately.
    # At t ~ 0 the fp32 scheduler forms 1 - abar_t and 1 - alpha_t by cancellation (abar_1 ~ 0.9998):
    # a relative error of ~3e-4 on the eps coefficient that the fp32 reference shares, hence the
    # looser bound at t = 1.
    betas, acp = s.betas.double().numpy(), s.alphas_cumprod.double().numpy()
    g = torch.Generator().manual_seed(0)
    x, eps = torch.randn(4, 1, 8, 3, generator=g), torch.randn(4, 1, 8, 3, generator=g)
    for t in (T - 1, T // 2, 1):
        got = ddpm_step(s, eps, t, x, torch.zeros_like(x)).double().numpy()
        want = (x.double().numpy() - betas[t] / np.sqrt(1 - acp[t]) * eps.double().numpy()) / np.sqrt(1 - betas[t])
        assert np.abs(got - want).max() < (5e-4 if t == 1 else 2e-6)


@pytest.mark.parametrize("T", [20, 100, 1000])
def test_tables_close_to_f64(T):
    s = LinearBetaSchedule(T)
    betas, acp = _f64_tables(T)
    assert np.abs(s.betas.double().numpy() / betas - 1).max() < 1e-6
    assert np.abs(s.alphas_cumprod.double().numpy() / acp - 1).max() < 1e-4


def test_ddpm_variance_and_t0():
    T = 100
    s = LinearBetaSchedule(T)
    s.set_timesteps(T)
    betas, acp = _f64_tables(T)
    x, eps = torch.ones(1, 1, 4, 2), torch.zeros(1, 1, 4, 2)
    z = torch.full_like(x, 3.0)
    for t in (99, 10):
        d = (ddpm_step(s, eps, t, x, z) - ddpm_step(s, eps, t, x, torch.zeros_like(x))).double().numpy()
        sigma = np.sqrt((1 - acp[t - 1]) / (1 - acp[t]) * betas[t])
        assert np.abs(d - 3.0 * sigma).max() < 1e-6
    # t == 0: no noise is added and a_prev == 1 -> x_0 prediction itself
    got = ddpm_step(s, eps, 0, x, z).double().numpy()
    assert np.abs(got - x.numpy() / np.sqrt(acp[0])).max() < 1e-6


def test_ddim_eta0_exact_on_true_noise():
    T, n = 1000, 50
    s = LinearBetaSchedule(T)
    s.set_timesteps(n)
    _, acp = _f64_tables(T)
    g = torch.Generator().manual_seed(1)
    x0, eps = torch.randn(2, 1, 16, 3, generator=g), torch.randn(2, 1, 16, 3, generator=g)
    for t in (980, 500, 20, 0):
        xt = np.sqrt(acp[t]) * x0.double() + np.sqrt(1 - acp[t]) * eps.double()
        got = ddim_step(s, eps, t, xt.float()).double().numpy()
        a_prev = acp[t - 20] if t - 20 >= 0 else 1.0
        want = np.sqrt(a_prev) * x0.double().numpy() + np.sqrt(1 - a_prev) * eps.double().numpy()
        assert np.abs(got - want).max() < 5e-5

"""GPU parity tests (pytest -m gpu): the HIP path, called through the C ABI, against
 (1) the committed golden vectors produced from the IMPORTED reference U-Net,
 (2) the oracle (oracle/*.py) on fresh seeded inputs,
 (3) size-independent properties at BASELINE.json's full batch (4096).
Tolerance: 1e-4 absolute, fp32 -- the bound BASELINE.json's north_star states."""
import glob
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import philox_ref
from oracle.scheduler_ref import sample_loop
from oracle.unet_film_ref import unet_film_forward
from state_policy_diffusionmodel_amd.weights import blob_sha256, random_state_dict

pytestmark = pytest.mark.gpu
TOL = 1e-4

UNET_FILES = sorted(glob.glob(os.path.join(GOLDEN, "unet_*.npz")))
TRAJ_FILES = sorted(glob.glob(os.path.join(GOLDEN, "traj_*.npz")))
TAPS = {"inc": "x1", "down1": "d1", "sa1": "x2", "down2": "d2", "sa2": "x3", "down3": "d3", "sa3": "x4",
        "bot3": "x5", "up1": "u1", "sa4": "a4", "up2": "u2", "sa5": "a5", "up3": "u3", "sa6": "a6"}
_SD = {}


def weights(cond_dim, seed, attention=True, sha=None):
    key = (cond_dim, seed, attention)
    if key not in _SD:
        _SD[key] = random_state_dict(cond_dim, seed=seed, attention=attention)
        if sha is not None:
            assert blob_sha256(_SD[key]) == sha, "weight generator drifted from the fixtures"
    return _SD[key]


def make_engine(H, D, cond_dim, B, sd, attention=True, T=1000, debug=False, exact_fp32=False):
    from state_policy_diffusionmodel_amd.engine import SpdmEngine
    eng = SpdmEngine(H, D, cond_dim, max_batch=B, attention=attention, num_train_timesteps=T, debug=debug,
                     exact_fp32=exact_fp32)
    eng.load_state_dict(sd)
    return eng


def test_native_library_is_the_one_loaded():
    from state_policy_diffusionmodel_amd import _lib
    _lib.load()
    maps = open("/proc/self/maps").read()
    assert "libspdm_hip.so" in maps


@pytest.mark.parametrize("path", UNET_FILES, ids=[os.path.basename(p) for p in UNET_FILES])
def test_unet_matches_reference_golden(path):
    g = np.load(path)
    H, D, B = int(g["H"]), int(g["D"]), int(g["B"])
    cond_dim = int(g["obs_h"]) * int(g["obs_dim"])
    attention = bool(int(g["attention"]))
    sd = weights(cond_dim, int(g["wseed"]), attention, str(g["weights_sha256"]))
    has_taps = any(k.startswith("tap_") for k in g.files)
    eng = make_engine(H, D, cond_dim, B, sd, attention, debug=has_taps)
    x, cond = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["cond"]).cuda()
    try:
        for t, want in zip(g["t"], g["eps"]):
            got = eng.unet_forward(x, np.atleast_1d(t), cond).cpu().numpy()
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= TOL, (path, t)
        if has_taps:
            eng.unet_forward(x, np.atleast_1d(g["t"][0]), cond)
            for ref_name, mine in TAPS.items():
                if "tap_" + ref_name in g.files and (attention or not ref_name.startswith("sa")):
                    d = np.abs(eng.debug_tensor(mine).cpu().numpy() - g["tap_" + ref_name]).max()
                    assert d <= TOL, (ref_name, d)
    finally:
        eng.close()


@pytest.mark.parametrize("exact", [False, True])
def test_both_contraction_paths_match_golden(exact):
    """Default split-fp16 MFMA path and the exact fp32 MFMA path (SPDM_FLAG_EXACT_FP32) against the same
    reference golden; the unfused attention path is exercised by the exact engine (fused kernel is split-only)."""
    g = np.load(os.path.join(GOLDEN, "unet_h32d3_b2.npz"))
    sd = weights(1350, 0, True, str(g["weights_sha256"]))
    eng = make_engine(32, 3, 1350, 2, sd, True, exact_fp32=exact)
    try:
        assert eng.split_precision == (not exact)
        got = eng.unet_forward(torch.from_numpy(g["x"]).cuda(), np.atleast_1d(g["t"][0]), torch.from_numpy(g["cond"]).cuda())
        assert np.abs(got.cpu().numpy() - g["eps"][0]).max() <= TOL
    finally:
        eng.close()


def test_unfused_fallbacks_match_fused_paths(monkeypatch):
    """The fused SelfAttention kernels / MFMA attention / zero-tap skipping / FiLM fold / wide conv each have a plain fallback
    (env switches); all must agree with the default build on the same input to fp32 rounding."""
    g = np.load(os.path.join(GOLDEN, "unet_h32d3_b2.npz"))
    sd = weights(1350, 0, True, str(g["weights_sha256"]))
    x, cond = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["cond"]).cuda()
    outs = []
    for env in ({}, {"SPDM_NO_SA_FUSED": "1"}, {"SPDM_NO_SA_FUSED": "1", "SPDM_ATTN_VALU": "1"}, {"SPDM_NO_W2": "1", "SPDM_NO_T512": "1"},
                {"SPDM_NO_FILM_FOLD": "1"}, {"SPDM_NO_SA_TAIL": "1"}, {"SPDM_NO_WIDE": "1"}, {"SPDM_NO_FILM_LOCAL": "1"},
                {"SPDM_NO_FUSED_SRC": "1"}, {"SPDM_NO_FUSED_SRC": "1", "SPDM_NO_FILM_LOCAL": "1", "SPDM_NO_SKINNY": "1"},
                {"SPDM_FILM_LOCAL": "1"}, {"SPDM_NO_WP4": "1"}, {"SPDM_G2": "1"}, {"SPDM_NO_WP8": "1"}, {"SPDM_SA_HEAD": "1"}, {"SPDM_NO_REG64": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = make_engine(32, 3, 1350, 2, sd, True)
        try:
            outs.append(eng.unet_forward(x, [500], cond).cpu().numpy())
        finally:
            eng.close()
        for k in env:
            monkeypatch.delenv(k)
    for o in outs:
        assert np.abs(o - g["eps"][0]).max() <= TOL
        assert np.abs(o - outs[0]).max() <= 2e-5


@pytest.mark.parametrize("H,D,B,attention,cond", [(24, 4, 5, True, True), (8, 1, 3, True, True),
                                                  (48, 8, 2, False, True), (16, 3, 4, True, False),
                                                  (32, 3, 33, True, True), (32, 3, 96, True, True),
                                                  (16, 3, 128, True, True)])
def test_unet_matches_oracle_on_fresh_inputs(H, D, B, attention, cond):
    obs_h, obs_dim = 3, 11
    sd = weights(obs_h * obs_dim, 21, attention)
    g = torch.Generator().manual_seed(H * 1000 + D * 10 + B)
    x = torch.randn(B, 1, H, D, generator=g) * 1.5
    y = torch.randn(B, 1, obs_h, obs_dim, generator=g) if cond else None
    eng = make_engine(H, D, obs_h * obs_dim, B, sd, attention)
    try:
        for t in (torch.tensor([17]), (torch.arange(B) * 37) % 1000):
            want = unet_film_forward(sd, x, t, y, attention=attention).numpy()
            got = eng.unet_forward(x.cuda(), t, None if y is None else y.cuda()).cpu().numpy()
            assert np.abs(got - want).max() <= TOL
    finally:
        eng.close()


@pytest.mark.parametrize("path", TRAJ_FILES, ids=[os.path.basename(p) for p in TRAJ_FILES])
def test_sampling_loop_matches_golden_trajectories(path):
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler
    g = np.load(path)
    kind, T, N = str(g["kind"]), int(g["T"]), int(g["N"])
    H, D, B = int(g["H"]), int(g["D"]), int(g["B"])
    cond_dim = int(g["obs_h"]) * int(g["obs_dim"])
    sd = weights(cond_dim, int(g["wseed"]), True, str(g["weights_sha256"]))
    eng = make_engine(H, D, cond_dim, B, sd, T=T)
    try:
        sched = (DDPMScheduler if kind == "ddpm" else DDIMScheduler)(num_train_timesteps=T)
        sched.set_timesteps(N)
        eng.set_scheduler(sched)
        inpaint = torch.from_numpy(g["inpaint"]).cuda() if "inpaint" in g.files else None
        noise = torch.from_numpy(g["noise"]).cuda() if kind == "ddpm" else None
        x0, hist = eng.sample(torch.from_numpy(g["cond"]).cuda(), torch.from_numpy(g["x_T"]).cuda(), noise=noise,
                              inpaint=inpaint, history=True)
        hist = hist.cpu().numpy()
        assert hist.shape == g["history"].shape
        per_step = np.abs(hist - g["history"]).reshape(hist.shape[0], -1).max(axis=1)
        assert per_step.max() <= TOL, per_step                 # every intermediate iterate, not only x_0
        np.testing.assert_array_equal(x0.cpu().numpy(), hist[-1])
        if inpaint is not None:                                  # add_constraints: exact overwrite
            k = int(g["inp_h"])
            np.testing.assert_array_equal(hist[1:, :, :, :k, :], np.broadcast_to(g["inpaint"], hist[1:, :, :, :k, :].shape))
    finally:
        eng.close()


def test_long_ddpm_chain_against_oracle():
    """100-step DDPM (BASELINE configs[0] geometry: batch 1, horizon 16, state_dim 3, 100 steps) against the
    oracle loop on this box, every iterate."""
    from state_policy_diffusionmodel_amd.schedulers import DDPMScheduler
    T = 100
    B, H, D, obs_h, obs_dim = 1, 16, 3, 10, 135
    sd = weights(obs_h * obs_dim, 0)
    g = torch.Generator().manual_seed(5)
    cond = torch.randn(B, 1, obs_h, obs_dim, generator=g)
    x_T = torch.rand(B, 1, H, D, generator=g)
    noise = torch.randn(T, B, 1, H, D, generator=g)
    inpaint = torch.rand(B, 1, 1, D, generator=g) * 2 - 1
    want = sample_loop(lambda x, t, y: unet_film_forward(sd, x, t, y), "ddpm", T, T, cond, x_T, noise, inpaint, history=True)
    eng = make_engine(H, D, obs_h * obs_dim, B, sd, T=T)
    try:
        s = DDPMScheduler(num_train_timesteps=T)
        s.set_timesteps(T)
        eng.set_scheduler(s)
        _, hist = eng.sample(cond.cuda(), x_T.cuda(), noise=noise.cuda(), inpaint=inpaint.cuda(), history=True)
        err = (hist.cpu() - torch.stack(want)).abs().reshape(T + 1, -1).max(dim=1).values
        assert float(err.max()) <= TOL, err
    finally:
        eng.close()


def test_builtin_schedule_equals_installed_tables():
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler
    B, H, D, cd = 2, 16, 3, 14
    sd = weights(cd, 5)
    g = torch.Generator().manual_seed(9)
    cond, x_T = torch.randn(B, 1, 2, 7, generator=g).cuda(), torch.rand(B, 1, H, D, generator=g).cuda()
    eng = make_engine(H, D, cd, B, sd, T=50)
    try:
        s = DDIMScheduler(num_train_timesteps=50)
        s.set_timesteps(10)
        eng.set_scheduler(s)
        a = eng.sample(cond, x_T).cpu()
        eng.set_builtin_schedule(1, 50, 10)
        b = eng.sample(cond, x_T).cpu()
        assert float((a - b).abs().max()) <= 1e-4
    finally:
        eng.close()


def test_batch_independence_and_determinism():
    """B independent trajectories == B single runs (the reference hard-wires B = 1).  Two identical calls give identical
    bits; a trajectory's result does not depend on what else is in the batch -- bit for bit between batches that select the
    same kernels (here 6 vs 5 trajectories), and to fp32 rounding (<= 2e-6) against a single-trajectory run, which at the
    coarse levels runs on the few-row kernel (conv_skinny.hip) with a different, equally fixed, summation order."""
    B, H, D, cd = 6, 32, 3, 33
    sd = weights(cd, 21)
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, 1, H, D, generator=g).cuda()
    y = torch.randn(B, 1, 3, 11, generator=g).cuda()
    eng = make_engine(H, D, cd, B, sd)
    try:
        full = eng.unet_forward(x, [3], y).cpu()
        again = eng.unet_forward(x, [3], y).cpu()
        assert torch.equal(full, again)
        part = eng.unet_forward(x[1:], [3], y[1:]).cpu()           # the same trajectories in another batch, same kernels
        assert torch.equal(part, full[1:])
        for b in (0, 3, 5):
            one = eng.unet_forward(x[b:b + 1], [3], y[b:b + 1]).cpu()
            one2 = eng.unet_forward(x[b:b + 1], [3], y[b:b + 1]).cpu()
            assert torch.equal(one, one2)
            assert float((one[0] - full[b]).abs().max()) <= 2e-6
    finally:
        eng.close()


def test_device_philox_stream_matches_oracle_and_is_shard_invariant():
    from state_policy_diffusionmodel_amd.schedulers import DDPMScheduler
    T, B, H, D, cd = 6, 8, 16, 3, 14
    sd = weights(cd, 5)
    g = torch.Generator().manual_seed(11)
    cond, x_T = torch.randn(B, 1, 2, 7, generator=g), torch.rand(B, 1, H, D, generator=g)
    noise = np.stack([philox_ref.step_noise(1234, i, 0, B, H * D) for i in range(T)]).reshape(T, B, 1, H, D)
    want = sample_loop(lambda x, t, y: unet_film_forward(sd, x, t, y), "ddpm", T, T, cond, x_T,
                       torch.from_numpy(noise), None)
    eng = make_engine(H, D, cd, B, sd, T=T)
    try:
        s = DDPMScheduler(num_train_timesteps=T)
        s.set_timesteps(T)
        eng.set_scheduler(s)
        got = eng.sample(cond.cuda(), x_T.cuda(), noise=None, seed=1234).cpu()
        assert float((got - want).abs().max()) <= TOL
        # a "rank" that owns global trajectories 5..7 reproduces them: same noise stream (keyed by the GLOBAL index), same
        # trajectories to fp32 rounding (a 3-trajectory shard runs its coarse levels on the few-row kernel) ...
        part = eng.sample(cond[5:].cuda(), x_T[5:].cuda(), noise=None, seed=1234, sample_offset=5).cpu()
        assert float((part - got[5:]).abs().max()) <= 1e-5
        # ... and bit for bit when the shard selects the same kernels as the full batch (7 of the 8 trajectories)
        part7 = eng.sample(cond[1:].cuda(), x_T[1:].cuda(), noise=None, seed=1234, sample_offset=1).cpu()
        assert torch.equal(part7, got[1:])
    finally:
        eng.close()


@pytest.mark.parametrize("B,H,D,kind", [(4096, 32, 3, "ddpm"), (1024, 64, 6, "ddpm"), (1500, 16, 3, "ddpm"),
                                        (256, 32, 3, "ddpm"), (512, 32, 3, "ddpm"), (1024, 32, 3, "ddim"),
                                        (1024, 64, 6, "ddim"), (512, 64, 6, "ddim")],
                         ids=["config4_b4096_h32d3", "config5_geometry_b1024_h64d6", "b1500_h16d3",
                              "config2_b256_h32d3", "config4_per_rank_b512_h32d3", "config3_ddim_b1024_h32d3",
                              "config5_ddim_b1024_h64d6", "config5_per_rank_ddim_b512_h64d6"])
def test_full_size_batch_properties(B, H, D, kind):
    """BASELINE.json's batches (4096 x horizon 32; horizon 64 x state_dim 6; plus a ragged one at horizon 16), checked on
    EVERY trajectory -- where a trajectory sits inside a 256-row tile selects different code (64 samples per tile at
    level 3, 16 at level 2, parity-permuted rows, ragged last tiles), so a handful of probes would not do:
      (1) one U-Net evaluation of the whole batch against the ORACLE on the same inputs, all B trajectories (the oracle
          runs in chunks of 512 on the host: seconds), tolerance 1e-4 as BASELINE.json states;
      (2) a 2-step sampling loop (device noise, in-painting) of the whole batch against the same trajectories run in
          chunks of 64 -- the oracle-checked small-batch regime -- all B of them, to fp32 rounding;
      (3) inpainted rows exact, everything finite."""
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler
    cd, T, N = 1350, (1000 if kind == "ddpm" else 50), 2
    sd = weights(cd, 0)
    g = torch.Generator().manual_seed(2)
    cond_h = torch.randn(B, 1, 10, 135, generator=g)
    x_h = torch.rand(B, 1, H, D, generator=g)
    cond, x_T = cond_h.cuda(), x_h.cuda()
    inpaint = (torch.rand(B, 1, 1, D, generator=g) * 2 - 1).cuda()
    eng = make_engine(H, D, cd, B, sd, T=T)
    try:
        s = (DDPMScheduler if kind == "ddpm" else DDIMScheduler)(num_train_timesteps=T)     # DDIM: generate.py:28-35's reading
        s.set_timesteps(T)
        eng.set_scheduler(s)
        # (1) every trajectory of ONE evaluation against the oracle
        t0 = torch.tensor([int(s.timesteps[0])])
        got = eng.unet_forward(x_T, t0, cond).cpu()
        assert not eng.nonfinite()
        worst = 0.0
        for c0 in range(0, B, 512):
            want = unet_film_forward(sd, x_h[c0:c0 + 512], t0, cond_h[c0:c0 + 512])
            per = (got[c0:c0 + 512] - want).abs().reshape(want.shape[0], -1).max(dim=1).values
            bad = torch.nonzero(per > TOL).flatten()
            assert bad.numel() == 0, ("trajectories off the oracle", (bad[:8] + c0).tolist(), float(per.max()))
            worst = max(worst, float(per.max()))
        # (2) + (3) the loop, all trajectories against chunks of 64
        eng.sample_begin(cond, x_T, inpaint=inpaint, seed=3)
        eng.sample_run(0, N)
        big = eng.sample_result().cpu()
        assert not eng.nonfinite()
        assert bool(torch.isfinite(big).all())
        assert torch.equal(big[:, :, :1, :], inpaint.cpu())
        for c0 in range(0, B, 64):
            c1 = min(B, c0 + 64)
            eng.sample_begin(cond[c0:c1], x_T[c0:c1], inpaint=inpaint[c0:c1], seed=3, sample_offset=c0)
            eng.sample_run(0, N)
            small = eng.sample_result().cpu()
            per = (small - big[c0:c1]).abs().reshape(c1 - c0, -1).max(dim=1).values
            bad = torch.nonzero(per > 1e-5).flatten()
            assert bad.numel() == 0, ("trajectories differ from their small-batch run", (bad[:8] + c0).tolist(), float(per.max()))
    finally:
        eng.close()


def test_graph_replay_equals_plain_launches(monkeypatch):
    """spdm_sample_run replays each denoise step as a hipGraph; the trajectory (every iterate) must be bit-identical
    to the plain-launch loop, for both schedulers, and a changed session (new seed) must rebuild the graph."""
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler
    H, D, B, T = 16, 3, 3, 12
    cond_dim = 1350
    sd = random_state_dict(cond_dim, seed=5, attention=True)
    g = torch.Generator().manual_seed(11)
    cond = torch.randn(B, 1, 10, 135, generator=g).cuda()
    x_T = torch.rand(B, 1, H, D, generator=g).cuda()
    inpaint = (torch.rand(B, 1, 1, D, generator=g) * 2 - 1).cuda()
    for kind in (DDPMScheduler, DDIMScheduler):
        eng = make_engine(H, D, cond_dim, B, sd, T=T)
        sched = kind(num_train_timesteps=T)
        sched.set_timesteps(T)
        eng.set_scheduler(sched)
        res = {}
        for mode in ("graph", "plain"):
            eng.set_switch("SPDM_NO_GRAPH", mode == "plain")
            for seed in (3, 4):      # device Philox noise: the seed lives in a device word, so both seeds replay ONE graph
                x0, hist = eng.sample(cond, x_T, noise=None, inpaint=inpaint, seed=seed, history=True)
                res[(mode, seed)] = (x0.cpu(), hist.cpu())
        for seed in (3, 4):
            assert torch.equal(res[("graph", seed)][0], res[("plain", seed)][0])
            assert torch.equal(res[("graph", seed)][1], res[("plain", seed)][1])
        if kind is DDPMScheduler:
            assert not torch.equal(res[("graph", 3)][0], res[("graph", 4)][0])
        eng.close()


def test_error_behaviour():
    B, H, D, cd = 2, 16, 3, 14
    sd = weights(cd, 5)
    eng = make_engine(H, D, cd, B, sd)
    try:
        x = torch.zeros(3, 1, H, D).cuda()
        with pytest.raises(RuntimeError, match="max_batch"):
            eng.unet_forward(x, [0], None)
        with pytest.raises(RuntimeError, match="outside"):
            eng.unet_forward(x[:1], [1000], None)
        with pytest.raises(RuntimeError, match="scheduler"):
            eng.sample(None, x[:2])
        with pytest.raises(RuntimeError, match="already loaded"):
            eng.load_state_dict(sd)
    finally:
        eng.close()
    with pytest.raises(KeyError):
        from state_policy_diffusionmodel_amd.engine import SpdmEngine
        e2 = SpdmEngine(H, D, cd, max_batch=1)
        try:
            e2.load_state_dict({k: v for k, v in sd.items() if not k.startswith("sa3")})
        finally:
            e2.close()


def test_split_range_guard_weights(monkeypatch):
    """|w| >= 511 cannot be represented by the split-fp16 weight format (x 2^7 -> fp16 inf).  spdm_load_weights must keep
    such a layer on the exact fp32 kernels: still the oracle's result, never inf.  One outlier each in a conv of the
    wide kernel, a conv of the level-3 path, a Linear of the fused C = 64 attention block, of the C = 128 tail kernel,
    of the C = 256 GEMM chain, a time-embedding Linear and a FiLM encoder."""
    H, D, B, cd = 32, 3, 3, 33
    base = weights(cd, 21)
    sd = {k: v.copy() for k, v in base.items()}
    hit = ["up3.doubleConv1.first.weight", "bot2.second.weight", "sa6.ff_self.1.weight", "sa1.attention.out_proj.weight",
           "sa2.attention.in_proj_weight", "down2.emb_layer.1.weight", "up1.cond_encoder.2.weight"]
    for i, k in enumerate(hit):
        flat = sd[k].reshape(-1)
        flat[(7 * i + 3) % flat.size] = 600.0 if i % 2 == 0 else -650.0
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 1, H, D, generator=g)
    y = torch.randn(B, 1, 3, 11, generator=g)
    t = torch.tensor([5, 400, 999])
    want = unet_film_forward(sd, x, t, y).numpy()
    eng = make_engine(H, D, cd, B, sd)
    try:
        assert eng.split_precision and eng.demoted_tensors >= len(hit)
        got = eng.unet_forward(x.cuda(), t, y.cuda()).cpu().numpy()
        assert np.isfinite(got).all() and not eng.nonfinite()
        assert np.abs(got - want).max() <= TOL * max(1.0, np.abs(want).max())
    finally:
        eng.close()
    eng = make_engine(H, D, cd, B, base)
    try:
        assert eng.demoted_tensors == 0
    finally:
        eng.close()


def test_split_range_guard_activations():
    """Activations beyond the split format's range (|x| > 4094) overflow to inf in the fp16 hi part.  The library must
    say so (device flag -> FloatingPointError from sample(), nonfinite() after unet_forward) instead of returning
    garbage with SPDM_OK; the exact-fp32 engine handles the same model."""
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler
    H, D, B, cd = 16, 3, 2, 33
    sd = {k: v.copy() for k, v in weights(cd, 21).items()}
    for k in sd:                       # FiLM scale/bias x 3e5: the up-path activations leave the fp16 range
        if "cond_encoder.2" in k:
            sd[k] = sd[k] * 3.0e5
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, 1, H, D, generator=g)
    y = torch.randn(B, 1, 3, 11, generator=g)
    want = unet_film_forward(sd, x, torch.tensor([7]), y).numpy()
    assert np.isfinite(want).all()
    eng = make_engine(H, D, cd, B, sd)
    try:
        got = eng.unet_forward(x.cuda(), [7], y.cuda()).cpu().numpy()
        ok = np.isfinite(got).all() and np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max())
        assert ok or eng.nonfinite(), "silent garbage: neither correct nor flagged"
        if not ok:
            s = DDIMScheduler(num_train_timesteps=50)
            s.set_timesteps(5)
            eng.set_scheduler(s)
            with pytest.raises(FloatingPointError):
                eng.sample(y.cuda(), torch.rand(B, 1, H, D).cuda())
    finally:
        eng.close()
    exact = make_engine(H, D, cd, B, sd, exact_fp32=True)
    try:
        got = exact.unet_forward(x.cuda(), [7], y.cuda()).cpu().numpy()
        assert not exact.nonfinite()
        assert np.abs(got - want).max() <= 1e-3 * max(1.0, np.abs(want).max())
    finally:
        exact.close()


@pytest.mark.parametrize("H,D", [(32, 3), (64, 6), (24, 5)], ids=["h32d3", "h64d6", "h24d5"])
def test_every_batch_regime_agrees_with_single_trajectory_runs(H, D):
    """Kernel selection (tile sizes, split-K, the few-row kernel, 128- vs 256-row tiles, width-2 rules ...) depends on the
    batch.  Sweep batches across every regime boundary and require each probed trajectory of the batch to equal the same
    trajectory evaluated alone (itself oracle-checked above) to fp32 rounding.  A statistics-layout or tiling bug in any one
    branch shows here as a 1e-3-sized deviation (this is the test that caught one)."""
    cd = 33
    sd = weights(cd, 21)
    Bs = [2, 3, 5, 8, 9, 16, 33, 64, 100, 129, 200, 256, 300, 400, 511, 512, 700, 1024, 1500, 2048]
    if H == 64:
        Bs = [2, 5, 8, 17, 64, 127, 128, 256, 300, 512, 1024]
    Bmax = max(Bs)
    g = torch.Generator().manual_seed(H * 10 + D)
    x = torch.randn(Bmax, 1, H, D, generator=g).cuda()
    y = torch.randn(Bmax, 1, 3, 11, generator=g).cuda()
    eng = make_engine(H, D, cd, Bmax, sd)
    try:
        probes = sorted({0, 1, 2, Bmax // 3, Bmax - 1})
        single = {i: eng.unet_forward(x[i:i + 1], [321], y[i:i + 1]).cpu()[0] for i in probes}
        for B in Bs:
            got = eng.unet_forward(x[:B], [321], y[:B]).cpu()
            assert not eng.nonfinite()
            for i in probes:
                if i < B:
                    d = float((got[i] - single[i]).abs().max())
                    assert d <= 5e-6, (B, i, d)
            last = eng.unet_forward(x[B - 1:B], [321], y[B - 1:B]).cpu()[0]       # the batch's last trajectory (ragged tiles)
            assert float((got[B - 1] - last).abs().max()) <= 5e-6, (B, "last")
    finally:
        eng.close()


def test_pinned_geometry_makes_a_shard_bit_identical_to_the_whole_batch():
    """Kernel selection follows the batch (small-grid kernel, split-K, tile sizes, fused sources), so a shard of a batch agrees
    with the whole-batch run only to fp32 rounding.  pin_geometry=True (SPDM_PIN_GEOMETRY) selects for max_batch whatever the
    call's batch: engines sized for the GLOBAL batch then reproduce it bit for bit on any aligned shard -- the contract the
    multi-GPU path offers callers that need it (Diffusion_DDPM.sample(..., shard_exact=True))."""
    from state_policy_diffusionmodel_amd.engine import SpdmEngine
    from state_policy_diffusionmodel_amd.schedulers import DDPMScheduler
    B, H, D, cd, T = 192, 32, 3, 33, 4
    sd = weights(cd, 21)
    g = torch.Generator().manual_seed(4)
    cond = torch.randn(B, 1, 3, 11, generator=g).cuda()
    x_T = torch.rand(B, 1, H, D, generator=g).cuda()
    res = {}
    for pin in (False, True):
        eng = SpdmEngine(H, D, cd, max_batch=B, num_train_timesteps=T, pin_geometry=pin)
        eng.load_state_dict(sd)
        s = DDPMScheduler(num_train_timesteps=T)
        s.set_timesteps(T)
        eng.set_scheduler(s)
        try:
            full = eng.sample(cond, x_T, seed=9).cpu()
            parts = [eng.sample(cond[a:b], x_T[a:b], seed=9, sample_offset=a).cpu() for a, b in ((0, 64), (64, 128), (128, 192))]
            res[pin] = (full, torch.cat(parts))
        finally:
            eng.close()
    assert torch.equal(res[True][0], res[True][1])                              # pinned: bit for bit
    assert float((res[False][0] - res[False][1]).abs().max()) <= 1e-5           # default: fp32 rounding
    assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-5            # and both are the same trajectories


def test_in_kernel_film_coefficients_match_the_coefficient_launch():
    """The attention kernels may evaluate the FiLM-tail coefficients of their samples themselves (default at batch <= 4,
    SPDM_FILM_LOCAL=1 forces it).  Forced on at batches and horizons where a row tile straddles several samples (L = 48, 12, 3
    at horizon 24), it must agree with the film_coef_kernel path."""
    cd = 33
    sd = weights(cd, 21)
    for H, D, B in ((24, 5, 37), (32, 3, 70), (8, 2, 9)):
        g = torch.Generator().manual_seed(H + B)
        x = torch.randn(B, 1, H, D, generator=g).cuda()
        y = torch.randn(B, 1, 3, 11, generator=g).cuda()
        t = (torch.arange(B) * 13) % 1000
        eng = make_engine(H, D, cd, B, sd)
        try:
            ref = eng.unet_forward(x, t, y).cpu()
            eng.set_switch("SPDM_FILM_LOCAL", True)
            got = eng.unet_forward(x, t, y).cpu()
            assert not eng.nonfinite()
            assert float((got - ref).abs().max()) <= 2e-6, (H, D, B)
        finally:
            eng.close()


def test_fused_attention_head_kernel_matches_the_three_launch_path():
    """sa_head_kernel (LayerNorm + in_proj + attention core in one kernel: q, k, v never reach memory) runs by itself only on
    large grids (>= 2048 row tiles: measured).  Forced on (SPDM_SA_HEAD=1) at sizes the oracle checks in seconds -- every tile
    geometry it supports (64 / 16 tokens at C = 128; 16 / 4 at C = 256; 32 / 8 at horizon 64), ragged batches, per-sample t --
    it must match the oracle and the default path."""
    cd = 33
    sd = weights(cd, 21)
    for H, D, B in ((32, 3, 37), (64, 6, 5), (16, 3, 9)):
        g = torch.Generator().manual_seed(H * 7 + B)
        x = torch.randn(B, 1, H, D, generator=g)
        y = torch.randn(B, 1, 3, 11, generator=g)
        t = (torch.arange(B) * 29) % 1000
        want = unet_film_forward(sd, x, t, y).numpy()
        eng = make_engine(H, D, cd, B, sd)
        try:
            ref = eng.unet_forward(x.cuda(), t, y.cuda()).cpu().numpy()
            eng.set_switch("SPDM_SA_HEAD", True)
            got = eng.unet_forward(x.cuda(), t, y.cuda()).cpu().numpy()
            assert not eng.nonfinite()
            assert np.abs(got - want).max() <= TOL, (H, D, B)
            assert np.abs(got - ref).max() <= 5e-6, (H, D, B)
        finally:
            eng.close()

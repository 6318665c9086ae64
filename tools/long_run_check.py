"""1000-step DDPM at B = 256 (BASELINE config 2 geometry) through the facade: hipGraph replay vs plain launches -- wall time,
finiteness and bit-equality of the final sample.  Run on the GPU box: python tools/long_run_check.py"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM, load_model
g = torch.Generator().manual_seed(0)
B, oh = 256, 10
m = Diffusion_DDPM(noise_steps=1000, obs_horizon=oh, pred_horizon=31, observation_dim=135, prediction_dim=3,
                   model="UNet_Film", inpaint_horizon=1, weight_seed=0, max_batch=B)
batch = {"position": torch.rand(B, oh, 2, generator=g) * 2 - 1, "velocity": torch.rand(B, oh, 2, generator=g),
         "action": torch.rand(B, oh, 3, generator=g), "image_features": torch.randn(B, oh, 128, generator=g)}
# prediction_dim=3 -> inpaint vector needs 3 dims: use explicit tensors
obs = {"obs_cond": torch.randn(B, oh, 135, generator=g), "inpaint": torch.rand(B, 1, 3, generator=g) * 2 - 1}
x_T = torch.rand(B, 1, 32, 3, generator=g).cuda()
m.noise_steps = 4
m.sample({k: v.clone() for k, v in obs.items()}, x_T=x_T.clone(), batched=True, seed=5)   # warm-up: engine, weights, tables
m.noise_steps = 1000
res = {}
for mode in ("graph", "plain"):
    m._engine.set_switch("SPDM_NO_GRAPH", mode == "plain")      # (the environment is read once, at engine creation)
    torch.cuda.synchronize(); t0 = time.time()
    out = m.sample({k: v.clone() for k, v in obs.items()}, x_T=x_T.clone(), batched=True, seed=5)
    torch.cuda.synchronize(); el = time.time() - t0
    res[mode] = out.cpu()
    print(mode, "1000-step DDPM, B=256, H=32, D=3:", round(el, 3), "s =", round(el, 3), "ms/step", "nonfinite flag", m._engine.nonfinite(), "finite", bool(torch.isfinite(out).all()), "absmax", float(out.abs().max()))
print("graph == plain bit for bit:", torch.equal(res["graph"], res["plain"]))

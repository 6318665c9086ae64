#!/usr/bin/env python3
"""Generate the committed golden vectors under tests/golden/.

Runs ONLY in the build container (needs /root/reference).  It imports the
reference U-Net *module* (`models/Unet_FiLmLayer.py`, `_noAttention.py`) with a
stub for the unused top-level `import torchvision`, loads OUR deterministic
generated weights into it (`load_state_dict(strict=True)` -> pins the tensor
inventory key-for-key), runs it on seeded inputs and stores inputs + outputs.
Trajectory fixtures are produced by the oracle's loop restatement
(oracle/scheduler_ref.py) DRIVING THE IMPORTED REFERENCE U-NET.

Only data (inputs, expected outputs, seeds, a weight-blob hash) is written; no
reference source or bytecode is copied.  Weights are not stored: the GPU box
regenerates them from the seed (state_policy_diffusionmodel_amd/weights.py)
and checks the hash.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from state_policy_diffusionmodel_amd.weights import random_state_dict, blob_sha256
from oracle.scheduler_ref import sample_loop

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    if "torchvision" not in sys.modules:
        sys.modules["torchvision"] = types.ModuleType("torchvision")  # unused by the path
    sys.path.insert(0, REF)
    from models.Unet_FiLmLayer import UNet_Film
    from models.Unet_FiLmLayer_noAttention import UNet_Film_noAttention
    return UNet_Film, UNet_Film_noAttention


def build_reference_model(cond_dim, seed, attention=True):
    UNet_Film, UNet_Film_noAttention = import_reference()
    cls = UNet_Film if attention else UNet_Film_noAttention
    m = cls(in_channels=1, out_channels=1, noise_steps=1000, time_dim=256, global_cond_dim=cond_dim)
    sd = random_state_dict(cond_dim, seed=seed, attention=attention)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    m.eval()
    return m, sd


def gen(seed):
    return torch.Generator().manual_seed(seed)


def forward_case(name, H, D, B, obs_h, obs_dim, t_list, wseed=0, attention=True, per_sample_t=False,
                 with_taps=False):
    cond_dim = obs_h * obs_dim
    m, sd = build_reference_model(cond_dim, wseed, attention)
    x = torch.rand(B, 1, H, D, generator=gen(100 + H + D))
    y = torch.randn(B, 1, obs_h, obs_dim, generator=gen(200 + B))
    out = {"x": x.numpy(), "cond": y.numpy(), "H": H, "D": D, "B": B, "obs_h": obs_h,
           "obs_dim": obs_dim, "wseed": wseed, "attention": int(attention),
           "weights_sha256": blob_sha256(sd)}
    ts, eps = [], []
    with torch.no_grad():
        for t in t_list:
            tt = torch.tensor(t if per_sample_t else [t], dtype=torch.int64)
            ts.append(np.asarray(tt))
            eps.append(m(x, tt, y).numpy())
    out["t"] = np.stack(ts)
    out["eps"] = np.stack(eps)
    if with_taps:
        # block-level intermediates via forward hooks on the reference modules (NCHW)
        taps = {}
        hooks = []
        names = ["inc", "down1", "sa1", "down2", "sa2", "down3", "sa3", "bot1", "bot2", "bot3",
                 "up1", "sa4", "up2", "sa5", "up3", "sa6"]
        for n in names:
            if hasattr(m, n):
                hooks.append(getattr(m, n).register_forward_hook(
                    lambda mod, i, o, n=n: taps.__setitem__(n, o.detach().numpy().copy())))
        with torch.no_grad():
            m(x, torch.tensor([t_list[0]]), y)
        for h in hooks:
            h.remove()
        for n, v in taps.items():
            out["tap_" + n] = v
    path = os.path.join(OUT, f"unet_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def trajectory_case(name, kind, T, N, H, D, B, obs_h=10, obs_dim=135, inp_h=1, wseed=0, attention=True):
    cond_dim = obs_h * obs_dim
    m, sd = build_reference_model(cond_dim, wseed, attention)
    x_T = torch.rand(B, 1, H, D, generator=gen(2))                      # uniform, ddpm.py:252
    cond = torch.randn(B, 1, obs_h, obs_dim, generator=gen(1))
    noise = torch.randn(N, B, 1, H, D, generator=gen(3))
    inpaint = torch.rand(B, 1, inp_h, D, generator=gen(4)) * 2 - 1 if inp_h > 0 else None
    hist = sample_loop(lambda x, t, y: m(x, t, y), kind, T, N, cond, x_T,
                       noise if kind == "ddpm" else None, inpaint, history=True)
    out = {"kind": kind, "T": T, "N": N, "H": H, "D": D, "B": B, "obs_h": obs_h, "obs_dim": obs_dim,
           "inp_h": inp_h, "wseed": wseed, "attention": int(attention),
           "weights_sha256": blob_sha256(sd),
           "x_T": x_T.numpy(), "cond": cond.numpy(), "noise": noise.numpy(),
           "history": np.stack([h.numpy() for h in hist])}
    if inpaint is not None:
        out["inpaint"] = inpaint.numpy()
    path = os.path.join(OUT, f"traj_{name}.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def encoder_case(name, n, wseed=5, iseed=11):
    """The observation encoder (models/encoder/autoencoder.py:7-20): import the reference ``Autoencoder`` class -- its module
    needs ``pytorch_lightning`` and ``torchvision`` only for names it never touches on this path (the LightningModule
    base of the *training* wrapper, an unused ``models`` import), so both are stubbed exactly like ``torchvision`` above --
    load OUR generated tensors into its ``.encoder`` with strict=True (pins the key inventory) and record its output."""
    from oracle.encoder_ref import make_encoder_state_dict
    if "pytorch_lightning" not in sys.modules:
        pl = types.ModuleType("pytorch_lightning")
        pl.LightningModule = torch.nn.Module          # base class of the training wrapper defined further down the file
        sys.modules["pytorch_lightning"] = pl
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    if not hasattr(tv, "models"):
        tv.models = types.ModuleType("torchvision.models")
    sys.modules["torchvision"] = tv
    sys.modules.setdefault("torchvision.models", tv.models)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.encoder.autoencoder import Autoencoder
    ae = Autoencoder()
    sd = make_encoder_state_dict(wseed)
    ae.encoder.load_state_dict(sd, strict=True)
    ae.eval()
    images = torch.rand(n, 3, 96, 96, generator=gen(iseed))          # frames are in [0,1]
    with torch.no_grad():
        latent = ae.encoder(images)
    path = os.path.join(OUT, f"encoder_{name}.npz")
    # images are regenerated from the seed by the tests (torch's CPU generator is deterministic); a checksum pins them
    np.savez_compressed(path, n=n, wseed=wseed, iseed=iseed, latent=latent.numpy(),
                        images_sum=np.float64(images.double().sum().item()), first_image_row=images[0, 0, 0].numpy(),
                        weights_sum=np.float64(sum(v.double().sum().item() for v in sd.values())))
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--encoder-only" in sys.argv:
        encoder_case("n5", 5)
        return
    torch.set_num_threads(8)
    # U-Net forward, the shape set of SURVEY.md section 8(c)
    forward_case("h16d3_b1", 16, 3, 1, 10, 135, [0, 1, 50, 99])
    forward_case("h32d3_b2", 32, 3, 2, 10, 135, [500, 0, 999], with_taps=True)
    forward_case("h31d5_b2", 31, 5, 2, 10, 135, [7, 500])
    forward_case("h64d6_b1", 64, 6, 1, 10, 135, [999, 3])
    forward_case("h32d3_b3_tvec", 32, 3, 3, 10, 135, [[0, 500, 999]], per_sample_t=True)
    forward_case("h32d3_b2_noattn", 32, 3, 2, 10, 135, [250], attention=False, with_taps=True)
    forward_case("h40d2_b2_smallcond", 40, 2, 2, 2, 7, [123], wseed=5)
    # trajectories: oracle loop driving the imported reference U-Net
    trajectory_case("ddpm_T20_h16d3_b2", "ddpm", 20, 20, 16, 3, 2)
    trajectory_case("ddim_T10_h16d3_b2", "ddim", 10, 10, 16, 3, 2)
    trajectory_case("ddim_T100_n10_h32d3_b1", "ddim", 100, 10, 32, 3, 1)
    trajectory_case("ddpm_T12_h32d3_b2_inp4", "ddpm", 12, 12, 32, 3, 2, inp_h=4)
    encoder_case("n5", 5)


if __name__ == "__main__":
    main()

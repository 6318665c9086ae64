"""CLI with the reference's ``generate.py`` surface (``/root/reference/generate.py:10-21,44-83``):
same flags, same "*** Time taken for sampling" print around ``model.sample(..., option='sample_history')``.

The reference reads a private Lightning checkpoint, STATS.pkl and a zarr dataset that are not
part of its repository (``.gitignore:12,21,23``); here ``--checkpoint`` takes a ``.pt``/``.ckpt``
whose ``state_dict`` holds the ``noise_estimator.*`` tensors (read with
``torch.load(weights_only=True)``), and without one the model is random-init and the
observation batch synthetic -- the timing path is identical.  The mp4 plotting
(``plt_toVideo``) is out of scope; ``--out`` saves the sampling history as .npy instead.
"""
from __future__ import annotations

import argparse
import time

import numpy as np
import torch

from .diffusion import load_model


def parse_arguments(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--model_name", type=str, default="DDIM", help="Can be 'DDPM', 'DDIM'")
    p.add_argument("--version", type=str, default="860", help="Version control variable")
    p.add_argument("--checkpoint_epoch", type=str, default="4", help="Checkpoint epoch")
    p.add_argument("--stats_file_name", type=str, default="STATS.pkl", help="Stats file name")
    p.add_argument("--dataset_dir", type=str, default="./data", help="Directory of the dataset")
    p.add_argument("--dataset_name", type=str, default="", help="Name of the dataset")
    p.add_argument("--batch_size", type=int, default=1, help="Batch size")
    p.add_argument("--seed", type=int, default=125, help="Random seed")
    # extensions
    p.add_argument("--checkpoint", type=str, default=None, help="state_dict file with noise_estimator.* tensors")
    p.add_argument("--hparams", type=str, default=None, help="hparams.yaml written next to the checkpoint (overrides the size flags)")
    p.add_argument("--model", type=str, default="UNet_Film")
    p.add_argument("--obs_horizon", type=int, default=10)
    p.add_argument("--pred_horizon", type=int, default=15)
    p.add_argument("--inpaint_horizon", type=int, default=1)
    p.add_argument("--observation_dim", type=int, default=135)
    p.add_argument("--prediction_dim", type=int, default=5)
    p.add_argument("--noise_steps", type=int, default=1000)
    p.add_argument("--ddim_steps", type=int, default=100)
    p.add_argument("--batched", action="store_true", help="sample all batch_size trajectories (reference: only the first)")
    p.add_argument("--out", type=str, default=None)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_arguments(argv)
    torch.manual_seed(args.seed)
    size = dict(noise_steps=args.noise_steps, obs_horizon=args.obs_horizon, pred_horizon=args.pred_horizon,
                observation_dim=args.observation_dim, prediction_dim=args.prediction_dim, model=args.model,
                inpaint_horizon=args.inpaint_horizon)
    if args.hparams:            # generate.py:46-58 of the reference: everything comes from tb_logs/version_*/hparams.yaml
        size = {}
    model = load_model(args.model_name, args.checkpoint or None, args.hparams, num_of_ddim_steps=args.ddim_steps,
                       max_batch=args.batch_size, **size)
    args.obs_horizon, args.observation_dim = model.obs_horizon, model.observation_dim
    B, oh = args.batch_size, args.obs_horizon
    g = torch.Generator().manual_seed(args.seed)
    batch = {"position": torch.rand(B, oh, 2, generator=g) * 2 - 1, "velocity": torch.rand(B, oh, 2, generator=g) * 2 - 1,
             "action": torch.rand(B, oh, 3, generator=g) * 2 - 1,
             "image_features": torch.randn(B, oh, max(args.observation_dim - 7, 0), generator=g)}
    observation_batch = model.prepare_observation_batch(batch)
    print(f"***Sampling with {args.model_name}...")
    start = time.time()
    sampling_history = model.sample(batch=observation_batch, option="sample_history", batched=args.batched)
    torch.cuda.synchronize()
    end = time.time()
    print(f"*** Time taken for sampling: {end - start} ***")
    if args.out:
        np.save(args.out, torch.stack(sampling_history).cpu().numpy())
    return sampling_history


if __name__ == "__main__":
    main()

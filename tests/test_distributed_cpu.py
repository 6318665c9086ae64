"""N > 1 path on CPU: world_size-2 (and 3, uneven shards) gloo groups run the batch-sharded
sampling driver with a stand-in per-rank sampler and must reproduce the single-process result,
rank-major, independent of the rank count."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import philox_ref
from state_policy_diffusionmodel_amd.distributed import all_gather_trajectories, sample_sharded, shard_bounds


def fake_sampler(cond, x_T, noise=None, inpaint=None, seed=0, sample_offset=0):
    """Deterministic per-trajectory function of (inputs, GLOBAL index): x_T + mean(cond) + Philox noise
    keyed by the global trajectory index -- the property the real device stream has."""
    B = x_T.shape[0]
    z = philox_ref.step_noise(seed, 0, sample_offset, B, x_T[0].numel())
    out = x_T + cond.reshape(B, -1).mean(dim=1).reshape(B, 1, 1, 1) + torch.from_numpy(z).reshape(x_T.shape)
    if inpaint is not None:
        out[:, :, : inpaint.shape[2], :] = inpaint
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(B, 1, 2, 7, generator=g)
    x_T = torch.rand(B, 1, 8, 3, generator=g)
    inpaint = torch.rand(B, 1, 1, 3, generator=g)
    out = sample_sharded(fake_sampler, cond, x_T, inpaint=inpaint, seed=99)
    s, e = shard_bounds(B, rank, world)
    part = all_gather_trajectories(torch.full((e - s, 2), float(rank)), B)
    if rank == 0:
        q.put((out.numpy(), part.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 8), (3, 7)])
def test_sharded_sampling_matches_single_process(world, B):
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(B, 1, 2, 7, generator=g)
    x_T = torch.rand(B, 1, 8, 3, generator=g)
    inpaint = torch.rand(B, 1, 1, 3, generator=g)
    want = fake_sampler(cond, x_T, inpaint=inpaint, seed=99, sample_offset=0).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, part = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(got, want)                  # rank-count invariant, bit for bit
    owners = np.concatenate([np.full(shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0], r) for r in range(world)])
    np.testing.assert_array_equal(part[:, 0], owners)         # rank-major order


class _FakeEngine:
    """The engine's loop interface (sample_begin / sample_run / sample_result, history) with the fake per-trajectory arithmetic."""
    n_steps = 3

    def sample_begin(self, cond, x_T, noise=None, inpaint=None, seed=0, sample_offset=0, history=False):
        self.x = x_T.clone()
        self.cond, self.inpaint, self.seed, self.off = cond, inpaint, seed, sample_offset
        self.hist = [self.x.clone()] if history else None
        return torch.stack(self.hist) if history else None

    def sample_run(self, a, b):
        for i in range(a, b):
            z = philox_ref.step_noise(self.seed, i, self.off, self.x.shape[0], self.x[0].numel())
            self.x = 0.5 * self.x + self.cond.reshape(self.x.shape[0], -1).mean(dim=1).reshape(-1, 1, 1, 1) + torch.from_numpy(z).reshape(self.x.shape)
            if self.hist is not None:
                self.hist.append(self.x.clone())

    def sample_result(self):
        return self.x


def _sampler_worker(rank, world, port, B, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from state_policy_diffusionmodel_amd.distributed import ShardedSampler
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(B, 1, 2, 7, generator=g)
    x_T = torch.rand(B, 1, 8, 3, generator=g)
    eng = _FakeEngine()
    ss = ShardedSampler(eng)
    ss.begin(cond, x_T, seed=5)
    ss.run(0, 2)
    mid = ss.result()
    ss.run(2, 3)
    out = ss.result()
    if rank == 0:
        q.put((mid.numpy(), out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,B", [(2, 6), (3, 7)])
def test_sharded_sampler_steps_and_gathers_like_one_process(world, B):
    """distributed.ShardedSampler -- what Diffusion_DDPM.sample(batched=True) and bench.py drive -- over step ranges: begin on
    GLOBAL tensors, run(a, b) on the shard, result() gathers all ranks' iterates, rank-major, at any point of the loop."""
    g = torch.Generator().manual_seed(0)
    cond = torch.randn(B, 1, 2, 7, generator=g)
    x_T = torch.rand(B, 1, 8, 3, generator=g)
    ref = _FakeEngine()
    ref.sample_begin(cond, x_T, seed=5)
    ref.sample_run(0, 2)
    want_mid = ref.sample_result().numpy().copy()
    ref.sample_run(2, 3)
    want = ref.sample_result().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    mid, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_array_equal(mid, want_mid)
    np.testing.assert_array_equal(out, want)

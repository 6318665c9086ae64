"""GPU tests of the batch-sharded sampling path with the REAL engine (BASELINE.json configs[3]):
  * world size 1: ShardedSampler / Diffusion_DDPM.sample(batched=True, sharded=True) equal the plain engine call;
  * world size 2 on ONE device (gloo rendezvous, both ranks on cuda:0 -- the 1-GPU box's rehearsal of the 8-GPU job):
    every rank returns all trajectories, equal to the single-process run to fp32 rounding, and the step graph is
    captured once per rank.
The RCCL transport itself needs more than one physical GPU and is exercised by the driver's scaling run; what is
checked here is everything else on that path: shard bounds, global noise keying, gather order, the product entry."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(noise_steps=6, obs_horizon=3, pred_horizon=14, observation_dim=11, prediction_dim=5, model="UNet_Film",
          inpaint_horizon=2, weight_seed=4)


def _batch(B, obs_h, seed):
    g = torch.Generator().manual_seed(seed)
    return {"position": torch.rand(B, obs_h, 2, generator=g) * 2 - 1, "velocity": torch.rand(B, obs_h, 2, generator=g),
            "action": torch.rand(B, obs_h, 3, generator=g), "image_features": torch.randn(B, obs_h, 4, generator=g)}


def _x_T(B):
    return torch.rand(B, 1, 16, 5, generator=torch.Generator().manual_seed(99))


def test_sharded_entry_at_world_size_one_equals_the_plain_call():
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    from state_policy_diffusionmodel_amd.distributed import ShardedSampler, sample_sharded
    B = 7
    m = Diffusion_DDPM(max_batch=B, **KW)
    obs = m.prepare_observation_batch(_batch(B, 3, 1))
    plain = m.sample(dict(obs), batched=True, x_T=_x_T(B).cuda(), seed=11)
    shard = m.sample(dict(obs), batched=True, sharded=True, x_T=_x_T(B).cuda(), seed=11)
    assert torch.equal(plain, shard)
    hist = m.sample(dict(obs), option="sample_history", batched=True, sharded=True, x_T=_x_T(B).cuda(), seed=11)
    assert len(hist) == 7 and torch.equal(hist[-1], plain)
    eng = m._engine
    cond = m.prepare_obs_cond_vectors(obs).unsqueeze(1)
    inp = m.prepare_inpaint_vectors(obs).unsqueeze(1)
    a = ShardedSampler(eng).sample(cond, _x_T(B).cuda(), inpaint=inp, seed=11)
    b = sample_sharded(eng.sample, cond, _x_T(B).cuda(), inpaint=inp, seed=11)
    assert torch.equal(a, plain) and torch.equal(b, plain)
    with pytest.raises(ValueError):
        m.sample(dict(obs), sharded=True)                      # B = 1 form has nothing to shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, B, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    import torch.distributed as dist
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = Diffusion_DDPM(**KW)
        obs = m.prepare_observation_batch(_batch(B, 3, 1))
        out = m.sample(dict(obs), batched=True, x_T=_x_T(B).cuda(), seed=11)          # sharded=None: auto (group is up)
        out2 = m.sample(dict(obs), batched=True, x_T=_x_T(B).cuda(), seed=11)         # fresh tensors: replay, no re-capture
        torch.manual_seed(5 + rank)                                                   # ranks disagree on purpose ...
        drawn = m.sample(dict(obs), batched=True)                                     # ... x_T and seed come from rank 0
        q.put((rank, out.cpu().numpy(), bool(torch.equal(out, out2)), m._engine.graph_captures, m._engine._B,
               drawn.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [8, 7])
def test_two_ranks_on_one_device_reproduce_the_single_process_run(B):
    import torch.multiprocessing as mp
    from state_policy_diffusionmodel_amd.diffusion import Diffusion_DDPM
    m = Diffusion_DDPM(max_batch=B, **KW)
    obs = m.prepare_observation_batch(_batch(B, 3, 1))
    want = m.sample(dict(obs), batched=True, x_T=_x_T(B).cuda(), seed=11).cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, B, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            r, out, same, captures, local_b, drawn = q.get(timeout=300)
            got[r] = (out, same, captures, local_b, drawn)
    finally:
        for p in procs:
            p.join(timeout=120)
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    for r in (0, 1):
        out, same, captures, local_b, drawn = got[r]
        assert out.shape == want.shape
        assert np.abs(out - want).max() <= 1e-5, r            # every rank holds ALL trajectories, rank-major
        assert same and captures == 1                         # a second call with fresh tensors replays the captured step
        assert local_b == (B + 1 - r) // 2                    # contiguous, balanced shards (uneven for B = 7)
    np.testing.assert_array_equal(got[0][0], got[1][0])
    np.testing.assert_array_equal(got[0][4], got[1][4])       # rank 0's x_T / seed were broadcast

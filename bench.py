#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one denoise iteration of the sampling loop over one batch of synthetic
trajectories: U-Net eps prediction + scheduler update + inpaint (SURVEY.md section 8d
"sample-step", times the batch).  Workload at N = 1: BASELINE.json configs[1..3] geometry at the
metric's batch -- DDPM T = 1000, B = 4096 trajectories per GPU, horizon 32, state_dim 3,
cond (10 x 135), attention on, inpaint_horizon 1, random-init weights (seed 0), device Philox
noise.  Inputs are resident in HBM before the timed region.  Multi-GPU: the batch dimension is
sharded (independent trajectories, weak scaling: 4096 per GPU), no collective inside the loop,
one RCCL all-gather of the iterates closes the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=4096, help="trajectories per GPU")
    p.add_argument("--horizon", type=int, default=32)
    p.add_argument("--state-dim", type=int, default=3)
    p.add_argument("--kind", default="ddpm", choices=["ddpm", "ddim"])
    p.add_argument("--train-steps", type=int, default=1000)
    p.add_argument("--no-attention", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-batch", type=int, default=64)
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    return p.parse_args()


def host_cores() -> int:
    """CPU cores this process may actually use: min(affinity mask, cgroup CPU quota).  On the GPU
    box the affinity mask shows all 256 hardware threads but the container's share is 16 CPUs;
    running 256 torch threads against a 16-CPU quota measures throttling, not the CPU."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(round(int(quota) / int(period)))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
            if q > 0:
                n = min(n, max(1, int(round(q / per))))
        except Exception:
            pass
    return n


def cpu_baseline(args, sd, cond_dim):
    """The oracle (torch-CPU restatement of the reference path) timed on this box's host cores on a
    bounded sample of the same workload: same geometry, batch `cpu_batch`, as many full denoise
    steps as fit in ~cpu_seconds."""
    import torch
    from oracle.scheduler_ref import LinearBetaSchedule, ddpm_step, ddim_step
    from oracle.unet_film_ref import unet_film_forward
    cores = host_cores()
    torch.set_num_threads(cores)
    B, H, D = args.cpu_batch, args.horizon, args.state_dim
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(B, 1, 10, cond_dim // 10, generator=g)
    x = torch.rand(B, 1, H, D, generator=g)
    sd_t = {k: torch.from_numpy(v) for k, v in sd.items()}
    s = LinearBetaSchedule(args.train_steps)
    attention = not args.no_attention

    def one_step(x, t):
        eps = unet_film_forward(sd_t, x, torch.tensor([t]), cond, attention=attention)
        if args.kind == "ddpm":
            return ddpm_step(s, eps, t, x, torch.randn(x.shape, generator=g))
        return ddim_step(s, eps, t, x)

    x = one_step(x, args.train_steps - 1)          # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        x = one_step(x, args.train_steps - 2 - n)
        n += 1
        el = time.perf_counter() - t0
        if el >= args.cpu_seconds or n >= 400:
            break
    return {"value": B * n / el, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement), batch {B}, {n} full denoise steps "
                      f"(U-Net + {args.kind} update) in {el:.1f} s, {cores} threads"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from state_policy_diffusionmodel_amd.engine import SpdmEngine
    from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler
    from state_policy_diffusionmodel_amd.weights import random_state_dict

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using {world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    B, H, D = args.batch, args.horizon, args.state_dim
    obs_h, obs_dim = 10, 135
    cond_dim = obs_h * obs_dim
    K, W = args.steps, args.warmup
    T = args.train_steps
    if K + W > T:
        raise SystemExit(f"steps + warmup ({K + W}) exceed the schedule length ({T})")
    attention = not args.no_attention

    sd = random_state_dict(cond_dim, seed=0, attention=attention)
    eng = SpdmEngine(H, D, cond_dim, max_batch=B, device=local, attention=attention, num_train_timesteps=T)
    eng.load_state_dict(sd)
    sched = (DDPMScheduler if args.kind == "ddpm" else DDIMScheduler)(num_train_timesteps=T)
    sched.set_timesteps(T)
    eng.set_scheduler(sched)

    # synthetic inputs, global trajectory index = rank * B + b  (seeded per global batch)
    g = torch.Generator().manual_seed(1000 + rank)
    cond = torch.randn(B, 1, obs_h, obs_dim, generator=g).to(dev)
    x_T = torch.rand(B, 1, H, D, generator=g).to(dev)
    inpaint = (torch.rand(B, 1, 1, D, generator=g) * 2 - 1).to(dev)
    gathered = torch.empty((world * B, 1, H, D), device=dev) if world > 1 else None

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    eng.sample_begin(cond, x_T, noise=None, inpaint=inpaint, seed=7, sample_offset=rank * B)
    eng.sample_run(0, W)                                   # untimed warm-up steps
    if world > 1:
        dist.all_gather_into_tensor(gathered, x_T)         # untimed: RCCL sets its channels up on the first call of a kind
    eng.profile(True)                                      # HIP events around every run of consecutive conv3x3 launches
    barrier()
    t0 = time.perf_counter()
    eng.sample_run(W, W + K)                               # EXACTLY K denoise steps
    out = eng.sample_result()
    if world > 1:
        dist.all_gather_into_tensor(gathered, out)         # RCCL over xGMI, closes the timed region
    barrier()
    el = time.perf_counter() - t0
    launches, conv_ms, conv_flops = eng.profile_read()
    eng.profile(False)
    # the same K steps once more WITHOUT the per-launch events: the product's default path replays each step as a
    # hipGraph (spdm_sample_run), which the event-instrumented region above cannot -- reported beside, never as `value`
    el_graph = None
    if W + 2 * K <= T:
        barrier()
        t1 = time.perf_counter()
        eng.sample_run(W + K, W + 2 * K)
        barrier()
        el_graph = time.perf_counter() - t1
    if not bool(torch.isfinite(out).all()):
        raise SystemExit("bench: non-finite iterate")

    tmax = torch.tensor([el], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax.item())

    if rank == 0:
        value = world * B * K / el
        # Roofline of the dominant kernel class (all conv3x3 implicit-GEMM launches of the timed steps).
        # algorithmic = 2 * MACs the launches actually evaluate (level-3 zero taps skipped), fp32-equivalent.
        # The default path issues 3 fp16 MFMAs per fp32-equivalent product (split precision, fp32 accumulate),
        # so the matrix pipe executes 3x the algorithmic FLOPs and is priced against the DENSE fp16 MFMA peak
        # (1024 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz); the exact path is priced against the fp32 MFMA peak.
        algo = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        split = eng.split_precision
        achieved = algo * 3.0 if split else algo
        peak = 2516.6 if split else 157.3
        kernel = ("conv3x3_wide_kernel (3 x v_mfma_f32_16x16x32_f16 per K=32) / conv_gemm_kernel (3 x v_mfma_f32_32x32x16_f16 per K=16): "
                  "3x3 implicit GEMM, split-fp16 operands, fp32 accumulate" if split
                  else "conv_gemm_kernel (3x3 implicit GEMM; v_mfma_f32_32x32x2_f32)")
        traffic = None      # HBM bytes per launch of the same kernel class, from the committed rocprofv3 PMC passes
        try:
            with open(os.path.join(ROOT, "profiles", "r01_roofline_traffic.json")) as fh:
                traffic = float(json.load(fh)["traffic_bytes_per_launch"])
        except Exception:
            pass
        line = {
            "metric": "denoise-steps/sec (B=4096, horizon=32)",
            "value": value, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "contraction_path": "split-fp16 MFMA, fp32 accumulate" if eng.split_precision else "fp32 MFMA",
            "batch_steps_per_s": K / el,
            "config": {"workload": f"{args.kind.upper()} T={T}, {B} trajectories/GPU, horizon {H}, state_dim {D}, "
                                   f"cond {obs_h}x{obs_dim}, UNet_Film attention {'on' if attention else 'off'}, "
                                   f"inpaint_horizon 1, random-init weights, device Philox noise",
                       "global_batch": world * B, "horizon": H, "state_dim": D, "parallelism": f"batch-shard x{world}"},
            "roofline": {"bound": "mfma", "kernel": kernel,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": "profiles/r01_roofline_traffic.json (rocprofv3 PMC, B=4096)" if traffic else None,
                         "algorithmic_fp32_equiv_tflops": algo, "fp32_mfma_peak_tflops": 157.3,
                         "launches": launches, "avg_launch_ms": conv_ms / max(launches, 1),
                         "share_of_step_time": conv_ms / (el * 1e3)},
        }
        if el_graph is not None:
            line["graph_replay"] = {"ms_per_step": el_graph / K * 1e3, "value": world * B * K / el_graph,
                                    "note": "same K steps replayed as a hipGraph per step, no per-launch events (rank 0 clock)"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, sd, cond_dim)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- denoise-steps/sec of the MI355X hot path (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8                      # starts its own 8 ranks (one per GPU) when WORLD_SIZE is unset
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one denoise iteration of the sampling loop over one batch of synthetic
trajectories: U-Net eps prediction + scheduler update + inpaint (SURVEY.md section 8d
"sample-step", times the batch).

Workload.  N = 1: BASELINE.json configs[1..3] geometry at the metric's batch -- DDPM T = 1000, 4096
trajectories, horizon 32, state_dim 3, cond (10 x 135), attention on, inpaint_horizon 1, random-init
weights (seed 0), device Philox noise.  N > 1: BASELINE.json configs[3] as written -- the SAME global batch
of 4096 trajectories sharded over the N GPUs (4096 / N per rank, "scaling": "strong"); the weak-scaling
figure (4096 per GPU) is measured in the same process and reported beside it (`weak_scaling`), never as
`value`.  `--scaling weak` makes the weak figure the headline instead.  Inputs are resident in HBM before
the timed region.  The batch dimension shards with no collective inside the loop; one RCCL all-gather of
the iterates closes the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

OBS_H, OBS_DIM = 10, 135
TRAFFIC_GLOB = os.path.join("profiles", "r*_roofline_traffic*.json")      # rocprofv3 PMC passes (tools/pmc_traffic.py), one file per geometry


def find_traffic(B, H, D, attention, kind):
    """The committed PMC traffic summary whose geometry is this run's (latest round first); None if there is none."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, TRAFFIC_GLOB)), reverse=True):
        try:
            with open(path) as fh:
                t = json.load(fh)
        except Exception:
            continue
        c = t.get("config", {"batch": 4096, "horizon": 32, "state_dim": 3, "kind": "ddpm", "attention": True})
        if (c.get("batch"), c.get("horizon"), c.get("state_dim")) == (B, H, D) and bool(c.get("attention", True)) == attention \
                and "traffic_bytes_per_launch" in t:
            t["_file"] = os.path.relpath(path, ROOT)
            return t
    return None


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--batch", type=int, default=4096, help="trajectories per GPU of the weak-scaling workload (and of N = 1)")
    p.add_argument("--global-batch", type=int, default=None,
                   help="total trajectories of the strong-scaling workload (default: --batch, i.e. 4096)")
    p.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                   help="which figure is `value` for N > 1 (auto: strong = BASELINE config 4)")
    p.add_argument("--horizon", type=int, default=32)
    p.add_argument("--state-dim", type=int, default=3)
    p.add_argument("--kind", default="ddpm", choices=["ddpm", "ddim"])
    p.add_argument("--train-steps", type=int, default=1000)
    p.add_argument("--no-attention", action="store_true")
    p.add_argument("--instrument-steps", type=int, default=2,
                   help="how many of the K timed steps carry per-launch HIP events (the rest replay as hipGraphs, the product's path); 0 = all")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-batch", type=int, default=None,
                   help="batch of the CPU-oracle sample (default: the run's per-GPU batch, capped at 1024: SURVEY 8d asks for the target batch)")
    p.add_argument("--cpu-seconds", type=float, default=15.0)
    p.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                   help="collective backend (gloo + --stub-engine: CPU rehearsal of the launcher and the timing protocol)")
    p.add_argument("--shared-device", action="store_true",
                   help="every rank uses cuda:0 (rehearsal of the N > 1 path with the real engine on a 1-GPU box; use with --backend gloo)")
    p.add_argument("--stub-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)     # tests: this rank exits 3 before the rendezvous
    p.add_argument("--launch-timeout", type=float, default=1500.0,
                   help="seconds the self-launcher waits for its ranks before it terminates them (N > 1 without torchrun)")
    p.add_argument("--stub-engine", action="store_true",
                   help="replace the HIP engine by a sleep (tests/test_bench_launcher.py only; the line says so)")
    return p.parse_args(argv)


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
    B, H, D = (args.cpu_batch or min(args.batch, 1024)), args.horizon, args.state_dim
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(B, 1, OBS_H, cond_dim // OBS_H, generator=g)
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
        if (el >= args.cpu_seconds and n >= 3) or n >= 400 or el >= 4 * args.cpu_seconds:
            break
    return {"value": B * n / el, "unit": "trajectory-steps/s", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32 restatement), batch {B}, {n} full denoise steps "
                      f"(U-Net + {args.kind} update) in {el:.1f} s, {cores} threads"}


# ---------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment starts the N ranks itself
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """Start `args.gpus` fresh child processes of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one rank
    per GPU) and relay rank 0's line.  The parent has not imported torch or touched the GPU: nothing is re-exec'ed.
    Non-zero exit if any rank fails -- never a silent single-rank result."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        # The host driver of this pool only supports dmabuf IPC: without HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's intra-node
        # transport fails with `hipIpcGetMemHandle: invalid argument`.  A caller's own setting wins.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's stdout is drained by a thread so that polling every child never blocks on a full pipe
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.launch_timeout
    rcs = [None] * n
    failed = False
    while any(rc is None for rc in rcs):
        for i, p in enumerate(procs):
            if rcs[i] is None:
                rcs[i] = p.poll()
        if any(rc not in (None, 0) for rc in rcs) or time.monotonic() > deadline:
            failed = True          # one rank died (or the job overran): the others would sit in the rendezvous / a barrier
            break
        time.sleep(0.05)
    if failed:
        for i, p in enumerate(procs):
            if rcs[i] is None:
                p.terminate()
        for i, p in enumerate(procs):
            if rcs[i] is None:
                try:
                    rcs[i] = p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    rcs[i] = p.wait()
        why = "timed out" if all(rc in (None, 0) or rc < 0 for rc in rcs) and time.monotonic() > deadline else "a rank failed"
        print(f"[bench] {why}; ranks exited with {rcs}", file=sys.stderr)
        return 1
    reader.join(timeout=10)
    for ln in "".join(out0).splitlines():  # rank 0's JSON line to stdout; anything a library printed goes to stderr
        (sys.stdout if ln.lstrip().startswith("{") else sys.stderr).write(ln + "\n")
    sys.stdout.flush()
    return 0


# ---------------------------------------------------------------------------------------------------
class StubEngine:
    """CPU stand-in with the engine's loop interface (tests/test_bench_launcher.py: launcher, rendezvous, timing
    protocol and JSON contract on gloo, world size 2).  One 'step' sleeps 1 ms per 1024 trajectories."""
    split_precision = True

    def __init__(self, H, D):
        self.H, self.D = H, D

    def sample_begin(self, cond, x_T, noise=None, inpaint=None, seed=0, sample_offset=0):
        self.x = x_T.clone()

    def sample_run(self, a, b):
        time.sleep(max(b - a, 0) * 1e-3 * max(self.x.shape[0] / 1024.0, 0.05))

    def sample_result(self):
        return self.x

    def profile(self, on):
        pass

    def profile_read(self):
        return 0, 0.0, 0.0

    def close(self):
        pass


def timed_run(eng, dist, world, dev, cond, x_T, inpaint, W, K, T, profile, sync, instr=2):
    """W untimed + exactly K timed denoise steps of one workload, all-gather included.  cond / x_T / inpaint are the GLOBAL
    batch (the same tensors on every rank); the product's own sharding driver (distributed.ShardedSampler) slices this
    rank's contiguous shard, runs the loop on it and all-gathers the iterates.  Returns (max-over-ranks seconds, conv
    launches, conv ms, conv flops, graph-replay seconds or None, final iterates of ALL ranks, instrumented steps)."""
    import torch
    from state_policy_diffusionmodel_amd.distributed import ShardedSampler
    ss = ShardedSampler(eng)

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
            sync()

    ss.begin(cond, x_T, noise=None, inpaint=inpaint, seed=7)
    ss.run(0, W)                                           # untimed warm-up steps
    if world > 1:
        ss.result()                                        # untimed: the backend sets its channels up on the first call of a kind
    # The timed region is EXACTLY K denoise steps: the first K - P the way the product runs them (spdm_sample_run replays each
    # step as a hipGraph), the last P with HIP events around every run of consecutive conv3x3 launches on the launch stream --
    # what `roofline` is measured from (events cannot sit inside a replayed graph, and instrumenting all K steps with plain
    # launches timed a path the product does not take: 1 % slower at 4096 trajectories per GPU, 8 % at 512).
    # --instrument-steps 0 instruments all K.
    # (a graph is captured on the first call of >= 3 steps: the warm-up must have done that, or nothing is replayed here)
    P = K if (not profile or instr <= 0 or K - instr < 3 or W < 3) else min(instr, K)
    if profile:
        eng.profile("prepare")                             # events created here, outside the timed region
    barrier()
    t0 = time.perf_counter()
    if K - P > 0:
        ss.run(W, W + K - P)
    if profile:
        eng.profile(True)
    ss.run(W + K - P, W + K)
    out = ss.result()                                      # N > 1: one all-gather (RCCL over xGMI) closes the timed region
    barrier()
    el = time.perf_counter() - t0
    launches = conv_ms = conv_flops = 0
    if profile:
        launches, conv_ms, conv_flops = eng.profile_read()
        eng.profile(False)
    # the same K steps once more WITHOUT the per-launch events: the product's default path replays each step as a
    # hipGraph (spdm_sample_run), which the event-instrumented region above cannot -- reported beside, never as `value`
    el_graph = None
    if profile and W + 2 * K <= T:
        barrier()
        t1 = time.perf_counter()
        ss.run(W + K, W + 2 * K)
        barrier()
        el_graph = time.perf_counter() - t1
    tmax = torch.tensor([el, el_graph if el_graph is not None else 0.0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    el = float(tmax[0].item())
    if el_graph is not None:
        el_graph = float(tmax[1].item())
    return el, launches, conv_ms, conv_flops, el_graph, out, (P if profile else 0)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))        # BEFORE any torch / HIP call in this process

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but WORLD_SIZE {world}: refusing to report a {world}-rank run as {args.gpus} GPUs")
    stub = args.stub_engine
    if stub and rank == args.stub_fail_rank:
        raise SystemExit(3)
    if stub:
        dev = torch.device("cpu")
        sync = lambda: None                                                     # noqa: E731
    else:
        if args.shared_device:
            local = 0
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        sync = lambda: torch.cuda.synchronize(dev)                              # noqa: E731
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    H, D = args.horizon, args.state_dim
    cond_dim = OBS_H * OBS_DIM
    K, W, T = args.steps, args.warmup, args.train_steps
    if K + W > T:
        raise SystemExit(f"steps + warmup ({K + W}) exceed the schedule length ({T})")
    attention = not args.no_attention
    Bw = args.batch                                        # weak scaling: per-GPU batch
    Bg = args.global_batch or args.batch                   # strong scaling: global batch
    headline = args.scaling if args.scaling != "auto" else ("strong" if world > 1 else "weak")
    if world > 1 and Bg % world != 0:
        raise SystemExit(f"global batch {Bg} does not divide over {world} ranks")
    Bs = Bg // world                                       # strong scaling: per-GPU batch

    sd = None
    if stub:
        eng = StubEngine(H, D)
    else:
        from state_policy_diffusionmodel_amd.engine import SpdmEngine
        from state_policy_diffusionmodel_amd.schedulers import DDIMScheduler, DDPMScheduler
        from state_policy_diffusionmodel_amd.weights import random_state_dict
        sd = random_state_dict(cond_dim, seed=0, attention=attention)
        eng = SpdmEngine(H, D, cond_dim, max_batch=max(Bw, Bs), device=local, attention=attention, num_train_timesteps=T)
        eng.load_state_dict(sd)
        sched = (DDPMScheduler if args.kind == "ddpm" else DDIMScheduler)(num_train_timesteps=T)
        sched.set_timesteps(T)
        eng.set_scheduler(sched)

    def inputs(Btot):
        # synthetic GLOBAL inputs, identical on every rank (seeded); ShardedSampler takes this rank's rows
        g = torch.Generator().manual_seed(1000)
        cond = torch.randn(Btot, 1, OBS_H, OBS_DIM, generator=g).to(dev)
        x_T = torch.rand(Btot, 1, H, D, generator=g).to(dev)
        inpaint = (torch.rand(Btot, 1, 1, D, generator=g) * 2 - 1).to(dev)
        return cond, x_T, inpaint

    runs = {}
    modes = ["weak"] if world == 1 else (["strong", "weak"] if headline == "strong" else ["weak", "strong"])
    for mode in modes:
        B = Bw if mode == "weak" else Bs
        cond, x_T, inpaint = inputs(world * B)
        el, launches, conv_ms, conv_flops, el_graph, out, instr_steps = timed_run(
            eng, dist, world, dev, cond, x_T, inpaint, W, K, T, profile=True, sync=sync, instr=args.instrument_steps)
        if tuple(out.shape) != (world * B, 1, H, D):
            raise SystemExit(f"bench: gathered iterates have shape {tuple(out.shape)}, expected {(world * B, 1, H, D)}")
        if not bool(torch.isfinite(out).all()):
            raise SystemExit("bench: non-finite iterate")
        del cond, x_T, inpaint, out
        runs[mode] = dict(B=B, el=el, launches=launches, conv_ms=conv_ms, conv_flops=conv_flops, el_graph=el_graph,
                          instr_steps=instr_steps)

    if rank == 0:
        r = runs[headline]
        B, el = r["B"], r["el"]
        value = world * B * K / el
        workload = (f"{args.kind.upper()} T={T}, {world * B} trajectories ({B} per GPU), horizon {H}, state_dim {D}, "
                    f"cond {OBS_H}x{OBS_DIM}, UNet_Film attention {'on' if attention else 'off'}, "
                    f"inpaint_horizon 1, random-init weights, device Philox noise")
        line = {
            "metric": f"denoise-steps/sec (B={world * B}, horizon={H})",
            "value": value, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": headline,
            "vs_baseline": None, "data": "synthetic",
            "batch_steps_per_s": K / el,
            "config": {"workload": workload, "global_batch": world * B, "per_gpu_batch": B, "horizon": H, "state_dim": D,
                       "parallelism": f"batch-shard x{world}" + (" (REHEARSAL: all ranks on one GPU)" if args.shared_device else "")},
        }
        if stub:
            line["dtype"] = "none (stub engine: launcher rehearsal, no kernels)"
            line["roofline"] = None
        else:
            split = eng.split_precision
            line["dtype"] = "f32 (split-fp16 MFMA operands, fp32 accumulate)" if split else "f32"
            line["contraction_path"] = "split-fp16 MFMA, fp32 accumulate" if split else "fp32 MFMA"
            line["roofline"] = roofline(args, r, split, B, H, D, el, K)
            if r["el_graph"] is not None:
                line["graph_replay"] = {"ms_per_step": r["el_graph"] / K * 1e3, "value": world * B * K / r["el_graph"],
                                        "note": "K more steps, ALL replayed as hipGraphs, no per-launch events (max over ranks); the timed "
                                                "region itself replays K - P steps and instruments the last P (roofline.instrumented_steps)"}
        for mode, rr in runs.items():
            if mode != headline:
                line[f"{mode}_scaling"] = {"value": world * rr["B"] * K / rr["el"], "ms_per_step": rr["el"] / K * 1e3,
                                           "global_batch": world * rr["B"], "per_gpu_batch": rr["B"], "unit": "trajectory-steps/s"}
        if world == 1 and not args.no_cpu_baseline and not stub:
            line["cpu_baseline"] = cpu_baseline(args, sd, cond_dim)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


def roofline(args, r, split, B, H, D, el, K):
    """Roofline of the dominant kernel class (all conv3x3 implicit-GEMM launches of the timed steps), live HIP-event time.

    achieved / frac: ALGORITHMIC fp32-equivalent FLOPs (2 x MACs the launches evaluate; zero-padding taps that are skipped
    do not count) / measured time, priced against the dense peak of the pipe the kernel runs on (fp16 MFMA, 2516.6 TF, on
    the default split-precision path -- which issues 3 fp16 MFMAs per product, so this fraction tops out at 1/3; fp32 MFMA,
    157.3 TF, on the exact path).  executed_mfma_*: the same with the 3 MFMAs counted (matrix-pipe utilisation).
    hbm_*: the metric's "% HBM roofline" for the WHOLE step: PMC bytes per step / step time / 8 TB/s; bytes come from the
    committed rocprofv3 PMC passes and are only reported when this run's geometry is the profiled one."""
    conv_ms, conv_flops, launches = r["conv_ms"], r["conv_flops"], r["launches"]
    algo = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    peak = 2516.6 if split else 157.3
    executed = algo * 3.0 if split else algo
    kernel = ("conv3x3_wide_kernel / conv_reg64_kernel (3 x v_mfma_f32_16x16x32_f16 per K=32) / conv_gemm_kernel (3 x v_mfma_f32_32x32x16_f16 per K=16): "
              "3x3 implicit GEMM, split-fp16 operands, fp32 accumulate" if split
              else "conv_gemm_kernel (3x3 implicit GEMM; v_mfma_f32_32x32x2_f32)")
    traffic = hbm_step = src = None
    t = find_traffic(B, H, D, not args.no_attention, args.kind)
    if t is not None:
        traffic = float(t["traffic_bytes_per_launch"])
        hbm_step = t.get("hbm_bytes_per_step")
        src = f"static: {t['_file']} (rocprofv3 PMC passes of this geometry; not re-measured in this run)"
    out = {"bound": "mfma", "kernel": kernel,
           "achieved": algo, "peak": peak, "unit": "TFLOP/s", "frac": algo / peak,
           "algorithmic_frac": algo / peak,
           "executed_mfma_tflops": executed, "executed_mfma_frac": executed / peak,
           "fp32_mfma_peak_tflops": 157.3,
           "traffic": traffic, "traffic_source": src,
           "launches": launches, "avg_launch_ms": conv_ms / max(launches, 1),
           "instrumented_steps": r.get("instr_steps", K), "graph_replay_steps": K - r.get("instr_steps", K),
           "share_of_step_time": (conv_ms / max(r.get("instr_steps", K), 1)) / (el / K * 1e3)}
    if hbm_step is not None:
        gbs = float(hbm_step) / (el / K) / 1e9
        out.update({"hbm_bytes_per_step": float(hbm_step), "hbm_achieved_GBps": gbs, "hbm_peak_GBps": 8000.0, "hbm_frac": gbs / 8000.0})
    else:
        out.update({"hbm_bytes_per_step": None, "hbm_achieved_GBps": None, "hbm_peak_GBps": 8000.0, "hbm_frac": None})
    return out


if __name__ == "__main__":
    main()

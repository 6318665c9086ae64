"""bench.py's multi-rank plumbing, rehearsed on CPU: `python bench.py --gpus 2` must start two ranks BY ITSELF
(gloo, stub engine: no kernels), report n_gpus = 2 with the strong-scaling workload (BASELINE config 4: one global
batch sharded over the ranks) as `value` and the weak figure beside it, and must refuse -- non-zero exit -- to report
a run whose world size differs from --gpus."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_gpus_2_launches_two_ranks_and_reports_strong_scaling():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--backend", "gloo", "--steps", "3",
                        "--warmup", "1", "--batch", "64", "--horizon", "16"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3 and d["warmup"] == 1
    assert d["config"]["global_batch"] == 64 and d["config"]["per_gpu_batch"] == 32
    assert d["config"]["parallelism"] == "batch-shard x2"
    assert abs(d["value"] - 64 * 1000.0 / d["ms_per_step"]) <= 1e-6 * d["value"]
    w = d["weak_scaling"]
    assert w["global_batch"] == 128 and w["per_gpu_batch"] == 64 and w["value"] > 0
    assert "B=64" in d["metric"] and "horizon=16" in d["metric"]


def test_weak_headline_on_request():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "64", "--scaling", "weak"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["scaling"] == "weak" and d["config"]["global_batch"] == 128 and d["strong_scaling"]["global_batch"] == 64


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "8"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "refusing" in r.stderr
    assert r.stdout.strip() == ""


def test_single_rank_stub_line():
    r = subprocess.run([sys.executable, BENCH, "--stub-engine", "--steps", "2", "--warmup", "1", "--batch", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip())
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["global_batch"] == 8


def test_launcher_terminates_the_other_ranks_when_one_dies():
    """A rank that dies before the rendezvous must not leave rank 0 waiting for the backend's own timeout: the launcher
    polls every child, terminates the rest on the first failure and exits non-zero."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--stub-engine", "--backend", "gloo", "--steps", "2",
                        "--warmup", "1", "--batch", "8", "--stub-fail-rank", "1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1
    assert "a rank failed" in r.stderr
    assert time.monotonic() - t0 < 60
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]     # never a line from a broken job

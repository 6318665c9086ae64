"""bench.py's output contract (one JSON line with the driver's keys, `roofline` and `cpu_baseline`), on a small workload."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "64", "--steps", "3", "--warmup", "1",
                        "--cpu-batch", "2", "--cpu-seconds", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"].startswith("f32")
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 64 * 1000.0 / d["ms_per_step"]) <= 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_frac",
              "executed_mfma_frac", "hbm_frac", "hbm_bytes_per_step"):
        assert k in rf, k
    assert rf["frac"] == rf["algorithmic_frac"]                      # `frac` is the ALGORITHMIC fraction (task contract)
    assert abs(rf["executed_mfma_frac"] - 3.0 * rf["frac"]) <= 1e-9  # split path: 3 fp16 MFMAs per product
    assert rf["traffic"] is None and rf["hbm_frac"] is None          # batch 64 is not the profiled geometry: no static figure
    assert "B=64" in d["metric"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) <= 1e-9 and 0.0 < rf["frac"] < 1.0
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1


def test_traffic_summary_is_found_by_geometry():
    """`roofline.traffic` / `hbm_*` come from the committed PMC summaries; bench.py must pick the file whose recorded geometry is
    the run's (one hard-coded file name), and none for a geometry that was never profiled."""
    sys.path.insert(0, ROOT)
    import bench
    head = bench.find_traffic(4096, 32, 3, True, "ddpm")
    assert head is not None and head["config"]["horizon"] == 32 and head["traffic_bytes_per_launch"] > 1e8
    c5 = bench.find_traffic(4096, 64, 6, True, "ddim")
    assert c5 is not None and c5["config"]["state_dim"] == 6 and c5["hbm_bytes_per_step"] > head["hbm_bytes_per_step"]
    assert c5["_file"] != head["_file"]
    assert bench.find_traffic(64, 32, 3, True, "ddpm") is None
    assert bench.find_traffic(4096, 32, 3, False, "ddpm") is None        # the no-attention model was not profiled

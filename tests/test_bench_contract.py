"""bench.py's one JSON line: the keys the driver and the judge read (a small batch, one step; the CPU leg and the side legs are switched off — the default run has them)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_line_carries_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1", "--windows", "128", "--distinct", "8", "--distinct-lidar", "2",
           "--ragged-windows", "0", "--converging-windows", "0", "--td-windows", "0", "--no-latency", "--no-pcie", "--no-stress-leg", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0 and d["ms_per_step"] > 0
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms", "kernels_ms_per_step", "binding", "window_kernels", "whole_iteration"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert rf["avg_launch_ms"] > 0 and all(v >= 0 for v in rf["kernels_ms_per_step"].values())
    # the step is the sum of its launch groups (nothing of the frame runs outside a timed group; profiling waits add a little)
    assert 0.5 * d["ms_per_step"] < sum(rf["kernels_ms_per_step"].values()) < 1.5 * d["ms_per_step"]

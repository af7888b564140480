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


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line"
    return json.loads(lines[0])


def test_bench_gpus_2_starts_two_ranks_and_prints_one_line():
    """`python bench.py --gpus 2` — the shape of the driver's command — must itself start two rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one per
    GPU) and relay rank 0's line. CPU rehearsal of the plumbing (VILF_BENCH_DRYRUN=1: gloo, no GPU, no product code): both ranks reach the pose gather."""
    env = dict(os.environ, VILF_BENCH_DRYRUN="1", VILF_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2
    assert d["gather"]["ranks_in_table"] == 2 and d["gather"]["rows"] == 8 and d["gather"]["global_unit_order"] is True
    # under a launcher (WORLD_SIZE set) bench.py is a rank and starts nothing itself
    env1 = dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env1, capture_output=True, text=True, timeout=300)
    assert r1.returncode == 0, r1.stderr[-2000:]
    assert _one_json_line(r1.stdout)["n_gpus"] == 1


@pytest.mark.gpu
def test_bench_gpus_2_rehearsal_on_one_gpu():
    """the real bench line through the N = 2 code path on the one-GPU box (VILF_BENCH_REHEARSAL=1: both ranks on cuda:0, gloo collectives — never a measurement):
    `python bench.py --gpus 2` starts the ranks itself, the line says n_gpus 2 and the gathered table holds both shards in global unit order."""
    env = dict(os.environ, VILF_BENCH_REHEARSAL="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None); env.pop("VILF_BENCH_DRYRUN", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "64", "--distinct", "4", "--distinct-lidar", "2",
           "--ragged-windows", "0", "--converging-windows", "0", "--td-windows", "0", "--no-latency", "--no-pcie", "--no-stress-leg", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0
    g = d["gather"]
    assert g["ranks_in_table"] == 2 and g["rows"] == 128 and g["global_unit_order"] is True

# the general path: its parity tests (+ the pose graph, which uses the same Cholesky), then group timing with the per-group device times
mkdir -p gpurun_out/lwg
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_pose_graph.py -x -q -m gpu -k "large_window or group or td_estimation or extrinsic or time_limit or max_solver or pose_graph or loop" > gpurun_out/lwg/tests.txt 2>&1; echo "tests rc $?"; tail -5 gpurun_out/lwg/tests.txt
python - <<P
import time, numpy as np, torch
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
from vil_fusion_amd.lib import default_options
opts = default_options(); opts.window_size = 50
wins = [synth.make_window(900 + 7 * k, opts, synth.SynthConfig(n_frames=51, n_features=2500, with_prior=False))[0] for k in range(4)]
s = BackendSolver(opts)
for S in (1, 8, 32):
    wl = [wins[k % 4] for k in range(S)]
    s.optimization_group(wl)
    s.set_profiling(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): r = s.optimization_group(wl)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    s.set_profiling(True)
    p0 = s.get_profile_large_window()
    s.optimization_group(wl)
    p1 = s.get_profile_large_window()
    s.set_profiling(False)
    print("S", S, "wall ms", 1e3 * dt, "it/s", sum(x.summary["num_iterations"] for x in r) / dt, {k: round(p1[k]["ms"] - p0[k]["ms"], 3) for k in p1})
P

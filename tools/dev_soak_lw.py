"""Soak check of the general single-window path (development tool, not collected by pytest): N random windows of 12..16 frames — feature count, state noise, constant
fraction — through vilf_window_solve (trust-region loop on the device) and through the oracle: iteration / accepted-step / linear-solve counts, termination, states."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
import oracle_lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(77)
bad = 0; rej = 0; conv = 0; worst = [0.0, 0.0]; tg = 0.0
for i in range(N):
    o = oracle_lib.default_options()
    nf = int(rng.integers(12, 17)); o.window_size = nf - 1
    if i % 2: o.max_num_iterations = 40                     # long budgets: the solves end by a tolerance (function / parameter / gradient), not by the iteration count
    nz = float(rng.choice([0.0, 0.05, 0.2, 0.8, 2.0, 3.0]))
    c = synth.SynthConfig(n_frames=nf, n_features=int(rng.integers(20, 200)), with_prior=False, const_fraction=float(rng.choice([0.0, 0.3])),
                          state_noise=(nz, np.deg2rad(8.0 * nz), nz))
    w, _, _ = synth.make_window(770000 + i, o, c)
    ref = oracle_lib.window_solve(o, w, None)
    s = BackendSolver(o)
    t = time.time(); got = s.optimization(w); tg += time.time() - t
    s.close()
    keys = ("num_iterations", "num_successful_steps", "num_linear_solves", "termination")
    a = tuple(got.summary[k] for k in keys); b = tuple(ref.summary[k] for k in keys)
    rej += b[1] < b[0]; conv += b[3] != 0
    dP = float(np.abs(got.Ps - ref.Ps).max()); dc = abs(got.summary["final_cost"] - ref.summary["final_cost"]) / max(ref.summary["final_cost"], 1e-30)
    worst[0] = max(worst[0], dP); worst[1] = max(worst[1], dc)
    if a != b or dP > 1e-4 or dc > 1e-4:
        bad += 1
        print("MISMATCH window", i, "frames", nf, "noise", nz, "counts", a, b, "dP", dP, "rel cost", dc)
print(f"{N} windows, {rej} with rejected steps, {conv} terminated by a tolerance, {bad} mismatches; worst |dP| {worst[0]:.3e} m, worst relative cost difference {worst[1]:.3e}; GPU {tg/N*1e3:.1f} ms per solve")

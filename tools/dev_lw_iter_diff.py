"""general path vs oracle, iteration by iteration: the same window solved with budgets of 1, 2, ... iterations — where do the two part, and by how much?"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
import oracle_lib
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 17
rng = np.random.default_rng(77)
for i in range(idx + 1):
    o = oracle_lib.default_options()
    nf = int(rng.integers(12, 17)); o.window_size = nf - 1
    if i % 2: o.max_num_iterations = 40
    nz = float(rng.choice([0.0, 0.05, 0.2, 0.8, 2.0, 3.0]))
    c = synth.SynthConfig(n_frames=nf, n_features=int(rng.integers(20, 200)), with_prior=False, const_fraction=float(rng.choice([0.0, 0.3])), state_noise=(nz, np.deg2rad(8.0 * nz), nz))
    w, _, _ = synth.make_window(770000 + i, o, c)
print("window", idx, "frames", nf, "features", w.n_features, "noise", nz, "const", c.const_fraction)
for it in list(range(1, 13)) + [20, 30, 40]:
    o.max_num_iterations = it
    ref = oracle_lib.window_solve(o, w, None)
    s = BackendSolver(o); got = s.optimization(w); s.close()
    print(it, "cost", got.summary["final_cost"], ref.summary["final_cost"], "rel %.2e" % (abs(got.summary["final_cost"] - ref.summary["final_cost"]) / ref.summary["final_cost"]),
          "dP %.2e" % np.abs(got.Ps - ref.Ps).max(), "succ", got.summary["num_successful_steps"], ref.summary["num_successful_steps"], "radius %.6e %.6e" % (got.summary["final_radius"], ref.summary["final_radius"]))

"""Soak check (development tool, not collected by pytest): N random windows of mixed shape — feature count, prior or none, marginalization flag, state noise, constant fraction —
solved as ONE batch on the HIP path and one by one by the oracle; solve -> marginalize -> compare the iteration / accepted-step / linear-solve counts, the states and the new priors."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
import oracle_lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(2026)
s = BackendSolver(); o = s.options
wins, priors, cfgs = [], [], []
for i in range(N):
    nz = float(rng.choice([0.05, 0.05, 0.2, 0.8, 2.0]))
    c = synth.SynthConfig(n_features=int(rng.integers(3, 330)), with_prior=bool(rng.random() < 0.8), const_fraction=float(rng.choice([0.0, 0.4, 0.4, 1.0])),
                          marginalization_flag=(0 if rng.random() < 0.7 else 1), state_noise=(nz, np.deg2rad(10.0 * nz), nz))
    w, p, _ = synth.make_window(910000 + i, o, c)
    wins.append(w); priors.append(p); cfgs.append(c)
s.batch_upload(wins, priors)
t = time.time(); s.batch_solve(); s.batch_marginalize(); dt = time.time() - t
res, sums = s.batch_download(), s.batch_summaries()
bad = 0; worst = [0.0, 0.0, 0.0]; rej = 0
for i in range(N):
    ref = oracle_lib.window_solve(o, wins[i], priors[i] if cfgs[i].with_prior else None)
    a = (sums[i].num_iterations, sums[i].num_successful_steps, sums[i].num_linear_solves)
    b = (ref.summary["num_iterations"], ref.summary["num_successful_steps"], ref.summary["num_linear_solves"])
    rej += b[1] < b[0]
    dP = float(np.abs(res[i].Ps - ref.Ps).max()); dc = abs(sums[i].final_cost - ref.summary["final_cost"]) / max(ref.summary["final_cost"], 1e-30)
    worst[0] = max(worst[0], dP); worst[1] = max(worst[1], dc)
    if a != b or dP > 1e-4 or dc > 1e-4:
        bad += 1
        print("MISMATCH window", i, "counts", a, b, "dP", dP, "rel cost", dc, vars(cfgs[i]) if hasattr(cfgs[i], '__dict__') else cfgs[i])
print(f"{N} windows, {rej} with rejected steps, {bad} mismatches; worst |dP| {worst[0]:.3e} m, worst relative cost difference {worst[1]:.3e}; GPU solve + marginalize {dt*1e3:.1f} ms")

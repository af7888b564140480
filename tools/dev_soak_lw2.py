"""Soak of the general path after round 4's rebuild: random windows of 12..60 frames with wide ranges of feature count / track structure (few features, all constant,
none constant), with and without Ex_Pose / td as variables, alone and in groups — against the oracle: counts, termination, states; and run-to-run bit equality."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
import oracle_lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(4242)
bad = 0; worst = 0.0; nonrep = 0
cases = []
for i in range(N):
    o = oracle_lib.default_options()
    nf = int(rng.choice([12, 13, 17, 24, 33, 48, 60])); o.window_size = nf - 1
    ext = int(rng.integers(0, 4))            # 0 plain, 1 td, 2 extrinsic, 3 both
    o.estimate_td = 1 if ext in (1, 3) else 0; o.estimate_extrinsic = 1 if ext in (2, 3) else 0
    nfeat = int(rng.choice([1, 5, 40, 150, 400, 900]))
    c = synth.SynthConfig(n_frames=nf, n_features=nfeat, with_prior=False, const_fraction=float(rng.choice([0.0, 0.4, 1.0])), state_noise=(0.05, np.deg2rad(0.5), 0.05))
    w, _, _ = synth.make_window(880000 + i, o, c)
    if o.estimate_td:
        w = synth.with_td_inputs(w, 10 + i)
    ref = oracle_lib.window_solve(o, w, None)
    s = BackendSolver(o)
    got = s.optimization(w); again = s.optimization(w)
    s.close()
    keys = ("num_iterations", "num_successful_steps", "num_linear_solves", "termination")
    a = tuple(got.summary[k] for k in keys); b = tuple(ref.summary[k] for k in keys)
    dP = float(np.abs(got.Ps - ref.Ps).max()); worst = max(worst, dP)
    rep = np.array_equal(got.Ps, again.Ps) and np.array_equal(got.para_feature, again.para_feature)
    if ext == 0 and not rep:
        nonrep += 1; print("NOT REPRODUCIBLE (plain path) window", i)
    if a != b or dP > 1e-5:
        bad += 1; print("MISMATCH window", i, "frames", nf, "features", nfeat, "ext", ext, "counts", a, b, "dP", dP)
    cases.append((o, w, ref, ext))
# groups of plain windows of mixed sizes
plain = [c for c in cases if c[3] == 0][:9]
if len(plain) >= 3:
    o = oracle_lib.default_options(); o.window_size = 12
    s = BackendSolver(o)
    got = s.optimization_group([c[1] for c in plain])
    s.close()
    for g, c in zip(got, plain):
        if g.summary["num_iterations"] != c[2].summary["num_iterations"] or np.abs(g.Ps - c[2].Ps).max() > 1e-5:
            bad += 1; print("GROUP MISMATCH", np.abs(g.Ps - c[2].Ps).max())
print(f"{N} windows ({sum(1 for c in cases if c[3])} with Ex_Pose / td variables), {bad} mismatches, {nonrep} non-reproducible plain solves; worst |dP| {worst:.3e} m")

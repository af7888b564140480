"""Dev check of k_linearize_split: the same windows solved with every window's linearisation over several workgroups (batches <= 32) and by one workgroup each
(VILF_NO_LIN_SPLIT=1): states, summaries and new priors must be identical to the bit. Then the single-window latency both ways."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
s = BackendSolver(); o = s.options
bad = 0; tot = 0
for B in (1, 2, 3, 4, 5, 6, 7, 8, 8, 8, 1, 1):
    wins, priors = [], []
    for i in range(B):
        nz = float(rng.choice([0.05, 0.2, 0.8, 2.0]))
        c = synth.SynthConfig(n_features=int(rng.integers(3, 330)), with_prior=bool(rng.random() < 0.8), const_fraction=float(rng.choice([0.0, 0.4, 1.0])),
                              marginalization_flag=(0 if rng.random() < 0.7 else 1), state_noise=(nz, np.deg2rad(10.0 * nz), nz))
        w, p, _ = synth.make_window(int(rng.integers(1, 10**6)), o, c)
        wins.append(w); priors.append(p)
    res = []
    for mode in (0, 1):
        if mode: os.environ["VILF_NO_LIN_SPLIT"] = "1"
        else: os.environ.pop("VILF_NO_LIN_SPLIT", None)
        s.batch_upload(wins, priors); s.batch_solve(); s.batch_marginalize()
        r = s.batch_download(); sm = s.batch_summaries()
        res.append((r, sm, [s.get_prior(i) for i in range(B)] if hasattr(s, "get_prior") else None))
    for i in range(B):
        a, b_ = res[0][0][i], res[1][0][i]
        same = all(np.array_equal(getattr(a, k), getattr(b_, k)) for k in ("Ps", "Rs", "Vs", "Bas", "Bgs", "para_feature")) and \
               (res[0][1][i].num_iterations, res[0][1][i].num_successful_steps, res[0][1][i].final_cost) == (res[1][1][i].num_iterations, res[1][1][i].num_successful_steps, res[1][1][i].final_cost)
        tot += 1
        if not same: bad += 1; print("DIFFERENT: batch", B, "window", i, "iterations", res[0][1][i].num_iterations, res[1][1][i].num_iterations, "dP", float(np.abs(a.Ps - b_.Ps).max()))
print("windows", tot, "different", bad)
os.environ.pop("VILF_NO_LIN_SPLIT", None)
win, prior, _ = synth.make_window(1000, o, synth.SynthConfig(n_features=230))
def med(f, n=30):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
def solve():
    s.set_prior(prior); return s.optimization(win)
solve(); print("single window, split: %.3f ms" % med(solve), "device usec", solve().summary["usec_solve"])
os.environ["VILF_NO_LIN_SPLIT"] = "1"
solve(); print("single window, one workgroup: %.3f ms" % med(solve), "device usec", solve().summary["usec_solve"])

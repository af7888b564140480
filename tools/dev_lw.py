import sys, time, os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np, oracle_lib as O
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
o = O.default_options(); o.window_size = 50
win, _, _ = synth.make_window(111, o, synth.SynthConfig(n_frames=51, n_features=2500, with_prior=False))
print("factors", len(win.obs_point) - win.n_features, "features", win.n_features)
t = time.perf_counter(); ref = O.window_solve(o, win, None); t_or = time.perf_counter() - t
s = BackendSolver(o)
s.optimization(win)
ts = []
for _ in range(5):
    t = time.perf_counter(); got = s.optimization(win); ts.append(time.perf_counter() - t)
print("oracle %.1f ms (%d it)  hip %.1f ms (%d it)  dP %.2e" % (1e3 * t_or, ref.summary["num_iterations"], 1e3 * min(ts), got.summary["num_iterations"], np.abs(got.Ps - ref.Ps).max()))

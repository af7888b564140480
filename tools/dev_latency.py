"""Dev measurement: single-frame latencies through the single-window / single-stream ABI (the reference's real-time use)."""
import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, Scan2Map
o = ol.default_options()
s = BackendSolver(o)
win, prior, _ = synth.make_window(1000, o, synth.SynthConfig(n_features=230))
def med(f, n=20):
    ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
def solve():
    s.set_prior(prior); return s.optimization(win)
solve()
print("window solve (upload + 8 iterations + download), ms:", round(med(solve), 3), " device-only usec:", solve().summary["usec_solve"])
def solve_marg():
    s.set_prior(prior); s.optimization(win); s.marginalize()
print("window solve + marginalize, ms:", round(med(solve_marg), 3))
t = time.perf_counter(); r = ol.window_solve(o, win, prior); t1 = time.perf_counter(); ol.window_marginalize(o, win, r, prior); t2 = time.perf_counter()
print("oracle (1 thread): solve ms", round(1e3 * (t1 - t), 2), "marginalize ms", round(1e3 * (t2 - t1), 2))
me, ms, scans, pl = synth.make_lidar_bench_case(7000)
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
m = Scan2Map(s); m.localMapInited(me, ms); m.set_pose(ident, pl)
t = time.perf_counter(); m.optimation_processing(*scans[0]); print("scan-to-map warm-up frame (raw map), ms:", round(1e3 * (time.perf_counter() - t), 3))
t = time.perf_counter(); r = m.optimation_processing(*scans[1]); print("scan-to-map steady-state frame, ms:", round(1e3 * (time.perf_counter() - t), 3), "queries", r.n_edge_ds + r.n_surf_ds)
mo = ol.OracleS2M(o); mo.init(me, ms); mo.set_pose(ident, pl); mo.step(*scans[0])
t = time.perf_counter(); mo.step(*scans[1]); print("oracle scan-to-map steady-state frame (1 thread), ms:", round(1e3 * (time.perf_counter() - t), 3))

"""Dev: time the batched scan-to-map step (config-3 LiDAR stage, steady-state maps) for S streams and check it against the oracle."""
import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, Scan2MapBatch
o = ol.default_options()
D = 2
raw = [synth.make_lidar_bench_case(100 + k) for k in range(D)]
ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
refs = []
for me, ms, scans, pl in raw:
    m = ol.OracleS2M(o); m.init(me, ms); m.set_pose(ident, pl); m.step(*scans[0])
    t = time.time(); r = m.step(*scans[1]); refs.append(r)
    print("oracle", round(time.time() - t, 3), "s  ds", r.n_edge_ds, r.n_surf_ds, "fac", list(r.n_edge_factors), list(r.n_surf_factors), "its", list(r.iterations), "map", r.map_edge_size, r.map_surf_size)
s = BackendSolver(o)
warm = Scan2MapBatch(s, D, max(len(c[2][0][0]) for c in raw) + 64, max(len(c[2][0][1]) for c in raw) + 64,
                     max(len(c[0]) + len(c[2][0][0]) for c in raw) + 64, max(len(c[1]) + len(c[2][0][1]) for c in raw) + 64)
for k, (me, ms, scans, pl) in enumerate(raw):
    warm.localMapInited(k, me, ms, None, pl); warm.set_scan(k, *scans[0])
warm.step(); wres = warm.results()
cases = [(warm.getMapCloud(k, 0), warm.getMapCloud(k, 1), raw[k][2][1], np.array(wres[k].pose_qt[:])) for k in range(D)]
print("steady-state map sizes", [(len(c[0]), len(c[1])) for c in cases], "scan", [(len(c[2][0]), len(c[2][1])) for c in cases])
for S in [int(a) for a in sys.argv[1:]] or [64]:
    t = time.time()
    b = Scan2MapBatch(s, S, max(len(c[2][0]) for c in cases) + 64, max(len(c[2][1]) for c in cases) + 64,
                      max(len(c[0]) + len(c[2][0]) for c in cases) + 64, max(len(c[1]) + len(c[2][1]) for c in cases) + 64)
    for i in range(S):
        me, ms, (se, ss), p1 = cases[i % D]
        b.localMapInited(i, me, ms, p1, ident); b.set_scan(i, se, ss)
    b.snapshot(); print("S", S, "setup", round(time.time() - t, 2), "s")
    for rep in range(3):
        b.rewind(); s.synchronize()
        t = time.time(); b.step(); dt = time.time() - t
        print("  step", round(dt * 1e3, 2), "ms ->", round(S / dt), "steps/s")
    got = b.results()
    bad = 0
    for i in range(S):
        g, r = got[i], refs[i % D]
        ok = (g.n_edge_ds, g.n_surf_ds, list(g.n_edge_factors), list(g.n_surf_factors), list(g.iterations), g.map_edge_size, g.map_surf_size) == (r.n_edge_ds, r.n_surf_ds, list(r.n_edge_factors), list(r.n_surf_factors), list(r.iterations), r.map_edge_size, r.map_surf_size)
        ok = ok and np.abs(np.array(g.pose_qt[:]) - np.array(r.pose_qt[:])).max() < 1e-9
        bad += 0 if ok else 1
    print("  mismatching streams vs oracle:", bad)
    if S == 512:
        s.set_profiling(True); b.rewind(); b.step(); print("  groups ms:", {k: round(v["ms"], 2) for k, v in s.get_profile_scan2map().items()}); s.set_profiling(False)

import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, FeatureExtraction
scene = synth.LidarScene(5, n_poles=80)
raw = scene.scan_raw(synth.euler_R(np.array(0.1), np.array(0.0), np.array(0.0)), np.array([-20.0, 3.0, scene.h]))
s = BackendSolver(); fe = FeatureExtraction(s)
fe.extractFeature(raw)
ts = []
for _ in range(20):
    t = time.perf_counter(); e, su = fe.extractFeature(raw); ts.append(time.perf_counter() - t)
t = time.perf_counter(); eo, so = ol.extract_features(raw); to = time.perf_counter() - t
print("points", len(raw), "edge", len(e), "surf", len(su), "| GPU median ms (H2D + kernels + D2H):", round(1e3 * float(np.median(ts)), 3), "| oracle ms:", round(1e3 * to, 3))

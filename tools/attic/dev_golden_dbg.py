import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
import oracle_lib as ol
from vil_fusion_amd import abi
from vil_fusion_amd.estimator import BackendSolver
o = ol.default_options()
win, prior, d = gu.load_window("window_all_const_depth")
s = BackendSolver(o); s.set_prior(prior); res = s.optimization(win); s.marginalize(); p = s.get_prior()
J, r, blocks = abi.prior_to_numpy(p)
print("gpu m", p.m, "n", p.n, [b["idx"] for b in blocks], [b["id"] for b in blocks])
print("ref m", int(d["newprior_m"]), d["newprior_J"].shape, list(d["newprior_idx"]), list(d["newprior_ids"]))
print("start frames", np.bincount(win.feature_start_frame), "const", win.feature_const.sum(), win.n_features)

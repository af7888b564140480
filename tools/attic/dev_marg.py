import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth, abi
from vil_fusion_amd.estimator import BackendSolver
o = ol.default_options()
s = BackendSolver(o)
for seed, wp, flag in ((21, False, 0), (22, True, 0), (23, True, 1)):
    win, prior, _ = synth.make_window(seed, o, synth.SynthConfig(with_prior=wp, marginalization_flag=flag, n_features=120))
    s.set_prior(prior if wp else None)
    got = s.optimization(win)
    t = time.time(); s.marginalize(); dt = time.time() - t
    pg = s.get_prior()
    ref = ol.window_solve(o, win, prior if wp else None)
    t = time.time(); pr = ol.window_marginalize(o, win, ref, prior if wp else None); dtr = time.time() - t
    print("seed", seed, "gpu prior n", pg.n, "blocks", pg.n_blocks, "ref n", pr.n, pr.n_blocks, "m", pr.m, "gpu ms", dt * 1e3, "cpu ms", dtr * 1e3)
    Jg, rg, bg = abi.prior_to_numpy(pg); Jr, rr, br = abi.prior_to_numpy(pr)
    print(" ids", [b["id"] for b in bg], [b["id"] for b in br])
    Lg, Lr = Jg.T @ Jg, Jr.T @ Jr
    print(" rel dLambda", np.abs(Lg - Lr).max() / np.abs(Lr).max(), "rel db", np.abs(Jg.T @ rg - Jr.T @ rr).max() / np.abs(Jr.T @ rr).max())

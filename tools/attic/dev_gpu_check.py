import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth, abi
from vil_fusion_amd.estimator import BackendSolver
np.set_printoptions(linewidth=200, precision=6)
o = ol.default_options()
s = BackendSolver(o)
for seed, wp in ((1, False), (2, True)):
    win, prior, truth = synth.make_window(seed, o, synth.SynthConfig(with_prior=wp))
    s.set_prior(prior if wp else None)
    t = time.time(); got = s.optimization(win); t = time.time() - t
    ref = ol.window_solve(o, win, prior if wp else None)
    print("seed", seed, "gpu", got.summary)
    print("         ref", ref.summary)
    print("  dP", np.abs(got.Ps - ref.Ps).max(), "dR", np.abs(got.Rs - ref.Rs).max(), "dV", np.abs(got.Vs - ref.Vs).max(), "dBa", np.abs(got.Bas-ref.Bas).max(), "dBg", np.abs(got.Bgs-ref.Bgs).max(),
          "ddepth", np.abs(1/got.para_feature - 1/ref.para_feature).max(), "wall", t)
    print(ol.last_trace()[:, [0, 1, 5, 6]])
# batch timing
for B in (64, 512, 2048):
    wins, priors = synth.make_batch(7, B, o, distinct=16)
    s.batch_upload(wins, priors)
    s.batch_solve()
    for rep in range(3):
        s.batch_rewind(); t = time.time(); s.batch_solve(); dt = time.time() - t
        sm = s.batch_summaries()
        its = sum(x.num_iterations for x in sm)
        print("B", B, "solve ms", dt * 1e3, "iterations", its, "iter/s", its / dt)

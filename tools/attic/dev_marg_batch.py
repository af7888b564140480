import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
s = BackendSolver(); o = s.options
B = 2048
wins, priors = synth.make_batch(1000, B, o, synth.SynthConfig(n_features=230), distinct=64)
for rep in range(3):
    s.batch_upload(wins, priors)
    t0 = time.perf_counter(); s.batch_solve(); t1 = time.perf_counter(); s.batch_marginalize(); t2 = time.perf_counter()
    print(f"solve {1e3*(t1-t0):.1f} ms  marginalize {1e3*(t2-t1):.1f} ms")
s.set_profiling(1)
s.batch_upload(wins, priors); s.batch_solve(); s.batch_marginalize()
print("marg profile", {k: round(v["ms"], 3) for k, v in s.get_profile_marginalize().items()})

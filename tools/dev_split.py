# the batch as n parts on n streams (VILF_SOLVE_SPLIT): does running the iteration's two kernels side by side help?
import sys, time, os
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BackendSolver(); opts = s.options
wins, priors = synth.make_batch(1000, B, opts, synth.SynthConfig(n_features=230), distinct=64)
s.batch_upload(wins, priors)
base = None
for split in ("", "2", "3", "4", "8"):
    if split: os.environ["VILF_SOLVE_SPLIT"] = split
    else: os.environ.pop("VILF_SOLVE_SPLIT", None)
    s.batch_rewind(); s.batch_solve()
    P = np.stack([r.Ps.ravel() for r in s.batch_download()])
    if base is None: base = P
    t = time.perf_counter()
    for _ in range(5):
        s.batch_rewind(); s.batch_solve(sync=True)
    dt = (time.perf_counter() - t) / 5
    print(f"split {split or '-':2s} ms/solve {dt*1e3:7.2f} identical results: {np.array_equal(P, base)}")

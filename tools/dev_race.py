"""Dev stress: re-run the same resident batch many times, count windows whose result differs from the first run."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 80
REPS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
s = BackendSolver()
o = s.options
wins, priors = synth.make_batch(5, B, o, synth.SynthConfig(n_features=NF), distinct=8)
s.batch_upload(wins, priors)
base = None
for rep in range(REPS):
    if rep: s.batch_rewind()
    s.batch_solve()
    res = s.batch_download()
    P = np.stack([r.Ps for r in res]); ic = np.array([r.summary["initial_cost"] for r in res]); fc = np.array([r.summary["final_cost"] for r in res])
    if base is None:
        base = (P, ic, fc)
        # also cross-check replicas of the same distinct window
        bad0 = [i for i in range(B) if not np.array_equal(P[i], P[i % 8])]
        print("rep 0 replicas differing:", len(bad0), bad0[:10])
        continue
    bad = [i for i in range(B) if not np.array_equal(P[i], base[0][i])]
    badic = [i for i in range(B) if ic[i] != base[1][i]]
    print("rep", rep, "diff windows:", len(bad), bad[:8], "init-cost diffs:", len(badic), ["%.1e" % np.abs(P[i] - base[0][i]).max() for i in bad[:4]])

"""solve a fixed mixed batch (different feature counts, with / without prior, 600 slots) under the current environment and dump every state array + the summaries:
python tools/dev_dump_solve.py out.npz ; python tools/dev_dump_solve.py --cmp a.npz b.npz"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if sys.argv[1] == "--cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    worst = 0.0; nd = 0
    for k in a.files:
        if not np.array_equal(a[k], b[k]):
            nd += 1; d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max(); worst = max(worst, d)
            print("differs:", k, "max |d| %.3e" % d)
    print("arrays differing:", nd, "of", len(a.files), "worst %.3e" % worst)
    sys.exit(0)
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
s = BackendSolver(); o = s.options
made = [synth.make_window(7101 + k, o, synth.SynthConfig(n_features=60 + 17 * k, with_prior=(k % 3 != 2))) for k in range(12)]
order = [(5 * i + i // 7) % 12 for i in range(600)]
s.batch_upload([made[k][0] for k in order], [made[k][1] for k in order]); s.batch_solve()
out = s.batch_download(); sm = s.batch_summaries()
d = {}
for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "para_pose", "para_speed_bias"):
    d[key] = np.stack([np.asarray(getattr(x, key)) for x in out])
d["feat"] = np.concatenate([np.asarray(x.para_feature).ravel() for x in out])
d["its"] = np.array([(x.num_iterations, x.num_successful_steps, x.termination, x.num_linear_solves) for x in sm])
d["cost"] = np.array([x.final_cost for x in sm])
s.batch_marginalize()
np.savez(sys.argv[1], **d)
print("dumped", sys.argv[1], "iterations", sorted(set(d["its"][:, 0].tolist())))

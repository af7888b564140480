"""k_iter (one launch per iteration) against the two-kernel sequence (VILF_NO_FUSED=1): states / summaries bit for bit on a mixed batch, then the solve time of a big batch.
usage: python tools/dev_fused_check.py [B]"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BackendSolver()
o = s.options
made = [synth.make_window(7101 + k, o, synth.SynthConfig(n_features=60 + 17 * k, with_prior=(k % 3 != 2))) for k in range(12)]
order = [(5 * i + i // 7) % 12 for i in range(600)]
def run(fused):
    if fused: os.environ["VILF_FUSED"] = "1"
    else: os.environ.pop("VILF_FUSED", None)
    s.batch_upload([made[k][0] for k in order], [made[k][1] for k in order]); s.batch_solve()
    out = s.batch_download(); sm = [(x.num_iterations, x.num_successful_steps, x.termination, x.final_cost, x.num_linear_solves) for x in s.batch_summaries()]
    return out, sm
a, sa = run(True); b, sb = run(False)
bad = sum(1 for x, y in zip(sa, sb) if x != y)
print("summaries equal:", sa == sb, "differing:", bad, "iterations", sorted(set(x[0] for x in sa)), "rejected somewhere:", any(x[0] != x[1] for x in sa))
nb = 0
for x, y in zip(a, b):
    for key in ("Ps", "Rs", "Vs", "Bas", "Bgs", "para_feature", "para_pose", "para_speed_bias"):
        if not np.array_equal(getattr(x, key), getattr(y, key)): nb += 1
print("state arrays differing:", nb, "max |dP|", max(np.abs(x.Ps - y.Ps).max() for x, y in zip(a, b)))
wins, priors = synth.make_batch(7, B, o, synth.SynthConfig(n_features=230), distinct=16)
for fused in (True, False, True, False):
    if fused: os.environ["VILF_FUSED"] = "1"
    else: os.environ.pop("VILF_FUSED", None)
    s.batch_upload(wins, priors); s.batch_solve()
    ts = []
    for rep in range(4):
        s.batch_rewind(); t = time.time(); s.batch_solve(); ts.append(time.time() - t)
    its = sum(x.num_iterations for x in s.batch_summaries())
    print("fused" if fused else "two-kernel", "B", B, "solve ms", ["%.3f" % (1e3 * t) for t in ts], "iter/s %.0f" % (its / min(ts)))
os.environ["VILF_FUSED"] = "1"; os.environ["VILF_NO_SLOTS"] = "1"
s.batch_upload(wins, priors); s.batch_solve()
ts = []
for rep in range(4):
    s.batch_rewind(); t = time.time(); s.batch_solve(); ts.append(time.time() - t)
print("fused, per-window workspaces (no slots)", ["%.3f" % (1e3 * t) for t in ts])

"""solve time of a resident batch under the current environment (VILF_SO / VILF_NO_FUSED / ...): python tools/dev_solve_time.py [B] [label]"""
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BackendSolver()
wins, priors = synth.make_batch(7, B, s.options, synth.SynthConfig(n_features=230), distinct=16)
s.batch_upload(wins, priors); s.batch_solve()
ts = []
for rep in range(6):
    s.batch_rewind(); t = time.time(); s.batch_solve(); ts.append(time.time() - t)
its = sum(x.num_iterations for x in s.batch_summaries())
print("%-28s B %d solve ms min %.3f med %.3f  iter/s %.0f" % (sys.argv[2] if len(sys.argv) > 2 else "", B, 1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], its / min(ts)))

"""Dev: the estimate_td batch of bench.py's R6 leg alone (64 windows through the general path as one group), for rocprofv3 --kernel-trace --stats; VILF_LW_TRACE=1 adds the host phases."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import torch
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
from vil_fusion_amd.lib import default_options
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
o2 = default_options(); o2.estimate_td = 1
ts = BackendSolver(o2, device=0)
cfg = synth.SynthConfig(n_features=230)
wins, priors = synth.make_batch(1000, 64, ts.options, cfg, distinct=64)
tdw = [synth.with_td_inputs(wins[i], 10 + i) for i in range(min(nb, 16))]
tdw = [tdw[i % len(tdw)] for i in range(nb)]
tdp = [priors[i % min(nb, 16)] for i in range(nb)]
ts.batch_upload(tdw, tdp)
for _ in range(2):
    ts.batch_rewind(); ts.batch_solve()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    ts.batch_rewind(); ts.batch_solve()
torch.cuda.synchronize(); print("estimate_td batch of %d: %.3f ms per solve, iterations %d" % (nb, (time.perf_counter() - t0) / 5 * 1e3, sum(x.num_iterations for x in ts.batch_summaries())))
ts.close()

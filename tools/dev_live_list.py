"""converging batch with / without the live-window lists (VILF_NO_LIVE_LIST=1), three placements of the finished windows; also the regular batch (nothing converges)"""
import os, sys, time, copy
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
from vil_fusion_amd.lib import default_options
opts = default_options()
B = 4096
wins, priors = synth.make_batch(1000, B, opts, synth.SynthConfig(n_features=230), distinct=32)
half = 16
o2 = type(opts).from_buffer_copy(opts); o2.max_num_iterations = 1000
pre = BackendSolver(o2); pre.batch_upload(wins[:half], priors[:half]); pre.batch_solve(sync=True); pres = pre.batch_download(); pre.close()
cw = []
for i in range(32):
    if i < half:
        w2 = copy.deepcopy(wins[i]); r_ = pres[i]
        w2.para_pose = np.ascontiguousarray(np.asarray(r_.para_pose).reshape(-1, 7)); w2.para_speed_bias = np.ascontiguousarray(np.asarray(r_.para_speed_bias).reshape(-1, 9)); w2.para_feature = np.ascontiguousarray(r_.para_feature)
        cw.append(w2)
    else:
        cw.append(wins[i])
def run(order, tag):
    s = BackendSolver()
    s.batch_upload([cw[k] for k in order], [priors[k] for k in order])
    s.batch_rewind(); s.batch_solve(sync=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        s.batch_rewind(); s.batch_solve(sync=True)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    its = sum(x.num_iterations for x in s.batch_summaries())
    s.close()
    print(f"{tag:34s} {1e3 * dt:7.2f} ms per solve of {B} windows, mean iterations {its / B:.2f}", flush=True)
rng = np.random.default_rng(77)
print("VILF_NO_LIVE_LIST =", os.environ.get("VILF_NO_LIVE_LIST"))
run([k % 32 for k in range(B)], "finished / regular in groups of 16")
run([int(k) for k in rng.integers(0, 32, B)], "random placement")
run([(k * 32) // B for k in range(B)], "contiguous halves")
run([16 + k % 16 for k in range(B)], "regular batch (nothing converges)")

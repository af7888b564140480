"""Dev experiment: does a heterogeneous mix of the two window kernels on a CU beat the homogeneous one? One handle with B windows vs
two handles with B / 2 each, solving concurrently on their own HIP streams (host thread each): k_linearize of one half then runs next
to k_solve_sb of the other whenever the two chains drift apart."""
import sys, time, threading
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
import torch
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
stagger = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
NH = int(sys.argv[3]) if len(sys.argv) > 3 else 2
one = BackendSolver(device=0)
opts = one.options
cfg = synth.SynthConfig(n_features=230)
wins, priors = synth.make_batch(1000, B, opts, cfg, distinct=64)
one.batch_upload(wins, priors)
halves = [BackendSolver(device=0) for _ in range(NH)]
for k, h in enumerate(halves):
    h.batch_upload(wins[k * B // NH:(k + 1) * B // NH], priors[k * B // NH:(k + 1) * B // NH])
def t_one():
    one.batch_rewind(); one.batch_solve(sync=True)
def t_two():
    def run(h, d):
        if d: time.sleep(d)
        h.batch_rewind(); h.batch_solve(sync=True)
    th = [threading.Thread(target=run, args=(h, stagger * k)) for k, h in enumerate(halves)]
    for t in th: t.start()
    for t in th: t.join()
def t_seq():
    for h in halves:
        h.batch_rewind(); h.batch_solve(sync=True)
def med(f, n=9):
    f(); ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); f(); ts.append(time.perf_counter() - t)
    return 1e3 * float(np.median(ts))
print("one handle, %d windows: %.3f ms" % (B, med(t_one)))
print("%d handles, %d windows each, concurrent (stagger %.1f ms): %.3f ms" % (NH, B // NH, 1e3 * stagger, med(t_two)))
print("the handles, one after the other: %.3f ms" % med(t_seq))
print("one handle again: %.3f ms" % med(t_one))

"""Dev: a stream of 1024-window batches alternating over two handles (upload of one while the other's solve is on the device): where does the host wait?"""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np, torch
from vil_fusion_amd import synth, abi as vabi
from vil_fusion_amd.estimator import BackendSolver
half = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NH = int(os.environ.get('NH', '2'))
hs = [BackendSolver(device=0) for _ in range(NH)]
cfg = synth.SynthConfig(n_features=230)
wins, priors = synth.make_batch(1000, NH * half, hs[0].options, cfg, distinct=64)
arrs, outs = [], []
for k in range(NH):
    hs[k].batch_upload(wins[k * half:(k + 1) * half], priors[k * half:(k + 1) * half])
    a = (vabi.WindowIn * half)()
    for i in range(half): a[i] = wins[k * half + i].as_struct()
    arrs.append(a); hs[k].batch_solve(sync=True); outs.append(hs[k].batch_download_states())
if os.environ.get('ASYNC_UPLOAD'):
    for h in hs: h.set_async_upload(True)
torch.cuda.synchronize()
rows = []
infl = [False] * NH
t00 = time.perf_counter()
for r in range(6):
    for k in range(NH):
        t0 = time.perf_counter()
        if infl[k]: hs[k].batch_download_states(out=outs[k])
        t1 = time.perf_counter()
        hs[k]._check(hs[k]._L.vilf_batch_upload(hs[k]._h, half, arrs[k]), "up")
        t2 = time.perf_counter()
        hs[k].batch_solve(sync=False); infl[k] = True
        t3 = time.perf_counter()
        rows.append((r, k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
for k in range(NH): hs[k].batch_download_states(out=outs[k])
t11 = time.perf_counter()
for x in rows: print("round %d handle %d: download %.2f ms, upload %.2f, solve call %.2f" % x)
print("total %.2f ms for %d batches of %d windows: %.2f ms per batch" % (1e3 * (t11 - t00), 6 * NH, half, 1e3 * (t11 - t00) / (6 * NH)))

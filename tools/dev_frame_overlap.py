"""Dev measurement: do two handles of one process run side by side? N frames of (window solve + marginalization) on one handle / thread and N scan-to-map steps of one
stream on another handle / thread: each alone, then both at once."""
import sys, os, time, threading
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
if os.environ.get("WITH_TORCH"):
    import torch; torch.cuda.set_device(0); _t = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, Scan2MapBatch
N = 40
extra = []
if os.environ.get('EXTRA_HANDLES'):
    import torch
    extra.append(BackendSolver(device=0, stream=torch.cuda.current_stream().cuda_stream))
    for _ in range(int(os.environ['EXTRA_HANDLES'])): extra.append(BackendSolver())
ls = BackendSolver(); o = ls.options
win, prior, _ = synth.make_window(1000, o, synth.SynthConfig(n_features=230))
ls.set_prior(prior); ls.optimization(win); ls.marginalize()
me, ms, scans, pl = synth.make_lidar_bench_case(7000)
lh = BackendSolver()
sb = Scan2MapBatch(lh, 1, len(scans[0][0]) + len(scans[1][0]) + 64, len(scans[0][1]) + len(scans[1][1]) + 64, len(me) + len(scans[0][0]) + 64, len(ms) + len(scans[0][1]) + 64)
sb.localMapInited(0, me, ms, None, pl); sb.set_scan(0, *scans[0]); sb.step(); sb.set_scan(0, *scans[1]); sb.snapshot()
def A():
    for _ in range(N): ls.optimization(win); ls.marginalize()
def Bf():
    for _ in range(N): sb.rewind(); sb.step(sync=True)
def wall(fs):
    th = [threading.Thread(target=f) for f in fs]
    t = time.perf_counter(); [x.start() for x in th]; [x.join() for x in th]; return 1e3 * (time.perf_counter() - t) / N
A(); Bf()
print("solve + marginalize alone: %.3f ms per frame" % wall([A]))
print("scan-to-map alone:         %.3f ms per frame" % wall([Bf]))
print("both at once:              %.3f ms per frame" % wall([A, Bf]))

"""Dev measurement: PCIe-inclusive rate of the batched window solve (host buffers in -> host buffers out)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
s = BackendSolver(); o = s.options
B = 2048
wins, priors = synth.make_batch(1000, B, o, synth.SynthConfig(n_features=230), distinct=64)
for rep in range(3):
    t0 = time.perf_counter(); s.batch_upload(wins, priors); t1 = time.perf_counter(); s.batch_solve(); t2 = time.perf_counter(); res = s.batch_download(); t3 = time.perf_counter()
    its = sum(r.summary["num_iterations"] for r in res)
    print(f"rep {rep}: upload(pack+H2D) {1e3*(t1-t0):.1f} ms  solve {1e3*(t2-t1):.1f} ms  download {1e3*(t3-t2):.1f} ms  -> {its/(t3-t0):.0f} iterations/s PCIe-inclusive, {its/(t2-t1):.0f} resident")

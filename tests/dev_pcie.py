"""Dev measurement: PCIe-inclusive rate of the batched window solve (host buffers in -> host buffers out)."""
import sys, time
import os; _R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R)
import numpy as np
from vil_fusion_amd import abi, synth
from vil_fusion_amd.estimator import BackendSolver
s = BackendSolver(); o = s.options
B = 2048
wins, priors = synth.make_batch(1000, B, o, synth.SynthConfig(n_features=230), distinct=64)
s.batch_upload(wins, priors)           # priors are uploaded once: in the running system they are produced on the device
for rep in range(3):
    t0 = time.perf_counter()
    arr = (abi.WindowIn * B)()
    for i, w in enumerate(wins):
        arr[i] = w.as_struct()
    t1 = time.perf_counter()
    s._check(s._L.vilf_batch_upload(s._h, B, arr), "vilf_batch_upload"); s._keep = (wins, arr)
    t2 = time.perf_counter(); s.batch_solve(); t3 = time.perf_counter(); res = s.batch_download(); t4 = time.perf_counter()
    its = sum(r.summary["num_iterations"] for r in res)
    print(f"rep {rep}: python structs {1e3*(t1-t0):.1f} ms | vilf_batch_upload (pack + H2D) {1e3*(t2-t1):.1f} ms | solve {1e3*(t3-t2):.1f} ms | download {1e3*(t4-t3):.1f} ms"
          f" -> {its/(t4-t1):.0f} iterations/s through the C ABI (host buffers in, host buffers out), {its/(t3-t2):.0f} resident")

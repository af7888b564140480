"""PCIe-inclusive rate of the batched window solve through the C ABI: host buffers in (vilf_batch_upload: pack + H2D) -> solve -> host buffers out
(vilf_batch_download_states). One handle, then two handles on two host threads (upload of one half overlaps the solve of the other)."""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from vil_fusion_amd import abi, synth
from vil_fusion_amd.estimator import BackendSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
s0 = BackendSolver(); o = s0.options
wins, priors = synth.make_batch(1000, B, o, synth.SynthConfig(n_features=230), distinct=64)


def prepared(solver, ws, ps):
    solver.batch_upload(ws, ps)            # priors are uploaded once: in the running system they are produced on the device
    arr = (abi.WindowIn * len(ws))()
    for i, w in enumerate(ws):
        arr[i] = w.as_struct()
    return arr, solver.batch_download_states()


def frame(solver, arr, out, t):
    t0 = time.perf_counter()
    solver._check(solver._L.vilf_batch_upload(solver._h, len(arr), arr), "vilf_batch_upload")
    t1 = time.perf_counter(); solver.batch_solve(); t2 = time.perf_counter()
    solver.batch_download_states(out=out); t3 = time.perf_counter()
    t[:] = [t1 - t0, t2 - t1, t3 - t2]


arr, out = prepared(s0, wins, priors)
ref = s0.batch_download()
s0.batch_solve()
chk = s0.batch_download_states()
assert np.array_equal(chk["Ps"][5], s0.batch_download()[5].Ps) and chk["summaries"][5].num_iterations == 8
for rep in range(4):
    t = [0, 0, 0]; ta = time.perf_counter(); frame(s0, arr, out, t); tb = time.perf_counter()
    its = sum(x.num_iterations for x in out["summaries"])
    print(f"one handle   B={B}: upload (pack + H2D) {1e3*t[0]:.1f} ms | solve {1e3*t[1]:.1f} ms | download {1e3*t[2]:.1f} ms -> {its/(tb-ta):.0f} iterations/s host-in/host-out, {its/t[1]:.0f} resident")
# two handles, half the windows each, one host thread per handle
h = B // 2
s1 = BackendSolver()
a0, o0 = prepared(s0, wins[:h], priors[:h]); a1, o1 = prepared(s1, wins[h:], priors[h:])
for rep in range(4):
    tt0, tt1 = [0, 0, 0], [0, 0, 0]
    ta = time.perf_counter()
    th = threading.Thread(target=frame, args=(s1, a1, o1, tt1)); th.start()
    frame(s0, a0, o0, tt0); th.join()
    tb = time.perf_counter()
    its = sum(x.num_iterations for x in o0["summaries"]) + sum(x.num_iterations for x in o1["summaries"])
    print(f"two handles  B={B}: {1e3*(tb-ta):.1f} ms per frame batch (upload {1e3*tt0[0]:.1f}/{1e3*tt1[0]:.1f}, solve {1e3*tt0[1]:.1f}/{1e3*tt1[1]:.1f}, download {1e3*tt0[2]:.1f}/{1e3*tt1[2]:.1f}) -> {its/(tb-ta):.0f} iterations/s host-in/host-out")

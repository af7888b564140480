# phase stamps of the Cholesky panel workgroup (diagnostic build of vilf_lw.hip)
set -e
touch vil_fusion_amd/csrc/vilf_lw.hip
make -s -C vil_fusion_amd/csrc DEFS=-DVILF_LW_STAMPS
python - <<P
import ctypes as C, numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
from vil_fusion_amd.lib import default_options, lib
opts = default_options(); opts.window_size = 50
win = synth.make_window(900, opts, synth.SynthConfig(n_frames=51, n_features=2500, with_prior=False))[0]
s = BackendSolver(opts)
for _ in range(3): s.optimization(win)
L = lib()
buf = (C.c_longlong * 64)()
L.vilf_debug_lw_stamps.argtypes = [C.POINTER(C.c_longlong)]
print("rc", L.vilf_debug_lw_stamps(buf))
st = np.array(buf[:20], dtype=np.int64)
names = ["start", "T loaded", "PRE done"] + [f"tc{t} {x}" for t in range(4) for x in ("A done", "barrier1", "B done", "barrier2")]
for i in range(1, 19):
    print("%-14s +%6.2f us  (at %7.2f)" % (names[i], (st[i] - st[i - 1]) / 100.0, (st[i] - st[0]) / 100.0))
P
touch vil_fusion_amd/csrc/vilf_lw.hip

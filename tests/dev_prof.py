import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s = BackendSolver()
wins, priors = synth.make_batch(7, B, s.options, distinct=16)
s.batch_upload(wins, priors)
s.batch_solve()
for rep in range(3):
    s.batch_rewind(); t = time.time(); s.batch_solve(); dt = time.time() - t
    its = sum(x.num_iterations for x in s.batch_summaries())
    print("B", B, "solve ms", dt * 1e3, "iter/s", its / dt)
import os, ctypes as C
if os.environ.get("VILF_DEBUG_STAMPS"):
    buf = (C.c_longlong * 96)()
    s._L.vilf_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    s._L.vilf_debug_stamps(s._h, buf)
    a = np.array(buf[:]).reshape(3, 32)
    for k, name in enumerate(["linearize", "solve", "step"]):
        v = a[k]; v = v[v != 0]
        if len(v) > 1:
            print(name, "phase cycles:", np.diff(v), "total", v[-1] - v[0])
if os.environ.get("VILF_DEBUG_STAMPS") and not os.environ.get("VILF_SOLVE_DENSE"):
    v = a[1]
    names = ["prologue", "setup", "P1 gather(w0)", "P2 reduce||chain", "P2b acc store+sums", "P3 Y chain+syrk", "P4 store", "P5 dense chol", "P6 dense backsub", "P6 chain (wave0)", "P6 wait dots+x", "P7"]
    print("k_solve_sb stamps (cycles, delta):", [(names[i], int(v[i] - v[i - 1])) for i in range(1, 12) if v[i] and v[i - 1]])
    print("  P1+P2 per-wave finish rel. to loop start:", [int(v[16 + q] - v[1]) for q in range(4)])
if os.environ.get("VILF_DEBUG_STAMPS"):
    print("linearize chunk-loop (wave 0): eval", a[2][16], "sync1", a[2][17], "mfma", a[2][18], "sync2", a[2][19])
if os.environ.get("VILF_DEBUG_STAMPS"):
    v = a[1]
    print("schur sub-stamps rel to stamp4: setup->", int(v[16] - v[4]), "mfma loop end->", int(v[15] - v[4]), "end->", int(v[5] - v[4]))

if os.environ.get("VILF_DEBUG_STAMPS"):
    print("cholesky (wave 0): potrf0", a[2][20], "trsm", a[2][21], "phase1", a[2][22], "phase2(potrf||rest)", a[2][23])

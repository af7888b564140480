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
        v = a[k][:10] if k == 0 else a[k]; v = v[v != 0]
        if len(v) > 1:
            print(name, "phase cycles:", np.diff(v), "total", v[-1] - v[0])
if os.environ.get("VILF_DEBUG_STAMPS"):
    v = a[0]
    if v[10] and v[13]:
        print("linearize [3->4] detail: IMU JtJ", int(v[10] - v[3]), "lidH/lidg", int(v[11] - v[10]), "prior cost + gradient", int(v[13] - v[11]), "barrier", int(v[4] - v[13]))
if os.environ.get("VILF_DEBUG_STAMPS") and not os.environ.get("VILF_SOLVE_DENSE"):
    v = a[1]
    names = ["prologue", "setup", "P1 gather(w0)", "P2 reduce||chain", "P2b acc store+sums", "P3 Y chain+syrk", "-", "P4+P5 dense chol (MFMA)", "P6 dense backsub", "P6 chain (wave0)", "P6 wait dots+x", "P7"]
    print("k_solve_sb stamps (cycles, delta):", [(names[i], int(v[i] - v[max(j for j in range(i) if v[j])])) for i in range(1, 12) if v[i]])
    print("  P1 detail (w0): decode+issue A", int(v[20]-v[1]), "process A", int(v[21]-v[20]), "decode+issue B", int(v[22]-v[21]), "process B", int(v[23]-v[22]), "rhs+cf", int(v[24]-v[23]), "| P2b: turns", int(v[26]-v[3]), "M/BP/N", int(v[27]-v[26]), "sums+cross", int(v[4]-v[27]))
    print("  P1+P2 per-wave finish rel. to loop start:", [int(v[16 + q] - v[1]) for q in range(4)])
if os.environ.get("VILF_DEBUG_STAMPS"):
    print("linearize chunk-loop (wave 0): eval", a[2][16], "sync1", a[2][17], "mfma", a[2][18], "sync2", a[2][19])
if os.environ.get("VILF_DEBUG_STAMPS"):
    v = a[1]
    print("schur sub-stamps rel to stamp4: setup->", int(v[16] - v[4]), "mfma loop end->", int(v[15] - v[4]), "end->", int(v[5] - v[4]))

if os.environ.get("VILF_DEBUG_STAMPS"):
    print("cholesky (wave 0): potrf0", a[2][20], "trsm", a[2][21], "phase1", a[2][22], "phase2(potrf||rest)", a[2][23])

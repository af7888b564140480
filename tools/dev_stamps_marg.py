# phase stamps of k_marg_prepare (diagnostic build: tools/dev_stamps_marg.sh). One mid-grid workgroup, cycles between consecutive stamps.
import sys, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import bench
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ragged-windows", "0", "--no-lidar-stage", "--no-pcie"]
bench.main()
from vil_fusion_amd import lib
L = lib.lib()
buf = (C.c_longlong * 128)()
L.vilf_debug_stamps_marg.argtypes = [C.POINTER(C.c_longlong)]
L.vilf_debug_stamps_marg(buf)
a = np.array(buf[:]).reshape(4, 32)
v = a[0]
idx = [i for i in range(32) if v[i]]
print("k_marg_prepare stamps:", [(idx[k + 1], int(v[idx[k + 1]] - v[idx[k]])) for k in range(len(idx) - 1)], "total", int(v[idx[-1]] - v[idx[0]]))
print("pair-loop sums (thread 0): top, stage, barrier, mfma, barrier:", [int(x) for x in a[0][24:30]])
print("tred2 Householder loop sums (thread 0): phaseA, barrier waits, matvec, phaseB, rank2, tail:", [int(x) for x in a[1][24:30]])
v = a[2]
nm = ["1/h", "arrow rows x MFMA", "S, Y -> LDS", "Cholesky of S", "Z = L^-1 Y", "trace chains", "reduce + guard", "Z^T Z (MFMA)", "Arr, br out"]
st = [int(v[i]) for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9)]
if st[0]: print("k_marg_schur fast path:", {nm[i]: st[i + 1] - st[i] for i in range(9)}, "total", st[-1] - st[0])
v = a[3]
if v[0]: print("k_mf_chol_tiles:", {"fill + H0 out": int(v[1] - v[0]), "tiles from LDS": int(v[2] - v[1]), "19 panels": int(v[3] - v[2]), "guard + J0, r0 out": int(v[4] - v[3]), "block table": int(v[5] - v[4])}, "panel-loop sums (thread 0): rank-4 MFMAs, panel -> LDS, barrier, row factor, barrier:", [int(x) for x in v[24:30]])

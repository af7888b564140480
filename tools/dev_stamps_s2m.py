# phase stamps of the scan-to-map kernels (diagnostic build: tools/dev_stamps_s2m.sh). One mid-grid workgroup of each launch, cycles.
import sys, json, ctypes as C
sys.path.insert(0, '.')
import numpy as np
import bench
sys.argv = [sys.argv[0], "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--ragged-windows", "0", "--no-marginalize"]
bench.main()
from vil_fusion_amd import lib
L = lib.lib()
buf = (C.c_longlong * 288)()
L.vilf_debug_stamps_s2m.argtypes = [C.POINTER(C.c_longlong)]
L.vilf_debug_stamps_s2m(buf)
a = np.array(buf[:]).reshape(9, 32)
import os
SVP = ["bbox", "keys", "radix sort", "heads+centroids"] if os.environ.get("VILF_SV_NO_RUNS") else ["points -> runs", "keys", "radix sort", "leaf heads", "leaf sums"]     # aux: key bits, runs, leaves
names = {0: ("scan_voxel (surf cloud)", SVP), 1: ("scan_voxel (edge cloud)", SVP),
         2: ("bucket_index surf", ["zero", "count pass", "wait", "scan", "scatter pass", "wait"]), 3: ("bucket_index edge", ["zero", "count pass", "wait", "scan", "scatter pass", "wait"]),
         4: ("map_update surf", ["tail sort", "sweep", "queued", "beyond"]), 5: ("map_update edge", ["tail sort", "sweep", "queued", "beyond"])}
for k, (nm, ph) in names.items():
    v = a[k]
    st = [int(x) for x in v[:len(ph) + 1]]
    if st[0] == 0: continue
    print(nm, "n", int(v[30]), "aux", int(v[31]), int(v[29]), int(v[28]), "total", st[-1] - st[0], {p: st[i + 1] - st[i] for i, p in enumerate(ph)})
    if v[16:24].any(): print("   tile-loop sums (thread 0):", [int(x) for x in v[16:24]])
for k, nm in ((6, "associate edge"), (7, "associate surf")):
    print(nm, "queries", int(a[k][30]), "wave 0 of block 2: knn cycles", int(a[k][0]), "fit + store", int(a[k][1]))
v = a[8]
print("b_solve pass 0: total", int(v[0]), "list build", int(v[1]), "evaluate sweeps", int(v[2]), "step (wave 0) + 2 barriers", int(v[3]), "accept test + copy", int(v[4]), "iterations", int(v[5]), "factors", int(v[6]), "slots", int(v[7]))

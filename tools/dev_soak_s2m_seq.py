"""Dev soak: LiDAR sequences of varied size (rings x azimuths, frames) through the HIP scan-to-map path and the oracle, frame by frame: counts identical, poses to 1e-9."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, Scan2Map
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
o = ol.default_options()
bad = 0; nf = 0; worst = 0.0; flips = 0
t0 = time.time()
for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    rings = int(rng.choice([16, 32, 64])); az = int(rng.choice([450, 900, 1800])); n = int(rng.integers(4, 9)); seed = int(rng.integers(1, 10000))
    scans, poses = synth.make_lidar_sequence(seed, n, rings=rings, azimuths=az)
    ref = ol.OracleS2M(o); ref.init(*scans[0])
    s = BackendSolver(o); dev = Scan2Map(s); dev.localMapInited(*scans[0])
    for k in range(1, n):
        r = ref.step(*scans[k]); g = dev.optimation_processing(*scans[k]); nf += 1
        same = (r.n_edge_ds, r.n_surf_ds, list(r.n_edge_factors), list(r.n_surf_factors), list(r.iterations), r.map_edge_size, r.map_surf_size) == \
               (g.n_edge_ds, g.n_surf_ds, list(g.n_edge_factors), list(g.n_surf_factors), list(g.iterations), g.map_edge_size, g.map_surf_size)
        dp = float(np.abs(np.array(r.pose_qt[:]) - np.array(g.pose_qt[:])).max()); worst = max(worst, dp)
        if not same or dp > 1e-9:
            bad += 1; print("MISMATCH case", case, "rings", rings, "az", az, "frame", k, "dpose", dp, "counts equal", same)
    for which in (0, 1):
        a, b = ref.get_map(which), dev.getMapCloud(which)
        if a.shape != b.shape: bad += 1; print("MAP SHAPE case", case, which, a.shape, b.shape)
        else: flips += int(np.any(a != b, axis=1).sum())
    s.close()
print("frames", nf, "mismatches", bad, "worst |dpose|", worst, "map rows that differ in a bit (optimised poses agree to 1e-9, not to the bit)", flips, "seconds", round(time.time() - t0, 1))

import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import oracle_lib as ol
from vil_fusion_amd import synth
from vil_fusion_amd.estimator import BackendSolver, Scan2Map
o = ol.default_options()
scans, poses = synth.make_lidar_sequence(3, 8, rings=32, azimuths=900)
ref = ol.OracleS2M(o); ref.init(*scans[0])
s = BackendSolver(o); dev = Scan2Map(s); dev.localMapInited(*scans[0])
for k in range(1, 8):
    t = time.time(); r = ref.step(*scans[k]); tr = time.time() - t
    t = time.time(); g = dev.optimation_processing(*scans[k]); tg = time.time() - t
    print(k, "ds", (r.n_edge_ds, r.n_surf_ds), (g.n_edge_ds, g.n_surf_ds), "fac", list(r.n_edge_factors), list(g.n_edge_factors), list(r.n_surf_factors), list(g.n_surf_factors),
          "its", list(r.iterations), list(g.iterations), "map", (r.map_edge_size, r.map_surf_size), (g.map_edge_size, g.map_surf_size))
    print("   dpose", np.abs(np.array(r.pose_qt[:]) - np.array(g.pose_qt[:])).max(), "cost", list(r.final_cost), list(g.final_cost), "ms cpu/gpu", round(tr * 1e3, 2), round(tg * 1e3, 2))
me, mg = ref.get_map(1), dev.getMapCloud(1)
print("map surf equal:", me.shape == mg.shape and np.abs(me - mg).max())

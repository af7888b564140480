"""Dev soak of the scan voxel grid (b_scan_voxel_runs + the handed-back streams) against the oracle: many cloud sizes around the tile / strip / capacity
boundaries, scan-ordered and random clouds, several leaf sizes, four streams per batch. The map after a step without optimisation (tiny edge map) is the voxel grid
of map + scan: down-sampled counts identical, maps bit-identical."""
import sys, os, time
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, 'tests'))
import numpy as np
import oracle_lib as ol
from vil_fusion_amd.estimator import BackendSolver, Scan2MapBatch

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
def rings(n, n_ring, jitter, scale):
    n_az = max(1, n // n_ring)
    az = np.linspace(0, 2 * np.pi, n_az, endpoint=False)
    pts = []
    for r in range(n_ring):
        rad = scale * (4.0 + 1.3 * r + 0.5 * np.sin(3 * az + r))
        pts.append(np.stack([rad * np.cos(az), rad * np.sin(az), -1.5 + 0.02 * r * np.cos(az), np.full_like(az, r)], 1))
    c = np.concatenate(pts)[:n].astype(np.float32)
    c[:, :3] += rng.normal(0, jitter, (len(c), 3)).astype(np.float32)
    return c
cloud = lambda n, h: np.concatenate([rng.uniform(-h, h, (n, 3)), rng.uniform(0, 1, (n, 1))], 1).astype(np.float32)
sizes = [1, 2, 63, 64, 65, 511, 512, 513, 2047, 2048, 2049, 4095, 4096, 4097, 7167, 7168, 7169, 8192, 12000, 16384, 20000, 21999]
S = 4
bad = 0; nb = 0
t0 = time.time()
for leaf in (0.1, 0.4, 0.8, 2.0):
    o = ol.default_options(); o.s2m_crop_half = 60.0; o.surf_leaf_size = leaf
    s = BackendSolver(o)
    b = Scan2MapBatch(s, S, 64, 22064, 256, 160000)
    refs = [ol.OracleS2M(o) for _ in range(S)]
    ident = np.array([0, 0, 0, 1, 0, 0, 0.0])
    for i in range(S):
        me, ms = cloud(5, 1.0), cloud(200, 20.0)
        b.localMapInited(i, me, ms, ident, ident); refs[i].init(me, ms); refs[i].set_pose(ident, ident)
    prev = [None] * S
    for rnd in range(10):
        scans = []
        for i in range(S):
            n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 22000))
            kind = rng.integers(0, 4)
            if kind == 0: sc = cloud(n, 20.0)
            elif kind == 1: sc = rings(n, int(rng.choice([1, 4, 16, 64])), 0.001 * rng.integers(0, 20), rng.uniform(0.3, 3.0))
            elif kind == 2: sc = np.repeat(cloud(max(1, n // 7), 20.0), 7, axis=0)[:n]            # runs of exact duplicates
            else: sc = np.concatenate([rings(n // 2 + 1, 8, 0.002, 1.0), cloud(n - n // 2, 20.0)])[:max(n, 1)]
            scans.append(sc)
        e = np.zeros((0, 4), np.float32)               # the edge map stays below ten points: no optimisation, the pose stays the identity (an optimised pose agrees to 1e-9, not to the bit,
                                                       # and a transformed point may then round the other way: one coordinate of one point in 20 k, seen with a growing edge map)
        want = []
        for i in range(S):
            b.set_scan(i, e, scans[i]); want.append(refs[i].step(e, scans[i]))
        if os.environ.get('SOAK_VERBOSE'): print('leaf', leaf, 'round', rnd, 'n', [len(x) for x in scans], flush=True)
        b.step(); got = b.results(); nb += 1
        new_prev = []
        for i in range(S):
            cur_map = refs[i].get_map(1); new_prev.append(cur_map)
            ok = (got[i].n_surf_ds == want[i].n_surf_ds) and np.array_equal(b.getMapCloud(i, 1), refs[i].get_map(1))
            if not ok:
                bad += 1
                m, r = b.getMapCloud(i, 1), refs[i].get_map(1)
                print("MISMATCH leaf", leaf, "round", rnd, "stream", i, "n", len(scans[i]), "ds", got[i].n_surf_ds, want[i].n_surf_ds, "map", m.shape, r.shape)
                if m.shape == r.shape:
                    d = np.flatnonzero(np.any(m != r, axis=1)); print("   rows differing", len(d), "first", d[:5], m[d[:2]], r[d[:2]])
                    ms_, rs_ = m[np.lexsort(m.T[::-1])], r[np.lexsort(r.T[::-1])]
                    print("   same multiset of points:", np.array_equal(ms_, rs_))
                if os.environ.get('SOAK_DUMP') and not os.path.exists(os.path.join(_R, 'gpurun_out', 'soak_case.npz')): np.savez(os.path.join(_R, 'gpurun_out', 'soak_case.npz'), scan=scans[i], got=m, want=r, leaf=leaf, prev=prev[i] if prev[i] is not None else np.zeros((0, 4), np.float32))
        prev = new_prev
    s.close()
print("batches", nb, "mismatches", bad, "seconds", round(time.time() - t0, 1))

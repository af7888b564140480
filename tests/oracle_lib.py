"""Loader for the CPU oracle (TEST INFRASTRUCTURE). Only tests/, smoke() and bench.py's cpu_baseline leg use it."""
import ctypes as C
import os
import subprocess
import numpy as np
from vil_fusion_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_lib = None

dpp = C.POINTER(abi.c_double_p)


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-j8"])


def lib():
    global _lib
    if _lib is None:
        so = os.environ.get("VILO_SO") or os.path.join(ORACLE_DIR, "liboracle_vilf.so")      # VILO_SO: another build of the oracle (a sanitizer build, tools/dev_oracle_asan.sh)
        if not os.path.exists(so):
            build()
        _lib = C.CDLL(so)
        _lib.vilo_default_options.argtypes = [C.POINTER(abi.Options)]
        _lib.vilo_default_options.restype = None
        _lib.vilo_window_solve.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowIn), C.POINTER(abi.Prior), C.POINTER(abi.WindowOut)]
        _lib.vilo_window_marginalize.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.WindowIn), C.POINTER(abi.WindowOut), C.POINTER(abi.Prior), C.POINTER(abi.Prior)]
        _lib.vilo_last_trace.argtypes = [abi.c_double_p, C.c_int]
    return _lib


def default_options():
    o = abi.Options()
    lib().vilo_default_options(C.byref(o))
    return o


def window_solve(opts, win, prior=None):
    res = abi.WindowResult(win.n_frames, win.n_features)
    s = win.as_struct()
    rc = lib().vilo_window_solve(C.byref(opts), C.byref(s), C.byref(prior) if prior is not None else None, C.byref(res.struct))
    assert rc == 0, rc
    return res.finish()


def last_trace():
    buf = np.zeros((64, 9))
    n = lib().vilo_last_trace(abi.dptr(buf), 64)
    return buf[:n].copy()


def window_marginalize(opts, win, solved, prior=None):
    out = abi.Prior()
    s = win.as_struct()
    rc = lib().vilo_window_marginalize(C.byref(opts), C.byref(s), C.byref(solved.struct), C.byref(prior) if prior is not None else None, C.byref(out))
    assert rc == 0, rc
    return out


def _params(arrs):
    arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
    p = (abi.c_double_p * len(arrs))(*[abi.dptr(a) for a in arrs])
    return arrs, p


def eval_factor(kind, opts, params, *consts, sizes, nres, want_jac=True):
    """Call a Ceres-layout hook; returns (residuals, [jacobians row-major nres x size])."""
    L = lib()
    arrs, p = _params(params)
    r = np.zeros(nres)
    jacs = [np.zeros((nres, s)) for s in sizes]
    jp = (abi.c_double_p * len(sizes))(*[abi.dptr(j) for j in jacs]) if want_jac else None
    fn = getattr(L, "vilo_eval_" + kind)
    fn.restype = C.c_int
    args = []
    if kind != "prior":
        args.append(C.byref(opts))
    else:
        args.append(C.byref(consts[0])); consts = consts[1:]
    args.append(p)
    for c in consts:
        if isinstance(c, np.ndarray):
            args.append(abi.dptr(np.ascontiguousarray(c, dtype=np.float64)))
        elif isinstance(c, float):
            args.append(C.c_double(c))
        else:
            args.append(C.byref(c))
    args += [abi.dptr(r), jp]
    rc = fn(*args)
    assert rc == 0
    return r, jacs


class OracleS2M:
    """stateful CPU mirror of EstimationMapping (oracle/scan2map.cpp)"""

    def __init__(self, opts, _handle=None):
        L = lib()
        L.vilo_s2m_clone.restype = C.c_void_p
        L.vilo_s2m_clone.argtypes = [C.c_void_p]
        L.vilo_s2m_create.restype = C.c_void_p
        L.vilo_s2m_create.argtypes = [C.POINTER(abi.Options)]
        L.vilo_s2m_destroy.argtypes = [C.c_void_p]
        fp = C.POINTER(C.c_float)
        L.vilo_s2m_init.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int]
        L.vilo_s2m_step.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int, C.POINTER(abi.Scan2MapResult)]
        L.vilo_s2m_get_map.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, C.POINTER(C.c_int)]
        L.vilo_s2m_set_pose.argtypes = [C.c_void_p, abi.c_double_p, abi.c_double_p]
        self.L = L
        self.opts = opts
        self.h = _handle if _handle is not None else L.vilo_s2m_create(C.byref(opts))

    def clone(self):
        """independent copy of the maps and poses (replay from a prepared state)"""
        return OracleS2M(self.opts, _handle=self.L.vilo_s2m_clone(self.h))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vilo_s2m_destroy(self.h); self.h = None

    @staticmethod
    def _fp(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        return a, a.ctypes.data_as(C.POINTER(C.c_float))

    def init(self, edge, surf):
        e, ep = self._fp(edge); s, sp = self._fp(surf)
        assert self.L.vilo_s2m_init(self.h, ep, len(e), sp, len(s)) == 0

    def step(self, edge, surf):
        e, ep = self._fp(edge); s, sp = self._fp(surf)
        res = abi.Scan2MapResult()
        assert self.L.vilo_s2m_step(self.h, ep, len(e), sp, len(s), C.byref(res)) == 0
        return res

    def get_map(self, which):
        n = C.c_int(0)
        self.L.vilo_s2m_get_map(self.h, which, None, 0, C.byref(n))
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self.L.vilo_s2m_get_map(self.h, which, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n))
        return out[:n.value]

    def set_pose(self, pose, pose_last):
        self.L.vilo_s2m_set_pose(self.h, abi.dptr(np.ascontiguousarray(pose, dtype=np.float64)), abi.dptr(np.ascontiguousarray(pose_last, dtype=np.float64)))


def extract_features(xyzi, n_scans=64, min_range=3.0, max_range=100.0, edge_threshold=0.1):
    """featureExtraction::extractFeature on the oracle: (edge_xyzi, surf_xyzi)"""
    L = lib()
    fp = C.POINTER(C.c_float)
    L.vilo_extract_features.argtypes = [fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, fp, C.c_int, C.POINTER(C.c_int), fp, C.c_int, C.POINTER(C.c_int)]
    a = np.ascontiguousarray(xyzi, dtype=np.float32)
    n = len(a)
    e = np.zeros((max(n, 1), 4), dtype=np.float32); s = np.zeros((max(n, 1), 4), dtype=np.float32)
    ne, ns = C.c_int(0), C.c_int(0)
    rc = L.vilo_extract_features(a.ctypes.data_as(fp), n, n_scans, min_range, max_range, edge_threshold, e.ctypes.data_as(fp), n, C.byref(ne), s.ctypes.data_as(fp), n, C.byref(ns))
    assert rc == 0, rc
    return e[:ne.value].copy(), s[:ns.value].copy()


def feature_depth(cloud_xyzi, feat_xyz):
    """getFeatureDepth on the oracle: depth per feature (-1 = none)"""
    L = lib()
    fp = C.POINTER(C.c_float)
    L.vilo_feature_depth.argtypes = [fp, C.c_int, fp, C.c_int, fp]
    c = np.ascontiguousarray(cloud_xyzi, dtype=np.float32); f = np.ascontiguousarray(feat_xyz, dtype=np.float32)
    out = np.zeros(max(len(f), 1), dtype=np.float32)
    assert L.vilo_feature_depth(c.ctypes.data_as(fp), len(c), f.ctypes.data_as(fp), len(f), out.ctypes.data_as(fp)) == 0
    return out[:len(f)].copy()


def visual_imu_alignment(opts, noise, frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, n_samples, dt, acc, gyr, bgs0):
    """VisualIMUAlignment on the oracle: dict(ok, delta_bg, g, x, pre[(n-1, 467)])"""
    L = lib()
    dp = abi.c_double_p
    L.vilo_visual_imu_alignment.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.ImuNoise), C.c_int, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_int), C.c_int, dp, dp, dp,
                                            dp, dp, dp, dp, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, dt, acc, gyr, bgs0 = [f64(v) for v in (frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, dt, acc, gyr, bgs0)]
    ns = np.ascontiguousarray(n_samples, dtype=np.int32)
    n = len(frame_R)
    dbg, g, x = np.zeros(3), np.zeros(3), np.zeros(3 * n + 4)
    pre = np.zeros((max(n - 1, 1), abi.IMU_DOUBLES))
    nx, ok = C.c_int(0), C.c_int(0)
    rc = L.vilo_visual_imu_alignment(C.byref(opts), C.byref(noise), n, abi.dptr(frame_R), abi.dptr(frame_T), abi.dptr(acc_0), abi.dptr(gyr_0), abi.dptr(lin_ba), abi.dptr(lin_bg),
                                     ns.ctypes.data_as(C.POINTER(C.c_int)), dt.shape[1], abi.dptr(dt), abi.dptr(acc), abi.dptr(gyr), abi.dptr(bgs0),
                                     abi.dptr(dbg), abi.dptr(g), abi.dptr(x), C.byref(nx), pre.ctypes.data, C.byref(ok))
    assert rc == 0, rc
    return dict(ok=bool(ok.value), delta_bg=dbg, g=g, x=x[:nx.value].copy(), pre=pre[:n - 1])


def imu_preintegrate(noise, acc_0, gyr_0, ba, bg, dt, acc, gyr):
    """IntegrationBase on the oracle: 467 doubles in vilf_imu_preint order"""
    L = lib()
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    acc_0, gyr_0, ba, bg, dt, acc, gyr = [f64(v) for v in (acc_0, gyr_0, ba, bg, dt, acc, gyr)]
    out = abi.ImuPreint()
    L.vilo_imu_preintegrate(C.byref(noise), abi.dptr(acc_0), abi.dptr(gyr_0), abi.dptr(ba), abi.dptr(bg), len(dt), abi.dptr(dt), abi.dptr(acc), abi.dptr(gyr), C.byref(out))
    return np.frombuffer(bytes(out), dtype=np.float64).copy()


def pg_between(pi, pj, meas, sigma, robust=0):
    """one whitened BetweenFactor<Pose3>: (e[6], A[6,6], B[6,6], cost)"""
    L = lib()
    dp = abi.c_double_p
    L.vilo_pg_between.argtypes = [dp, dp, dp, dp, C.c_int, dp, dp, dp, dp]
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    pi, pj, meas, sigma = f64(pi), f64(pj), f64(meas), f64(sigma)
    e, A, B, c = np.zeros(6), np.zeros((6, 6)), np.zeros((6, 6)), np.zeros(1)
    assert L.vilo_pg_between(abi.dptr(pi), abi.dptr(pj), abi.dptr(meas), abi.dptr(sigma), robust, abi.dptr(e), abi.dptr(A), abi.dptr(B), abi.dptr(c)) == 0
    return e, A, B, c[0]


def pg_retract(p, delta):
    L = lib()
    dp = abi.c_double_p
    L.vilo_pg_retract.argtypes = [dp, dp, dp]
    out = np.zeros(7)
    L.vilo_pg_retract(abi.dptr(np.ascontiguousarray(p, dtype=np.float64)), abi.dptr(np.ascontiguousarray(delta, dtype=np.float64)), abi.dptr(out))
    return out


def make_pg_edges(edges):
    """edges: iterable of (i, j, q[4], t[3], sigma[6], robust) -> ctypes array of abi.PgEdge"""
    arr = (abi.PgEdge * max(len(edges), 1))()
    for k, (i, j, q, t, sg, rb) in enumerate(edges):
        arr[k].i, arr[k].j, arr[k].robust = int(i), int(j), int(rb)
        arr[k].q[:] = list(map(float, q)); arr[k].t[:] = list(map(float, t)); arr[k].sigma[:] = list(map(float, sg))
    return arr


def posegraph_optimize(poses_qt, prior_sigma, edges, max_iterations=30, tol=1e-10):
    """batch Gauss-Newton on the oracle: (poses[n,7], iterations, cost)"""
    L = lib()
    dp = abi.c_double_p
    L.vilo_posegraph_optimize.argtypes = [C.c_int, dp, dp, C.c_int, C.POINTER(abi.PgEdge), C.c_int, C.c_double, C.POINTER(C.c_int), dp]
    x = np.ascontiguousarray(poses_qt, dtype=np.float64).copy()
    ps = np.ascontiguousarray(prior_sigma, dtype=np.float64)
    arr = make_pg_edges(edges)
    it, cost = C.c_int(0), np.zeros(1)
    rc = L.vilo_posegraph_optimize(len(x), abi.dptr(x), abi.dptr(ps), len(edges), arr, max_iterations, tol, C.byref(it), abi.dptr(cost))
    assert rc == 0, rc
    return x, it.value, cost[0]


class SeqFrameOut(C.Structure):
    _fields_ = [("status", C.c_int), ("marginalization_flag", C.c_int), ("solver_flag", C.c_int), ("frame_count", C.c_int),
                ("n_features_window", C.c_int), ("last_track_num", C.c_int), ("stamp", C.c_double), ("P", C.c_double * 3), ("q_xyzw", C.c_double * 4),
                ("summary", abi.Summary)]


class OracleSequence:
    """the C++ restatement of the estimator's per-frame host loop (oracle/sequence.cpp): feature manager, processIMU / processImage, failureDetection
    reboot, slideWindow around the oracle's window solve + marginalization. Independent of vil_fusion_amd/sequence.py."""

    def __init__(self, opts, noise=None):
        from vil_fusion_amd import synth
        L = lib()
        L.vilo_seq_create.restype = C.c_void_p
        L.vilo_seq_create.argtypes = [C.POINTER(abi.Options), C.POINTER(abi.ImuNoise)]
        L.vilo_seq_destroy.argtypes = [C.c_void_p]
        L.vilo_seq_process_imu.argtypes = [C.c_void_p, C.c_double, abi.c_double_p, abi.c_double_p]
        L.vilo_seq_process_odometry.argtypes = [C.c_void_p, abi.c_double_p, abi.c_double_p]
        L.vilo_seq_process_image.argtypes = [C.c_void_p, C.c_double, C.c_int, C.POINTER(C.c_int), abi.c_double_p, abi.c_double_p, C.POINTER(SeqFrameOut)]
        ip = C.POINTER(C.c_int)
        L.vilo_seq_features.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip, ip, abi.c_double_p]
        L.vilo_seq_state.argtypes = [C.c_void_p] + [abi.c_double_p] * 5
        L.vilo_seq_prior.argtypes = [C.c_void_p, C.POINTER(abi.Prior)]
        self.L = L
        self.noise = noise if noise is not None else abi.ImuNoise(synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)
        self.h = L.vilo_seq_create(C.byref(opts), C.byref(self.noise))
        assert self.h
        self.solver_flag = 0
        self.trajectory, self.flags, self.summaries, self.events = [], [], [], []

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vilo_seq_destroy(self.h); self.h = None

    def process_imu(self, dt, acc, gyr):
        a = np.ascontiguousarray(acc, dtype=np.float64); w = np.ascontiguousarray(gyr, dtype=np.float64)
        self.L.vilo_seq_process_imu(self.h, float(dt), abi.dptr(a), abi.dptr(w))

    def process_odometry(self, q, t):
        q = np.ascontiguousarray(q, dtype=np.float64); t = np.ascontiguousarray(t, dtype=np.float64)
        self.L.vilo_seq_process_odometry(self.h, abi.dptr(q), abi.dptr(t))

    def process_image(self, image, stamp, init_state=None):
        ids = np.ascontiguousarray(list(image.keys()), dtype=np.int32)
        p8 = np.ascontiguousarray(np.array(list(image.values()), dtype=np.float64).reshape(-1, 8))
        st = None
        if init_state is not None:
            P, R, V, ba, bg = init_state
            st = np.ascontiguousarray(np.concatenate([np.ravel(P), np.ravel(R), np.ravel(V), np.ravel(ba), np.ravel(bg)]), dtype=np.float64)
        out = SeqFrameOut()
        rc = self.L.vilo_seq_process_image(self.h, float(stamp), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int)), abi.dptr(p8), abi.dptr(st) if st is not None else None, C.byref(out))
        assert rc == 0, rc
        self.solver_flag = out.solver_flag
        self.events.append(("fill", "solved", "reboot")[out.status])
        if out.status == 1:
            self.trajectory.append((out.stamp, np.array(out.P[:]), np.array(out.q_xyzw[:])))
            self.flags.append(out.marginalization_flag)
            self.summaries.append({k: getattr(out.summary, k) for k, _ in abi.Summary._fields_})
        return out

    def features(self):
        """[(feature_id, start_frame, n_obs, solve_flag, lidar_depth_flag, estimated_depth)] in list order"""
        cap = 4096
        i = [np.zeros(cap, dtype=np.int32) for _ in range(5)]
        d = np.zeros(cap)
        ip = C.POINTER(C.c_int)
        n = self.L.vilo_seq_features(self.h, cap, *[a.ctypes.data_as(ip) for a in i], abi.dptr(d))
        assert n <= cap
        return [(int(i[0][k]), int(i[1][k]), int(i[2][k]), int(i[3][k]), int(i[4][k]), float(d[k])) for k in range(n)]

    def state(self):
        Ps, Rs, Vs, Bas, Bgs = np.zeros((11, 3)), np.zeros((11, 3, 3)), np.zeros((11, 3)), np.zeros((11, 3)), np.zeros((11, 3))
        self.L.vilo_seq_state(self.h, *[abi.dptr(a) for a in (Ps, Rs, Vs, Bas, Bgs)])
        return Ps, Rs, Vs, Bas, Bgs

    def run(self, seq, n_frames=None, on_frame=None):
        """the driver loop of vil_fusion_amd.sequence.run_sequence (startup=None) over this object"""
        n = len(seq["images"]) if n_frames is None else n_frames
        self.process_imu(0.0, *seq["imu0"])
        for k in range(n):
            if k >= 1:
                dt, acc, gyr = seq["imu"][k]
                for a, w in zip(acc, gyr):
                    self.process_imu(dt, a, w)
                self.process_odometry(*seq["lidar"][k])
            out = self.process_image(seq["images"][k], seq["stamps"][k], seq["init"][k] if self.solver_flag == 0 else None)
            if on_frame is not None:
                on_frame(k, self, out)
        return self

"""Host-side mirror of the reference's operator surface for the hot path.

`BackendSolver.optimization(window)` ≙ Estimator::optimization() (vins_estimator/estimator.cpp:689-1050): solve on the
MI355X through the C ABI, gauge fix, then (optionally) marginalization; the prior lives on the device between calls like
`last_marginalization_info` (estimator.h:123-124). `BackendSolver.batch_*` drive many independent window snapshots.
Errors: the C ABI returns status ints; this wrapper raises VilfError for status < 0 (invalid argument / device error /
unsupported) and returns the summary for status >= 0, mirroring the reference where solver failures are not checked
(estimator.cpp:852-855).
"""
import ctypes as C
import numpy as np
from . import abi
from .lib import lib, VilfError, default_options


class BackendSolver:
    def __init__(self, options=None, device=0, stream=None):
        self._L = lib()
        self.options = options if options is not None else default_options()
        h = C.c_void_p()
        rc = self._L.vilf_create(C.byref(self.options), int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise VilfError(f"vilf_create failed with status {rc}" + (" (no GPU visible: the solve path has no CPU fallback)" if rc == abi.VILF_ERR_NO_GPU else ""))
        self._h = h
        self._keep = None
        self._n = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.vilf_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise VilfError(f"{what}: status {rc}: {self._L.vilf_last_error(self._h).decode()}")
        return rc

    # ---- single window (drop-in for Estimator::optimization) -----------------------------------------------------
    def set_prior(self, prior, slot=0):
        """≙ last_marginalization_info / last_marginalization_parameter_blocks"""
        if prior is None:
            prior = abi.Prior()
        self._check(self._L.vilf_prior_import(self._h, slot, C.byref(prior)), "vilf_prior_import")

    def get_prior(self, slot=0):
        p = abi.Prior()
        self._check(self._L.vilf_prior_export(self._h, slot, C.byref(p)), "vilf_prior_export")
        return p

    def optimization(self, window):
        res = abi.WindowResult(window.n_frames, window.n_features)
        s = window.as_struct()
        self._check(self._L.vilf_window_solve(self._h, C.byref(s), C.byref(res.struct)), "vilf_window_solve")
        return res.finish()

    def prepare_group(self, windows):
        """the ctypes views of a group of windows (input structs + result buffers), built once: a caller that solves the same group repeatedly (bench) or keeps
        its windows in C structs anyway does not pay the per-call conversion"""
        n = len(windows)
        res = [abi.WindowResult(w.n_frames, w.n_features) for w in windows]
        ins = (abi.WindowIn * n)(); outs = (abi.WindowOut * n)()
        for i, w in enumerate(windows):
            ins[i] = w.as_struct(); outs[i] = res[i].struct
        return dict(n=n, ins=ins, outs=outs, res=res, windows=windows)

    def solve_group(self, group, finish=True):
        """vilf_window_solve_group on a prepared group; finish=False skips the numpy unpacking of the results (the summaries are in group["outs"][i].summary)"""
        self._check(self._L.vilf_window_solve_group(self._h, group["n"], group["ins"], group["outs"]), "vilf_window_solve_group")
        if not finish:
            return group["outs"]
        for i, r in enumerate(group["res"]):
            r.struct = group["outs"][i]          # the summaries were written into the array's copies (the buffers they point to are the results' own)
        return [r.finish() for r in group["res"]]

    def optimization_group(self, windows):
        """several windows of sizes other than 11 frames (the general path) side by side in one chain of launches ≙ one optimization() each"""
        return self.solve_group(self.prepare_group(windows))

    def marginalize(self):
        self._check(self._L.vilf_window_marginalize(self._h), "vilf_window_marginalize")

    def reset(self):
        """≙ Estimator::clearState() for the prior (estimator.cpp:72-77)"""
        self._check(self._L.vilf_reset(self._h), "vilf_reset")

    # ---- batch of independent window snapshots -------------------------------------------------------------------
    def batch_upload(self, windows, priors=None):
        n = len(windows)
        arr = (abi.WindowIn * n)()
        for i, w in enumerate(windows):
            arr[i] = w.as_struct()
        if priors is not None:
            for i, p in enumerate(priors):
                self.set_prior(p, i)
        self._keep = (windows, arr)
        self._check(self._L.vilf_batch_upload(self._h, n, arr), "vilf_batch_upload")
        self._n = n
        self._shapes = [(w.n_frames, w.n_features) for w in windows]

    def set_async_upload(self, on=True):
        """vilf_set_async_upload: batch_upload returns once its copies are enqueued (streams of batches over two handles)"""
        self._check(self._L.vilf_set_async_upload(self._h, 1 if on else 0), "vilf_set_async_upload")

    def batch_solve(self, sync=True):
        self._check(self._L.vilf_batch_solve(self._h, 1 if sync else 0), "vilf_batch_solve")

    def batch_rewind(self):
        self._check(self._L.vilf_batch_rewind(self._h), "vilf_batch_rewind")

    def synchronize(self):
        self._check(self._L.vilf_synchronize(self._h), "vilf_synchronize")

    def wait_for(self, other):
        """work enqueued on this handle from now on starts after everything enqueued on `other` so far has finished (device-side dependency, no host wait)"""
        self._check(self._L.vilf_wait_for(self._h, other._h), "vilf_wait_for")

    def set_profiling(self, on=True):
        self._check(self._L.vilf_set_profiling(self._h, 1 if on else 0), "vilf_set_profiling")

    def get_profile(self):
        ms = (C.c_double * 4)(); n = (C.c_long * 4)()
        self._check(self._L.vilf_get_profile(self._h, ms, n), "vilf_get_profile")
        names = ["k_linearize", "k_solve", "k_step", "other"]
        return {names[i]: dict(ms=ms[i], launches=n[i]) for i in range(4)}

    S2M_GROUPS = ("s2m_voxel_grid", "s2m_radix_sort", "s2m_neighbour_index", "s2m_associate", "s2m_lm_solve", "s2m_submap", "s2m_other", "s2m_map_update")

    def get_profile_scan2map(self):
        ms = (C.c_double * 8)(); n = (C.c_long * 8)()
        self._check(self._L.vilf_get_profile_scan2map(self._h, ms, n), "vilf_get_profile_scan2map")
        return {k: dict(ms=ms[i], launches=n[i]) for i, k in enumerate(self.S2M_GROUPS)}

    def get_profile_marginalize(self):
        ms = (C.c_double * 4)(); n = (C.c_long * 4)()
        self._check(self._L.vilf_get_profile_marginalize(self._h, ms, n), "vilf_get_profile_marginalize")
        return {k: dict(ms=ms[i], launches=n[i]) for i, k in enumerate(("k_marg_prepare", "k_marg_schur", "k_marg_finish", "k_prior_prep"))}

    def get_profile_large_window(self):
        ms = (C.c_double * 4)(); n = (C.c_long * 4)()
        self._check(self._L.vilf_get_profile_large_window(self._h, ms, n), "vilf_get_profile_large_window")
        return {k: dict(ms=ms[i], launches=n[i]) for i, k in enumerate(("lw_factor_scatter", "lw_schur_syrk", "lw_cholesky", "lw_other"))}

    def marginalize_stats(self):
        """paths of the last marginalization: windows with a new prior, of those Amm by Cholesky, of those the kept block by Cholesky, windows left unchanged"""
        c = (C.c_int * 4)()
        self._check(self._L.vilf_batch_marginalize_stats(self._h, c), "vilf_batch_marginalize_stats")
        return dict(new_prior=c[0], amm_cholesky=c[1], kept_cholesky=c[2], unchanged=c[3])

    def batch_marginalize(self, sync=True):
        self._check(self._L.vilf_batch_marginalize(self._h, 1 if sync else 0), "vilf_batch_marginalize")

    def batch_summaries(self, first=0, n=None):
        n = self._n - first if n is None else n
        arr = (abi.Summary * n)()
        self._check(self._L.vilf_batch_summaries(self._h, first, n, arr), "vilf_batch_summaries")
        return arr

    def batch_download(self, first=0, n=None):
        n = self._n - first if n is None else n
        results = [abi.WindowResult(*self._shapes[first + i]) for i in range(n)]
        arr = (abi.WindowOut * n)()
        for i, r in enumerate(results):
            arr[i] = r.struct
        self._check(self._L.vilf_batch_download(self._h, first, n, arr), "vilf_batch_download")
        for i, r in enumerate(results):
            r.struct = arr[i]
            r.finish()
        return results

    def batch_download_states(self, first=0, n=None, out=None):
        """Ps / Rs / Vs / Bas / Bgs + summaries of n windows into contiguous numpy arrays (re-used when `out` is given): the per-frame download"""
        n = self._n - first if n is None else n
        if out is None:
            out = dict(Ps=np.empty((n, 11, 3)), Rs=np.empty((n, 11, 3, 3)), Vs=np.empty((n, 11, 3)), Bas=np.empty((n, 11, 3)), Bgs=np.empty((n, 11, 3)), summaries=(abi.Summary * n)())
        self._L.vilf_batch_download_states.argtypes = [C.c_void_p, C.c_int, C.c_int, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.POINTER(abi.Summary)]
        self._check(self._L.vilf_batch_download_states(self._h, first, n, abi.dptr(out["Ps"]), abi.dptr(out["Rs"]), abi.dptr(out["Vs"]), abi.dptr(out["Bas"]), abi.dptr(out["Bgs"]), out["summaries"]),
                    "vilf_batch_download_states")
        return out

    def newest_poses_to_device(self, stamps, device_ptr):
        st = np.ascontiguousarray(stamps, dtype=np.float64)
        self._check(self._L.vilf_batch_newest_poses_device(self._h, abi.dptr(st), C.c_void_p(device_ptr)), "vilf_batch_newest_poses_device")

    # ---- Ceres-layout hooks (same signatures as the reference's Evaluate) ----------------------------------------
    def _params(self, arrs):
        arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
        return arrs, (abi.c_double_p * len(arrs))(*[abi.dptr(a) for a in arrs])

    def eval_projection(self, params, pts_i, pts_j, want_ex_jacobian=True):
        arrs, p = self._params(params)
        r = np.zeros(2)
        jacs = [np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 1))]
        ptrs = [abi.dptr(jacs[0]), abi.dptr(jacs[1]), abi.dptr(jacs[2]) if want_ex_jacobian else abi.c_double_p(), abi.dptr(jacs[3])]
        jp = (abi.c_double_p * 4)(*ptrs)
        self._check(self._L.vilf_eval_projection(self._h, p, abi.dptr(np.ascontiguousarray(pts_i, dtype=np.float64)),
                                                 abi.dptr(np.ascontiguousarray(pts_j, dtype=np.float64)), abi.dptr(r), jp), "vilf_eval_projection")
        return r, jacs

    def eval_imu(self, params, pre):
        arrs, p = self._params(params)
        r = np.zeros(15)
        jacs = [np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))]
        jp = (abi.c_double_p * 4)(*[abi.dptr(j) for j in jacs])
        self._check(self._L.vilf_eval_imu(self._h, p, C.byref(pre), abi.dptr(r), jp), "vilf_eval_imu")
        return r, jacs

    def eval_imu_raw(self, params, pre):
        """residual / jacobians before the multiplication by sqrt_info, and the device's sqrt_info [15, 15]"""
        arrs, p = self._params(params)
        r = np.zeros(15); S = np.zeros((15, 15))
        jacs = [np.zeros((15, 7)), np.zeros((15, 9)), np.zeros((15, 7)), np.zeros((15, 9))]
        jp = (abi.c_double_p * 4)(*[abi.dptr(j) for j in jacs])
        self._check(self._L.vilf_eval_imu_raw(self._h, p, C.byref(pre), abi.dptr(r), jp, abi.dptr(S)), "vilf_eval_imu_raw")
        return r, jacs, S

    def eval_lidar_between(self, params, c):
        arrs, p = self._params(params)
        r = np.zeros(6)
        jacs = [np.zeros((6, 7)), np.zeros((6, 7))]
        jp = (abi.c_double_p * 2)(*[abi.dptr(j) for j in jacs])
        self._check(self._L.vilf_eval_lidar_between(self._h, p, C.byref(c), abi.dptr(r), jp), "vilf_eval_lidar_between")
        return r, jacs

    def eval_projection_td(self, params, pts_i, pts_j, vel_i, vel_j, td_i, td_j, row_i, row_j):
        """ProjectionTdFactor::Evaluate on the device: params = [Pose_i, Pose_j, Ex_Pose, [inv depth], [td]]"""
        arrs, p = self._params(params)
        r = np.zeros(2)
        jacs = [np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 7)), np.zeros((2, 1)), np.zeros((2, 1))]
        jp = (abi.c_double_p * 5)(*[abi.dptr(j) for j in jacs])
        f = lambda v: abi.dptr(np.ascontiguousarray(v, dtype=np.float64))
        self._L.vilf_eval_projection_td.argtypes = [C.c_void_p, C.c_void_p, abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p,
                                                    C.c_double, C.c_double, C.c_double, C.c_double, abi.c_double_p, C.c_void_p]
        self._check(self._L.vilf_eval_projection_td(self._h, p, f(pts_i), f(pts_j), f(vel_i), f(vel_j), float(td_i), float(td_j), float(row_i), float(row_j),
                                                    abi.dptr(r), jp), "vilf_eval_projection_td")
        return r, jacs

    def eval_prior(self, prior, params):
        """MarginalizationFactor::Evaluate (marginalization_factor.cpp:333-381) on the device: residuals [n], jacobians [n x size_i]"""
        arrs, p = self._params(params)
        sizes = [prior.block_size[i] for i in range(prior.n_blocks)]
        r = np.zeros(prior.n)
        jacs = [np.zeros((prior.n, s)) for s in sizes]
        jp = (abi.c_double_p * len(sizes))(*[abi.dptr(j) for j in jacs])
        self._check(self._L.vilf_eval_prior(self._h, C.byref(prior), p, abi.dptr(r), jp), "vilf_eval_prior")
        return r, jacs

    def eval_edge(self, pose, cp, a, b):
        r = np.zeros(3); J = np.zeros((3, 7))
        f = lambda x: abi.dptr(np.ascontiguousarray(x, dtype=np.float64))
        self._check(self._L.vilf_eval_edge(self._h, f(pose), f(cp), f(a), f(b), abi.dptr(r), abi.dptr(J)), "vilf_eval_edge")
        return r, J

    def eval_surf(self, pose, cp, n, d):
        r = np.zeros(1); J = np.zeros((1, 7))
        f = lambda x: abi.dptr(np.ascontiguousarray(x, dtype=np.float64))
        self._check(self._L.vilf_eval_surf(self._h, f(pose), f(cp), f(n), float(d), abi.dptr(r), abi.dptr(J)), "vilf_eval_surf")
        return r, J

    def pose_plus(self, x, d, se3=False):
        out = np.zeros(7)
        f = lambda v: abi.dptr(np.ascontiguousarray(v, dtype=np.float64))
        fn = self._L.vilf_se3_plus if se3 else self._L.vilf_pose_plus
        self._check(fn(self._h, f(x), f(d), abi.dptr(out)), "vilf_pose_plus")
        return out


def imu_preintegrate(noise, acc_0, gyr_0, ba, bg, dt, acc, gyr):
    """≙ IntegrationBase (host): returns an abi.ImuPreint."""
    out = abi.ImuPreint()
    f = lambda v: abi.dptr(np.ascontiguousarray(v, dtype=np.float64))
    dt = np.ascontiguousarray(dt, dtype=np.float64)
    rc = lib().vilf_imu_preintegrate(C.byref(noise), f(acc_0), f(gyr_0), f(ba), f(bg), len(dt), abi.dptr(dt), f(acc), f(gyr), C.byref(out))
    if rc != 0:
        raise VilfError(f"vilf_imu_preintegrate: status {rc}")
    return out


def imu_preintegrate_batch(solver, noise, acc_0, gyr_0, ba, bg, n_samples, dt, acc, gyr):
    """≙ IntegrationBase for many intervals at once on the device (vilf_imu_preintegrate_batch). acc_0 / gyr_0 / ba / bg: (n, 3);
    n_samples: (n,); dt: (n, max); acc / gyr: (n, max, 3). Returns an (n, 467) array of vilf_imu_preint rows."""
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    acc_0, gyr_0, ba, bg, dt, acc, gyr = [f64(v) for v in (acc_0, gyr_0, ba, bg, dt, acc, gyr)]
    ns = np.ascontiguousarray(n_samples, dtype=np.int32)
    n, mx = len(ns), dt.shape[1] if dt.ndim == 2 else 0
    out = np.zeros((max(n, 1), abi.IMU_DOUBLES))
    L = solver._L
    L.vilf_imu_preintegrate_batch.argtypes = [C.c_void_p, C.c_int, C.POINTER(abi.ImuNoise), abi.c_double_p, abi.c_double_p, abi.c_double_p, abi.c_double_p,
                                              C.POINTER(C.c_int), C.c_int, abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_void_p]
    solver._check(L.vilf_imu_preintegrate_batch(solver._h, n, C.byref(noise), abi.dptr(acc_0), abi.dptr(gyr_0), abi.dptr(ba), abi.dptr(bg),
                                                ns.ctypes.data_as(C.POINTER(C.c_int)), mx, abi.dptr(dt), abi.dptr(acc), abi.dptr(gyr), out.ctypes.data), "vilf_imu_preintegrate_batch")
    return out[:n]


def visual_imu_alignment(solver, noise, frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, n_samples, dt, acc, gyr, bgs0):
    """≙ VisualIMUAlignment(all_image_frame, Bgs, g, x) (initial/initial_aligment.cpp:199) on the device (vilf_visual_imu_alignment).
    frame_R (n, 3, 3) = c0_R_bk, frame_T (n, 3) = c0_T_ck up to scale, the n - 1 raw IMU intervals as in imu_preintegrate_batch.
    Returns dict(ok, delta_bg, g, x, pre): the reference's bool, the gyro-bias correction, refined gravity in c0, x = [body velocities,
    2 tangent coefficients, scale] and the intervals re-integrated at (0, bgs0 + delta_bg) as (n - 1, 467) vilf_imu_preint rows."""
    f64 = lambda v: np.ascontiguousarray(v, dtype=np.float64)
    frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, dt, acc, gyr, bgs0 = [f64(v) for v in (frame_R, frame_T, acc_0, gyr_0, lin_ba, lin_bg, dt, acc, gyr, bgs0)]
    ns = np.ascontiguousarray(n_samples, dtype=np.int32)
    n = len(frame_R)
    assert frame_R.size == 9 * n and frame_T.size == 3 * n and len(ns) == n - 1
    dbg, g, x = np.zeros(3), np.zeros(3), np.zeros(3 * n + 4)
    pre = np.zeros((max(n - 1, 1), abi.IMU_DOUBLES))
    nx, ok = C.c_int(0), C.c_int(0)
    L = solver._L
    dp = abi.c_double_p
    L.vilf_visual_imu_alignment.argtypes = [C.c_void_p, C.c_int, dp, dp, C.POINTER(abi.ImuNoise), dp, dp, dp, dp, C.POINTER(C.c_int), C.c_int, dp, dp, dp,
                                            dp, dp, dp, dp, C.POINTER(C.c_int), C.c_void_p, C.POINTER(C.c_int)]
    solver._check(L.vilf_visual_imu_alignment(solver._h, n, abi.dptr(frame_R), abi.dptr(frame_T), C.byref(noise), abi.dptr(acc_0), abi.dptr(gyr_0), abi.dptr(lin_ba),
                                              abi.dptr(lin_bg), ns.ctypes.data_as(C.POINTER(C.c_int)), dt.shape[1] if dt.ndim == 2 else 0, abi.dptr(dt), abi.dptr(acc),
                                              abi.dptr(gyr), abi.dptr(bgs0), abi.dptr(dbg), abi.dptr(g), abi.dptr(x), C.byref(nx), pre.ctypes.data, C.byref(ok)),
                  "vilf_visual_imu_alignment")
    return dict(ok=bool(ok.value), delta_bg=dbg, g=g, x=x[:nx.value].copy(), pre=pre[:n - 1])


def posegraph_optimize(solver, poses_qt, prior_sigma, edges, max_iterations=30, tol=1e-9):
    """≙ the gtsam graph + isam update of global_fusion (poseGraphOptimization.cpp:349-374, :560-587, :433-436) on the device
    (vilf_posegraph_optimize). poses_qt (n, 7) [qx qy qz qw tx ty tz]; edges: iterable of (i, j, q[4], t[3], sigma[6], robust).
    Returns (poses (n, 7), iterations, cost)."""
    x = np.ascontiguousarray(poses_qt, dtype=np.float64).copy()
    ps = np.ascontiguousarray(prior_sigma, dtype=np.float64)
    arr = (abi.PgEdge * max(len(edges), 1))()
    for k, (i, j, q, t, sg, rb) in enumerate(edges):
        arr[k].i, arr[k].j, arr[k].robust = int(i), int(j), int(rb)
        arr[k].q[:] = [float(v) for v in q]; arr[k].t[:] = [float(v) for v in t]; arr[k].sigma[:] = [float(v) for v in sg]
    it, cost = C.c_int(0), np.zeros(1)
    L = solver._L
    L.vilf_posegraph_optimize.argtypes = [C.c_void_p, C.c_int, abi.c_double_p, abi.c_double_p, C.c_int, C.POINTER(abi.PgEdge), C.c_int, C.c_double, C.POINTER(C.c_int), abi.c_double_p]
    solver._check(L.vilf_posegraph_optimize(solver._h, len(x), abi.dptr(x), abi.dptr(ps), len(edges), arr, max_iterations, tol, C.byref(it), abi.dptr(cost)), "vilf_posegraph_optimize")
    return x, it.value, cost[0]


class Scan2Map:
    """Host mirror of EstimationMapping (feature_tracker/include/EstimationMapping.hpp): localMapInited / optimation_processing /
    getMapCloud over the device path. One LiDAR stream per BackendSolver handle."""

    def __init__(self, solver):
        self.s = solver
        self._L = solver._L

    @staticmethod
    def _fp(a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        return a, a.ctypes.data_as(C.POINTER(C.c_float))

    def localMapInited(self, edge_xyzi, surf_xyzi):
        e, ep = self._fp(edge_xyzi); s, sp = self._fp(surf_xyzi)
        self.s._check(self._L.vilf_scan2map_init(self.s._h, ep, len(e), sp, len(s)), "vilf_scan2map_init")

    def optimation_processing(self, edge_xyzi, surf_xyzi):
        e, ep = self._fp(edge_xyzi); s, sp = self._fp(surf_xyzi)
        res = abi.Scan2MapResult()
        self.s._check(self._L.vilf_scan2map_step(self.s._h, ep, len(e), sp, len(s), C.byref(res)), "vilf_scan2map_step")
        return res

    def getMapCloud(self, which):
        n = C.c_int(0)
        self.s._check(self._L.vilf_scan2map_get_map(self.s._h, which, None, 0, C.byref(n)), "vilf_scan2map_get_map")
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self.s._check(self._L.vilf_scan2map_get_map(self.s._h, which, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)), "vilf_scan2map_get_map")
        return out[:n.value]

    def set_pose(self, pose_qt, pose_last_qt):
        f = lambda v: abi.dptr(np.ascontiguousarray(v, dtype=np.float64))
        self.s._check(self._L.vilf_scan2map_set_pose(self.s._h, f(pose_qt), f(pose_last_qt)), "vilf_scan2map_set_pose")


class Scan2MapBatch:
    """n_streams independent EstimationMapping objects (one per LiDAR stream / replayed segment) stepped together on the
    device: fixed per-stream capacities, device-side counters, no host round trip inside a step."""

    def __init__(self, solver, n_streams, cap_scan_edge, cap_scan_surf, cap_map_edge, cap_map_surf):
        self.s = solver
        self._L = solver._L
        self.n = int(n_streams)
        self.s._check(self._L.vilf_scan2map_batch_create(self.s._h, self.n, int(cap_scan_edge), int(cap_scan_surf), int(cap_map_edge), int(cap_map_surf)),
                      "vilf_scan2map_batch_create")

    def localMapInited(self, stream, edge_xyzi, surf_xyzi, pose_qt=None, pose_last_qt=None):
        e, ep = Scan2Map._fp(edge_xyzi); s, sp = Scan2Map._fp(surf_xyzi)
        keep = [None if v is None else np.ascontiguousarray(v, dtype=np.float64) for v in (pose_qt, pose_last_qt)]
        pp, pl = [None if v is None else abi.dptr(v) for v in keep]
        self.s._check(self._L.vilf_scan2map_batch_init(self.s._h, stream, ep, len(e), sp, len(s), pp, pl), "vilf_scan2map_batch_init")

    def set_scan(self, stream, edge_xyzi, surf_xyzi):
        e, ep = Scan2Map._fp(edge_xyzi); s, sp = Scan2Map._fp(surf_xyzi)
        self.s._check(self._L.vilf_scan2map_batch_set_scan(self.s._h, stream, ep, len(e), sp, len(s)), "vilf_scan2map_batch_set_scan")

    def step(self, sync=True):
        self.s._check(self._L.vilf_scan2map_batch_step(self.s._h, 1 if sync else 0), "vilf_scan2map_batch_step")

    def snapshot(self):
        self.s._check(self._L.vilf_scan2map_batch_snapshot(self.s._h), "vilf_scan2map_batch_snapshot")

    def rewind(self):
        self.s._check(self._L.vilf_scan2map_batch_rewind(self.s._h), "vilf_scan2map_batch_rewind")

    def copy_stream(self, src, dst):
        """stream dst := stream src (maps, poses, resident scan), on the device"""
        self.s._check(self._L.vilf_scan2map_batch_copy_stream(self.s._h, src, dst), "vilf_scan2map_batch_copy_stream")

    def results(self, first=0, n=None):
        n = self.n - first if n is None else n
        arr = (abi.Scan2MapResult * n)()
        self.s._check(self._L.vilf_scan2map_batch_results(self.s._h, first, n, arr), "vilf_scan2map_batch_results")
        return list(arr)

    def getMapCloud(self, stream, which):
        n = C.c_int(0)
        self.s._check(self._L.vilf_scan2map_batch_get_map(self.s._h, stream, which, None, 0, C.byref(n)), "vilf_scan2map_batch_get_map")
        out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
        self.s._check(self._L.vilf_scan2map_batch_get_map(self.s._h, stream, which, out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)), "vilf_scan2map_batch_get_map")
        return out[:n.value]


class FeatureExtraction:
    """Host mirror of featureExtraction (feature_tracker/include/featureExtraction.hpp): extractFeature(raw scan) -> (edge, surf)
    over the device path. Parameters ≙ initParam (:43-52; velodyne_param_64.yaml)."""

    def __init__(self, solver, n_scans=64, min_range=3.0, max_range=100.0, edge_threshold=0.1):
        self.s, self._L = solver, solver._L
        self.n_scans, self.min_range, self.max_range, self.edge_threshold = int(n_scans), float(min_range), float(max_range), float(edge_threshold)
        fp = C.POINTER(C.c_float)
        self._L.vilf_lidar_extract_features.argtypes = [C.c_void_p, fp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, fp, C.c_int, C.POINTER(C.c_int),
                                                        fp, C.c_int, C.POINTER(C.c_int)]

    def extractFeature(self, cloud_xyzi):
        a = np.ascontiguousarray(cloud_xyzi, dtype=np.float32)
        assert a.ndim == 2 and a.shape[1] == 4
        n = len(a)
        fp = C.POINTER(C.c_float)
        e = np.zeros((max(n, 1), 4), dtype=np.float32); s = np.zeros((max(n, 1), 4), dtype=np.float32)
        ne, ns = C.c_int(0), C.c_int(0)
        self.s._check(self._L.vilf_lidar_extract_features(self.s._h, a.ctypes.data_as(fp), n, self.n_scans, self.min_range, self.max_range, self.edge_threshold,
                                                          e.ctypes.data_as(fp), n, C.byref(ne), s.ctypes.data_as(fp), n, C.byref(ns)), "vilf_lidar_extract_features")
        return e[:ne.value].copy(), s[:ns.value].copy()

    def getFeatureDepth(self, depth_cloud_xyzi, features_xyz):
        """≙ getFeatureDepth (feature_tracker_node.cpp:54-163): LiDAR depth per visual feature, -1 = none"""
        c = np.ascontiguousarray(depth_cloud_xyzi, dtype=np.float32); f = np.ascontiguousarray(features_xyz, dtype=np.float32)
        assert c.ndim == 2 and c.shape[1] == 4 and f.ndim == 2 and f.shape[1] == 3
        fp = C.POINTER(C.c_float)
        self._L.vilf_feature_depth.argtypes = [C.c_void_p, fp, C.c_int, fp, C.c_int, fp]
        out = np.zeros(max(len(f), 1), dtype=np.float32)
        self.s._check(self._L.vilf_feature_depth(self.s._h, c.ctypes.data_as(fp), len(c), f.ctypes.data_as(fp), len(f), out.ctypes.data_as(fp)), "vilf_feature_depth")
        return out[:len(f)].copy()

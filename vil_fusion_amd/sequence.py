"""sequence.py — host-side sliding-window loop around the back-end.

A Python mirror of the reference's per-frame host logic, so whole multi-frame sequences can be replayed through either back-end
(the HIP path or the CPU oracle) and their trajectories compared:

    FeatureManager            ≙ vins_estimator/feature_manager.cpp (addFeatureCheckParallax :45-109, getFeatureCount :28-43,
                                setDepth :150-168, removeFailures :170-180, getDepthVector :194-216, triangulate :218-276,
                                removeBackShiftDepth :292-349, removeBack :351-366, removeFront :368-388, compensatedParallax2 :390-423)
    SlidingWindowEstimator    ≙ vins_estimator/estimator.cpp: clearState :36-88, processOdometry :90-101, processIMU :103-137, processImage :139-234
                                (NON_LINEAR branch incl. the failureDetection reboot :212-220; the SfM initialisation :237-459 is out of scope —
                                the window is bootstrapped from given states), solveOdometry :492-503, vector2double :505-547,
                                failureDetection :640-686, slideWindow :1052-1186
    run_sequences_lockstep    several independent estimators (sequence segments) stepped frame by frame as ONE batch on the device
                                (vilf_batch_upload / _solve / _marginalize per frame; every segment's prior stays in its slot)
    write_tum                 ≙ utility/visualization.cpp:159-172 (t x y z qx qy qz qw, time relative to the first frame)

The back-end is pluggable: `backend.solve(window) -> WindowResult` and `backend.marginalize(window, result)`; see HipBackend /
adapters in tests. Everything here is host bookkeeping — no arithmetic of the hot path lives in this file.
"""
import numpy as np

from . import abi, synth

WINDOW_SIZE = 10
MIN_PARALLAX = 10.0 / 460.0          # kitti_config.yaml keyframe_parallax / FOCAL_LENGTH (parameters.cpp:119)
INIT_DEPTH = 5.0
MARGIN_OLD, MARGIN_SECOND_NEW = abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW


class FeaturePerFrame:
    __slots__ = ("point", "uv", "velocity", "depth", "cur_td")

    def __init__(self, p8, td):
        self.point = np.array(p8[0:3], dtype=np.float64)
        self.uv = np.array(p8[3:5], dtype=np.float64)
        self.velocity = np.array(p8[5:7], dtype=np.float64)
        self.depth = float(p8[7])
        self.cur_td = td


class FeaturePerId:
    def __init__(self, feature_id, start_frame, measured_depth):
        self.feature_id, self.start_frame = feature_id, start_frame
        self.feature_per_frame = []
        self.used_num, self.solve_flag = 0, 0
        self.estimated_depth, self.lidar_depth_flag = (measured_depth, True) if measured_depth > 0 else (-1.0, False)

    def end_frame(self):
        return self.start_frame + len(self.feature_per_frame) - 1


class FeatureManager:
    def __init__(self):
        self.feature = []            # insertion order, like the reference's std::list
        self.last_track_num = 0

    def _used(self, it):
        it.used_num = len(it.feature_per_frame)
        return it.used_num >= 2 and it.start_frame < WINDOW_SIZE - 2

    def get_feature_count(self):
        return sum(1 for it in self.feature if self._used(it))

    def add_feature_check_parallax(self, frame_count, image, td):
        parallax_sum, parallax_num = 0.0, 0
        self.last_track_num = 0
        by_id = {it.feature_id: it for it in self.feature}
        for feature_id, p8 in image.items():          # std::map: ascending id
            f = FeaturePerFrame(p8, td)
            it = by_id.get(feature_id)
            if it is None:
                it = FeaturePerId(feature_id, frame_count, f.depth)
                it.feature_per_frame.append(f)
                self.feature.append(it); by_id[feature_id] = it
            else:
                it.feature_per_frame.append(f)
                self.last_track_num += 1
                if f.depth > 0 and not it.lidar_depth_flag:
                    it.estimated_depth = f.depth
                    it.lidar_depth_flag = True
                    it.feature_per_frame[0].depth = f.depth
        if frame_count < 2 or self.last_track_num < 20:
            return True
        for it in self.feature:
            if it.start_frame <= frame_count - 2 and it.start_frame + len(it.feature_per_frame) - 1 >= frame_count - 1:
                parallax_sum += self.compensated_parallax2(it, frame_count)
                parallax_num += 1
        if parallax_num == 0:
            return True
        return parallax_sum / parallax_num >= MIN_PARALLAX

    @staticmethod
    def compensated_parallax2(it, frame_count):
        pi = it.feature_per_frame[frame_count - 2 - it.start_frame].point
        pj = it.feature_per_frame[frame_count - 1 - it.start_frame].point
        du, dv = pi[0] / pi[2] - pj[0], pi[1] / pi[2] - pj[1]
        return max(0.0, float(np.sqrt(du * du + dv * dv)))

    def set_depth(self, x):
        k = -1
        for it in self.feature:
            if not self._used(it):
                continue
            k += 1
            it.estimated_depth = 1.0 / x[k]
            it.solve_flag = 2 if it.estimated_depth < 0 else 1

    def remove_failures(self):
        self.feature = [it for it in self.feature if it.solve_flag != 2]

    def get_depth_vector(self):
        return np.array([1.0 / it.estimated_depth if it.estimated_depth > 0 else 1.0 / INIT_DEPTH for it in self.feature if self._used(it)])

    def triangulate(self, Ps, Rs, tic, ric):
        for it in self.feature:
            if not self._used(it) or it.estimated_depth > 0:
                continue
            i = it.start_frame
            t0, R0 = Ps[i] + Rs[i] @ tic, Rs[i] @ ric
            rows = []
            for k, fpf in enumerate(it.feature_per_frame):
                j = i + k
                t1, R1 = Ps[j] + Rs[j] @ tic, Rs[j] @ ric
                t, R = R0.T @ (t1 - t0), R0.T @ R1
                P = np.hstack([R.T, (-R.T @ t)[:, None]])
                f = fpf.point / np.linalg.norm(fpf.point)
                rows.append(f[0] * P[2] - f[2] * P[0]); rows.append(f[1] * P[2] - f[2] * P[1])
            v = np.linalg.svd(np.array(rows))[2][-1]
            it.estimated_depth = v[2] / v[3]
            if it.estimated_depth < 0.1:
                it.estimated_depth = INIT_DEPTH

    def remove_back_shift_depth(self, marg_R, marg_P, new_R, new_P):
        keep = []
        for it in self.feature:
            if it.start_frame != 0:
                it.start_frame -= 1
                keep.append(it); continue
            uv_i = it.feature_per_frame[0].point
            depth = -1.0
            if it.feature_per_frame[0].depth > 0:
                depth = it.feature_per_frame[0].depth
            elif it.estimated_depth > 0:
                depth = it.estimated_depth
            del it.feature_per_frame[0]
            if len(it.feature_per_frame) < 2:
                continue
            pts_j = new_R.T @ (marg_R @ (uv_i * depth) + marg_P - new_P)
            if it.feature_per_frame[0].depth > 0:
                it.estimated_depth, it.lidar_depth_flag = it.feature_per_frame[0].depth, True
            elif pts_j[2] > 0:
                it.estimated_depth, it.lidar_depth_flag = float(pts_j[2]), False
            else:
                it.estimated_depth, it.lidar_depth_flag = INIT_DEPTH, False
            keep.append(it)
        self.feature = keep

    def remove_front(self, frame_count):
        keep = []
        for it in self.feature:
            if it.start_frame == frame_count:
                it.start_frame -= 1
            else:
                j = WINDOW_SIZE - 1 - it.start_frame
                if it.end_frame() >= frame_count - 1:
                    del it.feature_per_frame[j]
                    if len(it.feature_per_frame) == 0:
                        continue
            keep.append(it)
        self.feature = keep


def R2ypr(R):
    """Utility::R2ypr (utility.h:70-85), degrees"""
    n, o, a = R[:, 0], R[:, 1], R[:, 2]
    y = np.arctan2(n[1], n[0])
    p = np.arctan2(-n[2], n[0] * np.cos(y) + n[1] * np.sin(y))
    r = np.arctan2(a[0] * np.sin(y) - a[1] * np.cos(y), -o[0] * np.sin(y) + o[1] * np.cos(y))
    return np.array([y, p, r]) / np.pi * 180.0


def ypr2R(ypr):
    """Utility::ypr2R (utility.h:88-117), degrees"""
    y, p, r = np.asarray(ypr, dtype=np.float64) / 180.0 * np.pi
    Rz = np.array([[np.cos(y), -np.sin(y), 0], [np.sin(y), np.cos(y), 0], [0, 0, 1]])
    Ry = np.array([[np.cos(p), 0, np.sin(p)], [0, 1, 0], [-np.sin(p), 0, np.cos(p)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(r), -np.sin(r)], [0, np.sin(r), np.cos(r)]])
    return Rz @ Ry @ Rx


def g2R(g):
    """Utility::g2R (utility.cpp:3-13): the rotation taking g to +z (Quaternion::FromTwoVectors), yaw removed"""
    a = np.asarray(g, dtype=np.float64) / np.linalg.norm(g)
    b = np.array([0.0, 0.0, 1.0])
    c = float(a @ b)
    if c < -1.0 + 1e-12:                      # antiparallel: FromTwoVectors falls back to an SVD-chosen axis; any axis orthogonal to a
        ax = np.cross(a, np.array([1.0, 0, 0])); ax /= np.linalg.norm(ax)
        q = np.array([ax[0], ax[1], ax[2], 0.0])
    else:
        ax = np.cross(a, b)
        sN = np.sqrt((1.0 + c) * 2.0)
        q = np.array([ax[0] / sN, ax[1] / sN, ax[2] / sN, sN * 0.5])
    R0 = synth.q_to_R(q)
    yaw = R2ypr(R0)[0]
    return ypr2R(np.array([-yaw, 0.0, 0.0])) @ R0


class Integration:
    """≙ IntegrationBase: the sample buffers + linearisation biases; the pre-integrated row is computed on demand by the library's host
    routine (vilf_imu_preintegrate ≙ push_back / propagate / midPointIntegration, integration_base.h:30-158) and cached until a sample
    or the biases change."""
    noise = abi.ImuNoise(synth.ACC_N, synth.GYR_N, synth.ACC_W, synth.GYR_W)

    def __init__(self, acc_0, gyr_0, ba, bg):
        self.acc = [np.array(acc_0, dtype=np.float64)]; self.gyr = [np.array(gyr_0, dtype=np.float64)]
        self.dt = []
        self._ba, self._bg = np.array(ba, dtype=np.float64), np.array(bg, dtype=np.float64)
        self._row = None

    ba = property(lambda self: self._ba)
    bg = property(lambda self: self._bg)

    @ba.setter
    def ba(self, v):
        self._ba, self._row = np.array(v, dtype=np.float64), None

    @bg.setter
    def bg(self, v):
        self._bg, self._row = np.array(v, dtype=np.float64), None

    def push_back(self, dt, acc, gyr):
        self.dt.append(float(dt)); self.acc.append(np.array(acc, dtype=np.float64)); self.gyr.append(np.array(gyr, dtype=np.float64))
        self._row = None

    def row(self):
        if self._row is None:
            if not self.dt:
                r = np.zeros(abi.IMU_DOUBLES); r[synth.IMU_OFF["delta_q"][0] + 3] = 1.0
            else:
                from .estimator import imu_preintegrate
                pre = imu_preintegrate(self.noise, self.acc[0], self.gyr[0], self._ba, self._bg, np.array(self.dt), np.array(self.acc[1:]), np.array(self.gyr[1:]))
                r = np.frombuffer(pre, dtype=np.float64).copy()
            self._row = r
        return self._row


class SlidingWindowEstimator:
    """Steady-state (NON_LINEAR) Estimator loop. `backend.solve(window) -> abi.WindowResult`, `backend.marginalize(window, result)`."""

    INITIAL, NON_LINEAR = 0, 1

    def __init__(self, opts, backend):
        self.o, self.backend = opts, backend
        self.trajectory = []            # (stamp, P[3], q[4] xyzw) of the newest frame after every solved frame
        self.flags = []
        self.summaries = []
        self.events = []                # per process_image call: "fill" | "solved" | "reboot"
        self.n_reboots = 0
        self.initial_ok, self.alignment = None, None
        self.last_R = self.last_P = self.last_R0 = self.last_P0 = None
        self.clear_state()

    # estimator.cpp:36-88 (+ setParameter :24-34)
    def clear_state(self):
        opts = self.o
        n = WINDOW_SIZE + 1
        self.Ps = np.zeros((n, 3)); self.Vs = np.zeros((n, 3)); self.Rs = np.tile(np.eye(3), (n, 1, 1))
        self.Bas = np.zeros((n, 3)); self.Bgs = np.zeros((n, 3))
        self.stamps = np.zeros(n)
        self.pre = [None] * n
        self.lidar = [(np.array([0, 0, 0, 1.0]), np.zeros(3)) for _ in range(n)]
        self.g = np.array(opts.G[:])
        self.ric = np.array(opts.RIC[:]).reshape(3, 3); self.tic = np.array(opts.TIC[:])
        self.td = 0.0
        self.f = FeatureManager()
        self.frame_count = 0
        self.first_imu = False
        self.acc_0 = np.zeros(3); self.gyr_0 = np.zeros(3)
        self.marginalization_flag = MARGIN_OLD
        self.solver_flag = self.INITIAL
        self.failure_occur = 0

    # estimator.cpp:90-101
    def process_odometry(self, q, t):
        if self.frame_count != 0:
            self.lidar[self.frame_count] = (np.array(q, dtype=np.float64), np.array(t, dtype=np.float64))

    # estimator.cpp:103-137
    def process_imu(self, dt, acc, gyr):
        acc, gyr = np.array(acc, dtype=np.float64), np.array(gyr, dtype=np.float64)
        if not self.first_imu:
            self.first_imu = True
            self.acc_0, self.gyr_0 = acc, gyr
        j = self.frame_count
        if self.pre[j] is None:
            self.pre[j] = Integration(self.acc_0, self.gyr_0, self.Bas[j], self.Bgs[j])
        if j != 0:
            self.pre[j].push_back(dt, acc, gyr)
            un_acc_0 = self.Rs[j] @ (self.acc_0 - self.Bas[j]) - self.g
            un_gyr = 0.5 * (self.gyr_0 + gyr) - self.Bgs[j]
            th = un_gyr * dt
            dq = np.array([th[0] / 2, th[1] / 2, th[2] / 2, 1.0])                  # Utility::deltaQ: (1, theta / 2), NOT normalised; Eigen's
            self.Rs[j] = self.Rs[j] @ synth.q_to_R(dq)                             # toRotationMatrix() does not normalise either (:127)
            un_acc = 0.5 * (un_acc_0 + self.Rs[j] @ (acc - self.Bas[j]) - self.g)
            self.Ps[j] = self.Ps[j] + dt * self.Vs[j] + 0.5 * dt * dt * un_acc
            self.Vs[j] = self.Vs[j] + dt * un_acc
        self.acc_0, self.gyr_0 = acc, gyr

    # estimator.cpp:139-234. `init_state` = (P, R, V, ba, bg) of this frame while the window is being filled (replaces the SfM start-up)
    def process_image(self, image, stamp, init_state=None, sfm_frames=None):
        """`sfm_frames` (only looked at when the window has just filled): the output of initialStructure's SfM + PnP stage (estimator.cpp:237-371),
        a time-ordered list of dict(stamp, R = c0_R_bk, T = c0_T_ck up to scale, pre = that frame's Integration) -> visualInitialAlign runs first.
        Returns the WindowResult of a solved frame, None while the window fills, "reboot" when failureDetection() fired."""
        win = self.begin_image(image, stamp, init_state, sfm_frames)
        if win is None:
            return None
        res = self.backend.solve(win)
        self.backend.marginalize(win, res)
        return self.end_image(res)

    def begin_image(self, image, stamp, init_state=None, sfm_frames=None):
        """processImage up to the solve: returns the window description (vector2double + the factor walk) or None when this frame only fills the window"""
        j = self.frame_count
        keyframe = self.f.add_feature_check_parallax(j, dict(sorted(image.items())), self.td)
        self.marginalization_flag = MARGIN_OLD if keyframe else MARGIN_SECOND_NEW
        self.stamps[j] = stamp
        if init_state is not None:
            self.Ps[j], self.Rs[j], self.Vs[j], self.Bas[j], self.Bgs[j] = [np.array(x, dtype=np.float64) for x in init_state]
            if self.pre[j] is not None:       # ≙ repropagate() with the start-up biases (initial_aligment.cpp / estimator.cpp:437-440)
                self.pre[j].ba, self.pre[j].bg = self.Bas[j].copy(), self.Bgs[j].copy()
        if self.solver_flag == self.INITIAL:
            if j < WINDOW_SIZE:
                self.frame_count += 1
                self.events.append("fill")
                return None
            if sfm_frames is not None:
                self.initial_ok = self.visual_initial_align(sfm_frames)
                if not self.initial_ok:             # the reference slides the window and retries on the next frame (:201-216)
                    self.slide_window()
                    self.events.append("fill")
                    return None
            self._initial_frame = True              # initialStructure() succeeded (:188-206): no failureDetection on this frame
            self.solver_flag = self.NON_LINEAR
        else:
            self._initial_frame = False
        # solveOdometry (:492-503)
        self.f.triangulate(self.Ps, self.Rs, self.tic, self.ric)
        return self.make_window()

    def end_image(self, res):
        """processImage after optimization(): double2vector's state, failureDetection -> reboot, slideWindow, removeFailures, the trajectory row"""
        self.Ps, self.Rs, self.Vs, self.Bas, self.Bgs = res.Ps.copy(), res.Rs.copy(), res.Vs.copy(), res.Bas.copy(), res.Bgs.copy()
        self.f.set_depth(res.para_feature)
        self.failure_occur = 0                      # consumed by double2vector (:554-559)
        if not self._initial_frame and self.failure_detection():
            self.failure_occur = 1                  # :212-220; clearState() resets it at once (:80), so the gauge override of double2vector is never
            self.clear_state()                      # reached through this loop — the ABI keeps it (vilf_window_in.gauge_R0 / gauge_P0)
            self.n_reboots += 1
            if hasattr(self.backend, "reset"):
                self.backend.reset()
            self.events.append("reboot")
            return "reboot"
        self.summaries.append(res.summary)
        self.slide_window()
        self.f.remove_failures()
        self.last_R, self.last_P = self.Rs[WINDOW_SIZE].copy(), self.Ps[WINDOW_SIZE].copy()
        self.last_R0, self.last_P0 = self.Rs[0].copy(), self.Ps[0].copy()
        q = synth.R_to_q(self.Rs[WINDOW_SIZE])
        self.trajectory.append((self.stamps[WINDOW_SIZE], self.Ps[WINDOW_SIZE].copy(), q))
        self.flags.append(self.marginalization_flag)
        self.events.append("solved")
        return res

    # estimator.cpp:640-686 (the two commented-out returns — too few tracked features, a big rotation — stay out)
    def failure_detection(self):
        W = WINDOW_SIZE
        if np.linalg.norm(self.Bas[W]) > 2.5 or np.linalg.norm(self.Bgs[W]) > 1.0:
            return True
        if np.linalg.norm(self.Ps[W] - self.last_P) > 5:
            return True
        return bool(abs(self.Ps[W][2] - self.last_P[2]) > 1)

    # estimator.cpp:383-459
    def visual_initial_align(self, frames):
        n, m = len(frames), len(frames) - 1
        S = max(len(fr["pre"].dt) for fr in frames[1:])
        dt = np.zeros((m, S)); acc = np.zeros((m, S, 3)); gyr = np.zeros((m, S, 3)); ns = np.zeros(m, dtype=np.int32)
        acc_0 = np.zeros((m, 3)); gyr_0 = np.zeros((m, 3)); lin_ba = np.zeros((m, 3)); lin_bg = np.zeros((m, 3))
        for k, fr in enumerate(frames[1:]):
            it = fr["pre"]
            ns[k] = len(it.dt); dt[k, :ns[k]] = it.dt; acc[k, :ns[k]] = it.acc[1:]; gyr[k, :ns[k]] = it.gyr[1:]
            acc_0[k], gyr_0[k], lin_ba[k], lin_bg[k] = it.acc[0], it.gyr[0], it.ba, it.bg
        r = self.backend.align(dict(frame_R=np.array([fr["R"] for fr in frames]), frame_T=np.array([fr["T"] for fr in frames]), acc_0=acc_0, gyr_0=gyr_0,
                                    lin_ba=lin_ba, lin_bg=lin_bg, n_samples=ns, dt=dt, acc=acc, gyr=gyr, bgs0=self.Bgs[0].copy()))
        self.alignment = r
        self.Bgs = self.Bgs + r["delta_bg"]                                   # solveGyroscopeBias (initial_aligment.cpp:29-30)
        for fr in frames[1:]:
            fr["pre"].ba, fr["pre"].bg = np.zeros(3), self.Bgs[0].copy()      # repropagate(0, Bgs[0]) (:32-36)
        if not r["ok"]:
            return False
        x, g = r["x"], r["g"].copy()
        by_stamp = {fr["stamp"]: fr for fr in frames}
        fc = self.frame_count
        for i in range(fc + 1):                                               # :396-403
            fr = by_stamp[self.stamps[i]]
            self.Ps[i], self.Rs[i] = np.array(fr["T"], dtype=np.float64), np.array(fr["R"], dtype=np.float64)
            fr["is_key_frame"] = True
        for it in self.f.feature:                                             # clearDepth(-1) (:405-408, feature_manager.cpp:181-192)
            if self._used_f(it):
                it.estimated_depth, it.lidar_depth_flag = -1.0, False
        self.f.triangulate(self.Ps, self.Rs, np.zeros(3), self.ric)           # on the camera positions, no tic (:410-416)
        s = x[-1]
        for i in range(WINDOW_SIZE + 1):                                      # :419-422
            if self.pre[i] is not None:
                self.pre[i].ba, self.pre[i].bg = np.zeros(3), self.Bgs[i].copy()
        P0 = s * self.Ps[0] - self.Rs[0] @ self.tic
        for i in range(fc, -1, -1):                                           # :423-424
            self.Ps[i] = s * self.Ps[i] - self.Rs[i] @ self.tic - P0
        kv = -1
        for fr in frames:                                                     # :425-434 (x is indexed by the key-frame counter, as the reference does)
            if fr.get("is_key_frame"):
                kv += 1
                self.Vs[kv] = np.array(fr["R"]) @ x[3 * kv: 3 * kv + 3]
        for it in self.f.feature:                                             # :435-441
            if self._used_f(it):
                it.estimated_depth *= s
        R0 = g2R(g)                                                           # :443-454
        yaw = R2ypr(R0 @ self.Rs[0])[0]
        R0 = ypr2R(np.array([-yaw, 0.0, 0.0])) @ R0
        self.g = R0 @ g
        for i in range(fc + 1):
            self.Ps[i] = R0 @ self.Ps[i]; self.Rs[i] = R0 @ self.Rs[i]; self.Vs[i] = R0 @ self.Vs[i]
        return True

    def _used_f(self, it):
        return self.f._used(it)

    # estimator.cpp:505-547 + the factor walk :722-794 -> the ABI's window description
    def make_window(self):
        n = WINDOW_SIZE + 1
        para_pose = np.zeros((n, 7)); para_sb = np.zeros((n, 9))
        for i in range(n):
            para_pose[i, :3] = self.Ps[i]; para_pose[i, 3:] = synth.R_to_q(self.Rs[i])
            para_sb[i, :3], para_sb[i, 3:6], para_sb[i, 6:] = self.Vs[i], self.Bas[i], self.Bgs[i]
        ex = np.concatenate([self.tic, synth.R_to_q(self.ric)])
        used = [it for it in self.f.feature if self.f._used(it)]
        depth = self.f.get_depth_vector()
        starts = np.array([it.start_frame for it in used], dtype=np.int32)
        const = np.array([1 if it.lidar_depth_flag else 0 for it in used], dtype=np.uint8)
        offs, pts = [0], []
        for it in used:
            pts.extend(fp.point for fp in it.feature_per_frame)
            offs.append(len(pts))
        imu = np.zeros((n, abi.IMU_DOUBLES)); imu[0, synth.IMU_OFF["delta_q"][0] + 3] = 1.0
        lid = np.zeros((n, 7)); lid[:, 3] = 1.0
        for k in range(1, n):
            imu[k] = self.pre[k].row()
            lid[k, :4], lid[k, 4:] = self.lidar[k]
        return abi.Window(para_pose, para_sb, ex, depth, const, starts, np.array(offs, dtype=np.int32), np.array(pts).reshape(-1, 3), imu,
                          lidar=lid, para_td=self.td, marginalization_flag=self.marginalization_flag)

    # estimator.cpp:1052-1186
    def slide_window(self):
        W = WINDOW_SIZE
        if self.marginalization_flag == MARGIN_OLD:
            back_R0, back_P0 = self.Rs[0].copy(), self.Ps[0].copy()
            for arr in (self.Ps, self.Vs, self.Rs, self.Bas, self.Bgs, self.stamps):
                arr[:W] = arr[1:].copy()
            self.pre = self.pre[1:] + [None]
            self.lidar = self.lidar[1:] + [(np.array([0, 0, 0, 1.0]), np.zeros(3))]
            self.pre[W] = Integration(self.acc_0, self.gyr_0, self.Bas[W], self.Bgs[W])
            # slideWindowOld (:1169-1186), solver_flag == NON_LINEAR
            R0, P0 = back_R0 @ self.ric, back_P0 + back_R0 @ self.tic
            R1, P1 = self.Rs[0] @ self.ric, self.Ps[0] + self.Rs[0] @ self.tic
            self.f.remove_back_shift_depth(R0, P0, R1, P1)
        else:
            last, prev = self.pre[W], self.pre[W - 1]
            for dt, a, w in zip(last.dt, last.acc[1:], last.gyr[1:]):
                prev.push_back(dt, a, w)
            for arr in (self.Ps, self.Vs, self.Rs, self.Bas, self.Bgs, self.stamps):
                arr[W - 1] = arr[W].copy()
            (qa, ta), (qb, tb) = self.lidar[W - 1], self.lidar[W]
            self.lidar[W - 1] = (synth.q_mul(qa, qb), synth.q_to_R(qa) @ tb + ta)         # merge the two LiDAR between-constraints (:1131-1134)
            self.pre[W] = Integration(self.acc_0, self.gyr_0, self.Bas[W], self.Bgs[W])
            self.lidar[W] = (np.array([0, 0, 0, 1.0]), np.zeros(3))
            self.f.remove_front(W)       # slideWindowNew (:1163-1167)


def write_tum(path, trajectory):
    """utility/visualization.cpp:159-172: `t x y z qx qy qz qw`, time relative to the first pose, precision 9 / 5."""
    t0 = trajectory[0][0]
    with open(path, "w") as fh:
        for t, P, q in trajectory:
            fh.write("%.9f %.5f %.5f %.5f %.5f %.5f %.5f %.5f\n" % (t - t0, P[0], P[1], P[2], q[0], q[1], q[2], q[3]))


# ---------------------------------------------------------------------------------------------------------------------
# synthetic multi-frame sequence (same motion / camera / IMU model as synth.make_window, feature tracks with ids)
def make_sequence(seed, n_frames, opts, max_cnt=150, lidar_depth_fraction=0.4, pixel_sigma=0.5 / 460.0, state_noise=(0.05, np.deg2rad(0.5), 0.05),
                  imu_noise_scale=1.0, bias_scale=1.0, yaw_amplitude=None, mean_speed=None, speed_modulation=0.0):
    """Returns a dict: stamps[n], imu[k] = (dt, acc[S,3], gyr[S,3]) for the interval ending at frame k (k >= 1, first-ever sample in
    imu0), images[k] = {feature_id: 8-vector}, lidar[k] = (q, t) relative LiDAR pose k-1 -> k, truth P/R/V, and `init[k]` = noisy
    (P, R, V, ba, bg) for every frame (stands in for the reference's SfM initialisation: the first WINDOW_SIZE + 1 frames, and the
    frames after a failureDetection reboot)."""
    rng = np.random.default_rng(seed)
    RIC = np.array(opts.RIC[:]).reshape(3, 3); TIC = np.array(opts.TIC[:])
    RCL = np.array(opts.RCL[:]).reshape(3, 3); TCL = np.array(opts.TCL[:])
    G = np.array(opts.G[:])
    S, dt = 10, 0.01
    nt = (n_frames - 1) * S + 1
    t = np.arange(nt) * dt
    speed = rng.uniform(8.0, 12.0)
    if mean_speed is not None:                # slow / stop-and-go drives: frames without enough parallax -> MARGIN_SECOND_NEW
        speed = float(mean_speed)
    As, ws = float(speed_modulation), 0.9     # speed(t) = speed (1 + As sin(ws t))
    Ay, wy, py = rng.uniform(0.05, 0.3), rng.uniform(0.3, 1.0), rng.uniform(0, 2 * np.pi)
    if yaw_amplitude is not None:
        Ay, wy = yaw_amplitude, 2.0
    Ap, wp, pp = rng.uniform(0.0, 0.03), rng.uniform(0.5, 2.0), rng.uniform(0, 2 * np.pi)
    Ar, wr, pr = rng.uniform(0.0, 0.03), rng.uniform(0.5, 2.0), rng.uniform(0, 2 * np.pi)
    yaw0 = rng.uniform(-np.pi, np.pi)

    def kin(tt):
        yaw = yaw0 + Ay * np.sin(wy * tt + py); dyaw = Ay * wy * np.cos(wy * tt + py)
        pit = Ap * np.sin(wp * tt + pp); dpit = Ap * wp * np.cos(wp * tt + pp)
        rol = Ar * np.sin(wr * tt + pr); drol = Ar * wr * np.cos(wr * tt + pr)
        R = synth.euler_R(yaw, pit, rol)
        w_b = np.stack([drol - dyaw * np.sin(pit), dpit * np.cos(rol) + dyaw * np.sin(rol) * np.cos(pit), -dpit * np.sin(rol) + dyaw * np.cos(rol) * np.cos(pit)], -1)
        sp = speed * (1.0 + As * np.sin(ws * tt)); dsp = speed * As * ws * np.cos(ws * tt)
        v = sp[..., None] * R[..., :, 0]
        a = dsp[..., None] * R[..., :, 0] + sp[..., None] * np.einsum('...ij,...j->...i', R, np.cross(w_b, np.array([1.0, 0, 0])))
        return R, w_b, v, a

    fine = 10
    tf = np.arange((nt - 1) * fine + 1) * (dt / fine)
    vf = kin(tf)[2]
    h = dt / fine
    seg = (vf[0:-2:2] + 4 * vf[1:-1:2] + vf[2::2]) * (h / 3.0)
    p2 = np.concatenate([np.zeros((1, 3)), np.cumsum(seg, axis=0)])
    P_imu = rng.uniform(-50, 50, 3) * np.array([1, 1, 0.02]) + p2[:: fine // 2][:nt]
    R_imu, w_b, v_imu, a_w = kin(t)
    ba_true = rng.normal(0, 0.02, 3) * bias_scale; bg_true = rng.normal(0, 0.002, 3) * bias_scale
    acc_m = np.einsum('tji,tj->ti', R_imu, a_w + G) + ba_true + rng.normal(0, synth.ACC_N, (nt, 3)) * imu_noise_scale
    gyr_m = w_b + bg_true + rng.normal(0, synth.GYR_N, (nt, 3)) * imu_noise_scale
    fidx = np.arange(n_frames) * S
    Pw, Rw, Vw = P_imu[fidx], R_imu[fidx], v_imu[fidx]
    Rc = Rw @ RIC
    Pc = Pw + np.einsum('kij,j->ki', Rw, TIC)

    images, active, next_id = [], {}, 0          # active: id -> (Xw, has_lidar_depth)
    for k in range(n_frames):
        img = {}
        for fid in list(active):
            Xw, has_d = active[fid]
            pc = Rc[k].T @ (Xw - Pc[k])
            if pc[2] < 1.0 or abs(pc[0] / pc[2]) > 1.3 or abs(pc[1] / pc[2]) > 0.6:
                del active[fid]; continue
            x, y = pc[0] / pc[2] + rng.normal(0, pixel_sigma), pc[1] / pc[2] + rng.normal(0, pixel_sigma)
            img[fid] = np.array([x, y, 1.0, synth.FX * x + synth.CX, synth.FY * y + synth.CY, 0.0, 0.0, pc[2] + rng.normal(0, 0.05) if has_d else -1.0])
        tries = 0
        while len(img) < max_cnt and tries < 20 * max_cnt:
            tries += 1
            u, v = rng.uniform(0, synth.IMG_W), rng.uniform(0, synth.IMG_H)
            d = rng.uniform(5.0, 50.0)
            ray = np.array([(u - synth.CX) / synth.FX, (v - synth.CY) / synth.FY, 1.0])
            Xw = Rc[k] @ (d * ray) + Pc[k]
            has_d = bool(rng.uniform() < lidar_depth_fraction)
            active[next_id] = (Xw, has_d)
            x, y = ray[0] + rng.normal(0, pixel_sigma), ray[1] + rng.normal(0, pixel_sigma)
            img[next_id] = np.array([x, y, 1.0, u, v, 0.0, 0.0, d + rng.normal(0, 0.05) if has_d else -1.0])
            next_id += 1
        images.append(img)

    Ril = RIC @ RCL; til = RIC @ TCL + TIC
    lidar = [None]
    for k in range(1, n_frames):
        Rij = Rw[k - 1].T @ Rw[k]; Pij = Rw[k - 1].T @ (Pw[k] - Pw[k - 1])
        Rl = Ril.T @ Rij @ Ril
        tl = Ril.T @ (Rij @ til + Pij - til)
        ql = synth.q_mul(synth.R_to_q(Rl), synth.q_exp(rng.normal(0, np.deg2rad(0.1), 3)))
        lidar.append((ql / np.linalg.norm(ql), tl + rng.normal(0, 0.02, 3)))

    sp, sr, sv = state_noise
    ba_est = ba_true + rng.normal(0, 0.005, 3); bg_est = bg_true + rng.normal(0, 0.0005, 3)
    init = []
    for k in range(n_frames):
        Rn = Rw[k] @ synth.q_to_R(synth.q_exp(rng.normal(0, sr, 3)))
        init.append((Pw[k] + rng.normal(0, sp, 3), Rn, Vw[k] + rng.normal(0, sv, 3), ba_est, bg_est))
    imu = [None] + [(dt, acc_m[(k - 1) * S + 1: k * S + 1], gyr_m[(k - 1) * S + 1: k * S + 1]) for k in range(1, n_frames)]
    return dict(stamps=fidx * dt, imu0=(acc_m[0], gyr_m[0]), imu=imu, images=images, lidar=lidar, init=init, P=Pw, R=Rw, V=Vw, ba=ba_true, bg=bg_true)


def make_sfm_frames(seq, opts, est, seed=0, scale=3.7, rot_noise=np.deg2rad(0.02), pos_noise=0.005):
    """Stand-in for initialStructure's SfM + PnP output (estimator.cpp:237-371) for the frames now in the window: body orientation and camera
    position of every frame in the frame of a reference camera c0 (the window's middle frame), positions divided by an unknown scale."""
    import copy
    rng = np.random.default_rng(seed + 4242)
    RIC = np.array(opts.RIC[:]).reshape(3, 3); TIC = np.array(opts.TIC[:])
    n = est.frame_count + 1
    Rw, Pw = seq["R"][:n], seq["P"][:n]
    l = n // 2
    R_c0_w = (Rw[l] @ RIC).T
    Pc = Pw + np.einsum('kij,j->ki', Rw, TIC)
    frames = []
    for k in range(n):
        frames.append(dict(stamp=est.stamps[k], R=R_c0_w @ Rw[k] @ synth.q_to_R(synth.q_exp(rng.normal(0, rot_noise, 3))),
                           T=(R_c0_w @ (Pc[k] - Pc[l]) + rng.normal(0, pos_noise, 3)) / scale, pre=copy.deepcopy(est.pre[k])))
    return frames


def drop_frames(seq, k0, n):
    """A sensor gap: camera / LiDAR frames k0 .. k0 + n - 1 never arrive. The IMU keeps running (the samples of the dropped intervals join the
    interval that ends at the next frame, as estimator_node.cpp:102-160 hands them over) and the LiDAR odometry of the next frame is relative to
    the last frame that did arrive. A gap of more than 5 m of travel trips failureDetection()'s translation gate (estimator.cpp:665-669)."""
    out = dict(seq)
    keep = [k for k in range(len(seq["images"])) if not (k0 <= k < k0 + n)]
    for key in ("stamps", "P", "R", "V"):
        out[key] = seq[key][keep]
    for key in ("images", "init"):
        out[key] = [seq[key][k] for k in keep]
    imu, lidar = list(seq["imu"]), list(seq["lidar"])
    dt = imu[k0][0]
    imu[k0 + n] = (dt, np.concatenate([imu[k][1] for k in range(k0, k0 + n + 1)]), np.concatenate([imu[k][2] for k in range(k0, k0 + n + 1)]))
    q, t = np.array([0, 0, 0, 1.0]), np.zeros(3)
    for k in range(k0, k0 + n + 1):               # T_(k0-1 -> k0+n) = T_(k0-1 -> k0) o ... o T_(k0+n-1 -> k0+n)
        qk, tk = lidar[k]
        q, t = synth.q_mul(q, qk), synth.q_to_R(q) @ tk + t
    lidar[k0 + n] = (q / np.linalg.norm(q), t)
    out["imu"] = [imu[k] for k in keep]; out["lidar"] = [lidar[k] for k in keep]
    return out


def _feed_measurements(est, seq, k):
    if k >= 1:
        dt, acc, gyr = seq["imu"][k]
        for a, w in zip(acc, gyr):
            est.process_imu(dt, a, w)
        est.process_odometry(*seq["lidar"][k])


def run_sequence(seq, opts, backend, n_frames=None, startup=None):
    """Feed a make_sequence() dict through SlidingWindowEstimator; returns the estimator (trajectory, flags, summaries).
    startup=None: while the estimator is INITIAL (the first WINDOW_SIZE + 1 frames, and again after a failureDetection reboot) the frames take
    seq["init"] (a state already initialised). startup=dict(make_sfm_frames kwargs): the window fills from a zero state and visualInitialAlign
    (SfM stand-in + VisualIMUAlignment on the back-end) initialises it."""
    est = SlidingWindowEstimator(opts, backend)
    n = len(seq["images"]) if n_frames is None else n_frames
    est.process_imu(0.0, *seq["imu0"])                 # first sample: only latches acc_0 / gyr_0 (frame_count == 0)
    for k in range(n):
        _feed_measurements(est, seq, k)
        if startup is None:
            est.process_image(seq["images"][k], seq["stamps"][k], seq["init"][k] if est.solver_flag == est.INITIAL else None)
        elif k == WINDOW_SIZE:
            est.stamps[k] = seq["stamps"][k]
            est.process_image(seq["images"][k], seq["stamps"][k], None, sfm_frames=make_sfm_frames(seq, opts, est, **startup))
        else:
            est.process_image(seq["images"][k], seq["stamps"][k], None)
    return est


class _SlotBackend:
    """what SlidingWindowEstimator needs from a back-end when the solve itself is driven from outside (run_sequences_lockstep)"""

    def __init__(self, solver, slot):
        self.s, self.slot, self.drop_prior = solver, slot, True

    def reset(self):                                    # ≙ clearState(): last_marginalization_info = nullptr (estimator.cpp:72-77)
        self.drop_prior = True


def run_sequences_lockstep(seqs, opts, solver, n_frames=None, on_frame=None):
    """len(seqs) independent sequence segments, one SlidingWindowEstimator each, stepped frame by frame as ONE device batch: per frame one
    vilf_batch_upload of the segments' windows (slot i = segment i, so every segment's prior stays in its slot on the device), one
    vilf_batch_solve, one vilf_batch_marginalize, one download. A segment that has no window to solve in a frame (its window is still
    filling, at the start or after a failureDetection reboot) contributes its previous window as a placeholder whose result is dropped; its
    slot's prior is cleared before its next real solve. Returns the estimators; `solver.lockstep_stats` counts the marginalization paths."""
    S = len(seqs)
    ests = [SlidingWindowEstimator(opts, _SlotBackend(solver, i)) for i in range(S)]
    n = min(len(q["images"]) for q in seqs) if n_frames is None else n_frames
    for est, seq in zip(ests, seqs):
        est.process_imu(0.0, *seq["imu0"])
    placeholder = [None] * S
    stats = dict(frames=0, new_prior=0, amm_cholesky=0, kept_cholesky=0, unchanged=0)
    for k in range(n):
        wins = [None] * S
        for i, (est, seq) in enumerate(zip(ests, seqs)):
            _feed_measurements(est, seq, k)
            wins[i] = est.begin_image(seq["images"][k], seq["stamps"][k], seq["init"][k] if est.solver_flag == est.INITIAL else None)
        live = [i for i in range(S) if wins[i] is not None]
        if not live:
            continue
        for i in live:
            placeholder[i] = wins[i]
            if ests[i].backend.drop_prior:
                solver.set_prior(None, i)
                ests[i].backend.drop_prior = False
        fill = placeholder[live[0]]
        batch = [wins[i] if wins[i] is not None else (placeholder[i] if placeholder[i] is not None else fill) for i in range(S)]
        solver.batch_upload(batch)
        solver.batch_solve(sync=False)
        solver.batch_marginalize(sync=True)
        results = solver.batch_download()
        if len(live) == S:
            st = solver.marginalize_stats()
            stats["frames"] += 1
            for key in ("new_prior", "amm_cholesky", "kept_cholesky", "unchanged"):
                stats[key] += st[key]
        for i in range(S):
            if wins[i] is None:
                ests[i].backend.drop_prior = True       # the placeholder's marginalization wrote a prior into this slot
                continue
            r = ests[i].end_image(results[i])
            if on_frame is not None:
                on_frame(i, k, ests[i], r)
    solver.lockstep_stats = stats
    return ests


def make_alignment_case(seed, opts, n_frames=14, scale=3.7, rot_noise=np.deg2rad(0.05), pos_noise=0.01, **seq_kw):
    """Inputs of VisualIMUAlignment (initial_aligment.cpp:199) from a synthetic drive: what the reference's SfM hands over — every
    frame's body orientation c0_R_bk and camera position c0_T_ck in the frame of a reference camera c0, positions up to an unknown
    scale — plus the raw IMU samples of every interval, first integrated at zero biases. Returns (inputs dict, truth dict)."""
    seq = make_sequence(seed, n_frames, opts, max_cnt=1, **seq_kw)
    rng = np.random.default_rng(seed + 77)
    RIC = np.array(opts.RIC[:]).reshape(3, 3); TIC = np.array(opts.TIC[:]); G = np.array(opts.G[:])
    Rw, Pw, Vw = seq["R"], seq["P"], seq["V"]
    l = n_frames // 2
    R_c0_w = (Rw[l] @ RIC).T
    Pc = Pw + np.einsum('kij,j->ki', Rw, TIC)
    frame_R = np.stack([R_c0_w @ Rw[k] @ synth.q_to_R(synth.q_exp(rng.normal(0, rot_noise, 3))) for k in range(n_frames)])
    frame_T = np.stack([(R_c0_w @ (Pc[k] - Pc[l]) + rng.normal(0, pos_noise, 3)) / scale for k in range(n_frames)])
    m = n_frames - 1
    S = max(len(seq["imu"][k][1]) for k in range(1, n_frames)) + 2
    dt = np.zeros((m, S)); acc = np.zeros((m, S, 3)); gyr = np.zeros((m, S, 3)); ns = np.zeros(m, dtype=np.int32)
    acc_0 = np.zeros((m, 3)); gyr_0 = np.zeros((m, 3))
    for k in range(1, n_frames):
        d, a, g = seq["imu"][k]
        ns[k - 1] = len(a); dt[k - 1, :len(a)] = d; acc[k - 1, :len(a)] = a; gyr[k - 1, :len(a)] = g
        acc_0[k - 1], gyr_0[k - 1] = (seq["imu0"] if k == 1 else (seq["imu"][k - 1][1][-1], seq["imu"][k - 1][2][-1]))
    inputs = dict(frame_R=frame_R, frame_T=frame_T, acc_0=acc_0, gyr_0=gyr_0, lin_ba=np.zeros((m, 3)), lin_bg=np.zeros((m, 3)), n_samples=ns,
                  dt=dt, acc=acc, gyr=gyr, bgs0=np.zeros(3))
    truth = dict(scale=scale, g=R_c0_w @ G, v_body=np.einsum('kji,kj->ki', Rw, Vw), bg=seq["bg"], ba=seq["ba"], l=l)
    return inputs, truth

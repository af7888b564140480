"""Seeded synthetic sliding-window generator.

KITTI data cannot be shipped or read in this pipeline, so every config is driven by synthetic windows whose
shapes follow config/kitti/kitti_config.yaml: 11 frames @10 Hz, IMU @100 Hz with the yaml noise densities,
~190 tracked features (~1.5 k reprojection factors, 40 % with LiDAR depth held constant), 10 IMU factors,
10 LiDAR between-factors and a dense marginalization prior (n = 75).

Pure numpy; it does NOT use the oracle (the pre-integration below is an independent restatement of
IntegrationBase::midPointIntegration, integration_base.h:54-128, vectorised over the frame intervals).
"""
import numpy as np
from .abi import Window, make_prior, IMU_DOUBLES, IMU_OFF, MARGIN_OLD

# config/kitti/kitti_config.yaml:29-40,78-82
FX, FY, CX, CY = 707.0912, 707.0912, 601.8873, 183.1104
IMG_W, IMG_H = 1226.0, 370.0
ACC_N, GYR_N, ACC_W, GYR_W = 0.08, 0.04, 0.00004, 2.0e-6


# ---- small quaternion helpers (x y z w order, Hamilton product) -------------------------------------------
def q_mul(a, b):
    ax, ay, az, aw = a[..., 0], a[..., 1], a[..., 2], a[..., 3]
    bx, by, bz, bw = b[..., 0], b[..., 1], b[..., 2], b[..., 3]
    return np.stack([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx,
                     aw * bw - ax * bx - ay * by - az * bz], axis=-1)


def q_conj(a):
    return a * np.array([-1.0, -1.0, -1.0, 1.0])


def q_to_R(q):
    x, y, z, w = q[..., 0], q[..., 1], q[..., 2], q[..., 3]
    R = np.empty(q.shape[:-1] + (3, 3))
    R[..., 0, 0] = 1 - 2 * (y * y + z * z); R[..., 0, 1] = 2 * (x * y - z * w); R[..., 0, 2] = 2 * (x * z + y * w)
    R[..., 1, 0] = 2 * (x * y + z * w); R[..., 1, 1] = 1 - 2 * (x * x + z * z); R[..., 1, 2] = 2 * (y * z - x * w)
    R[..., 2, 0] = 2 * (x * z - y * w); R[..., 2, 1] = 2 * (y * z + x * w); R[..., 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def R_to_q(R):
    """Shepperd's method (same branch structure as Eigen's Quaterniond(Matrix3d))."""
    R = np.asarray(R, dtype=np.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if t > 0:
        t = np.sqrt(t + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0] = (R[2, 1] - R[1, 2]) * t
        q[1] = (R[0, 2] - R[2, 0]) * t
        q[2] = (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (R[k, j] - R[j, k]) * t
        q[j] = (R[j, i] + R[i, j]) * t
        q[k] = (R[k, i] + R[i, k]) * t
    return q


def q_exp(v):
    """rotation vector -> unit quaternion (x y z w)"""
    v = np.asarray(v, dtype=np.float64)
    th = np.linalg.norm(v, axis=-1, keepdims=True)
    small = th < 1e-12
    k = np.where(small, 0.5, np.sin(0.5 * th) / np.where(small, 1.0, th))
    return np.concatenate([k * v, np.cos(0.5 * th)], axis=-1)


def skew(v):
    z = np.zeros(v.shape[:-1])
    return np.stack([np.stack([z, -v[..., 2], v[..., 1]], -1),
                     np.stack([v[..., 2], z, -v[..., 0]], -1),
                     np.stack([-v[..., 1], v[..., 0], z], -1)], -2)


def euler_R(yaw, pitch, roll):
    cy, sy, cp, sp, cr, sr = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
    R = np.empty(np.shape(yaw) + (3, 3))
    R[..., 0, 0] = cy * cp; R[..., 0, 1] = cy * sp * sr - sy * cr; R[..., 0, 2] = cy * sp * cr + sy * sr
    R[..., 1, 0] = sy * cp; R[..., 1, 1] = sy * sp * sr + cy * cr; R[..., 1, 2] = sy * sp * cr - cy * sr
    R[..., 2, 0] = -sp; R[..., 2, 1] = cp * sr; R[..., 2, 2] = cp * cr
    return R


# ---- mid-point pre-integration, vectorised over K intervals -------------------------------------------
def preintegrate(acc, gyr, dt, lin_ba, lin_bg, noise=(ACC_N, GYR_N, ACC_W, GYR_W)):
    """acc, gyr: (K, S+1, 3) samples (index 0 = acc_0/gyr_0 of the interval); dt scalar; lin_b*: (K,3).
    Returns (K, 467) rows in vilf_imu_preint layout. Restates integration_base.h:54-158."""
    K, S1, _ = acc.shape
    acc_n, gyr_n, acc_w, gyr_w = noise
    nd = np.concatenate([np.full(3, acc_n ** 2), np.full(3, gyr_n ** 2), np.full(3, acc_n ** 2), np.full(3, gyr_n ** 2),
                         np.full(3, acc_w ** 2), np.full(3, gyr_w ** 2)])
    dp = np.zeros((K, 3)); dv = np.zeros((K, 3)); dq = np.tile(np.array([0, 0, 0, 1.0]), (K, 1))
    J = np.tile(np.eye(15), (K, 1, 1)); P = np.zeros((K, 15, 15))
    I3 = np.eye(3)
    sum_dt = 0.0
    for s in range(1, S1):
        a0, a1, g0, g1 = acc[:, s - 1], acc[:, s], gyr[:, s - 1], gyr[:, s]
        Rd = q_to_R(dq)
        un_acc_0 = np.einsum('kij,kj->ki', Rd, a0 - lin_ba)
        un_gyr = 0.5 * (g0 + g1) - lin_bg
        dq_step = np.concatenate([un_gyr * dt / 2, np.ones((K, 1))], axis=-1)
        rq = q_mul(dq, dq_step)                                  # NOT normalised yet (toRotationMatrix of it is used below)
        Rr = q_to_R(rq)
        # Eigen applies the un-normalised quaternion through _transformVector: v + 2w(u x v) + 2 u x (u x v)
        u = rq[:, :3]; w = rq[:, 3:4]; vv = a1 - lin_ba
        uv = 2 * np.cross(u, vv)
        un_acc_1 = vv + w * uv + np.cross(u, uv)
        un_acc = 0.5 * (un_acc_0 + un_acc_1)
        rp = dp + dv * dt + 0.5 * un_acc * dt * dt
        rv = dv + un_acc * dt
        Rw = skew(0.5 * (g0 + g1) - lin_bg); Ra0 = skew(a0 - lin_ba); Ra1 = skew(a1 - lin_ba)
        F = np.zeros((K, 15, 15)); V = np.zeros((K, 15, 18))
        ImRw = I3 - Rw * dt
        F[:, 0:3, 0:3] = I3
        F[:, 0:3, 3:6] = -0.25 * Rd @ Ra0 * dt * dt + -0.25 * Rr @ Ra1 @ ImRw * dt * dt
        F[:, 0:3, 6:9] = I3 * dt
        F[:, 0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[:, 0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[:, 3:6, 3:6] = ImRw
        F[:, 3:6, 12:15] = -1.0 * I3 * dt
        F[:, 6:9, 3:6] = -0.5 * Rd @ Ra0 * dt + -0.5 * Rr @ Ra1 @ ImRw * dt
        F[:, 6:9, 6:9] = I3
        F[:, 6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[:, 6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[:, 9:12, 9:12] = I3
        F[:, 12:15, 12:15] = I3
        V[:, 0:3, 0:3] = 0.25 * Rd * dt * dt
        V[:, 0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[:, 0:3, 6:9] = 0.25 * Rr * dt * dt
        V[:, 0:3, 9:12] = V[:, 0:3, 3:6]
        V[:, 3:6, 3:6] = 0.5 * I3 * dt
        V[:, 3:6, 9:12] = 0.5 * I3 * dt
        V[:, 6:9, 0:3] = 0.5 * Rd * dt
        V[:, 6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[:, 6:9, 6:9] = 0.5 * Rr * dt
        V[:, 6:9, 9:12] = V[:, 6:9, 3:6]
        V[:, 9:12, 12:15] = I3 * dt
        V[:, 12:15, 15:18] = I3 * dt
        J = F @ J
        P = F @ P @ np.transpose(F, (0, 2, 1)) + (V * nd) @ np.transpose(V, (0, 2, 1))
        dp, dv = rp, rv
        dq = rq / np.linalg.norm(rq, axis=-1, keepdims=True)
        sum_dt += dt
    out = np.zeros((K, IMU_DOUBLES))
    out[:, 0] = sum_dt
    out[:, 1:4] = dp; out[:, 4:8] = dq; out[:, 8:11] = dv
    out[:, 11:14] = lin_ba; out[:, 14:17] = lin_bg
    out[:, 17:242] = J.reshape(K, 225); out[:, 242:467] = P.reshape(K, 225)
    return out


class SynthConfig:
    def __init__(self, n_frames=11, n_features=190, const_fraction=0.4, use_lidar=True, with_prior=True,
                 imu_rate=100.0, frame_rate=10.0, pixel_sigma=0.5 / 460.0, marginalization_flag=MARGIN_OLD,
                 state_noise=(0.05, np.deg2rad(0.5), 0.05), early_end_fraction=0.15, pitch_offset=0.0):
        self.n_frames = n_frames
        self.pitch_offset = pitch_offset      # radians added to the pitch profile (a climb; near +-pi/2: the Euler-singular branch of double2vector, estimator.cpp:567-574)
        self.n_features = n_features
        self.const_fraction = const_fraction
        self.use_lidar = use_lidar
        self.with_prior = with_prior
        self.imu_rate = imu_rate
        self.frame_rate = frame_rate
        self.pixel_sigma = pixel_sigma
        self.marginalization_flag = marginalization_flag
        self.state_noise = state_noise
        self.early_end_fraction = early_end_fraction


def make_window(seed, opts, cfg=None):
    """One seeded synthetic window. `opts` is an abi.Options (extrinsics, G). Returns (Window, prior_or_None, truth)."""
    cfg = cfg or SynthConfig()
    rng = np.random.default_rng(seed)
    NF = cfg.n_frames
    RIC = np.array(opts.RIC[:]).reshape(3, 3); TIC = np.array(opts.TIC[:])
    RCL = np.array(opts.RCL[:]).reshape(3, 3); TCL = np.array(opts.TCL[:])
    G = np.array(opts.G[:])
    S = int(round(cfg.imu_rate / cfg.frame_rate))
    dt = 1.0 / cfg.imu_rate
    nt = (NF - 1) * S + 1
    t = np.arange(nt) * dt

    # trajectory: forward motion along body x with yaw / pitch / roll sinusoids
    speed = rng.uniform(8.0, 12.0)
    Ay, wy, py = rng.uniform(0.05, 0.3), rng.uniform(0.3, 1.0), rng.uniform(0, 2 * np.pi)
    Ap, wp, pp = rng.uniform(0.0, 0.03), rng.uniform(0.5, 2.0), rng.uniform(0, 2 * np.pi)
    Ar, wr, pr = rng.uniform(0.0, 0.03), rng.uniform(0.5, 2.0), rng.uniform(0, 2 * np.pi)
    yaw0 = rng.uniform(-np.pi, np.pi)

    def angles(tt):
        yaw = yaw0 + Ay * np.sin(wy * tt + py); dyaw = Ay * wy * np.cos(wy * tt + py)
        pit = cfg.pitch_offset + Ap * np.sin(wp * tt + pp); dpit = Ap * wp * np.cos(wp * tt + pp)
        rol = Ar * np.sin(wr * tt + pr); drol = Ar * wr * np.cos(wr * tt + pr)
        return yaw, pit, rol, dyaw, dpit, drol

    def kin(tt):
        yaw, pit, rol, dyaw, dpit, drol = angles(tt)
        R = euler_R(yaw, pit, rol)
        w_b = np.stack([drol - dyaw * np.sin(pit),
                        dpit * np.cos(rol) + dyaw * np.sin(rol) * np.cos(pit),
                        -dpit * np.sin(rol) + dyaw * np.cos(rol) * np.cos(pit)], -1)
        ex = np.array([1.0, 0, 0])
        v = speed * R[..., :, 0]
        a = speed * np.einsum('...ij,...j->...i', R, np.cross(w_b, ex))
        return R, w_b, v, a

    # position by composite Simpson on a 10x finer grid
    fine = 10
    tf = np.arange((nt - 1) * fine + 1) * (dt / fine)
    _, _, vf, _ = kin(tf)
    h = dt / fine
    seg = (vf[0:-2:2] + 4 * vf[1:-1:2] + vf[2::2]) * (h / 3.0)            # integrals over 2h
    p2 = np.concatenate([np.zeros((1, 3)), np.cumsum(seg, axis=0)])       # at tf[::2]
    p0 = rng.uniform(-50, 50, 3) * np.array([1, 1, 0.02])
    P_imu = p0 + p2[:: fine // 2][:nt]
    R_imu, w_b, v_imu, a_w = kin(t)

    ba_true = rng.normal(0, 0.02, 3); bg_true = rng.normal(0, 0.002, 3)
    acc_m = np.einsum('tji,tj->ti', R_imu, a_w + G) + ba_true + rng.normal(0, ACC_N, (nt, 3))
    gyr_m = w_b + bg_true + rng.normal(0, GYR_N, (nt, 3))

    fidx = np.arange(NF) * S
    Pw, Rw, Vw = P_imu[fidx], R_imu[fidx], v_imu[fidx]
    Qw = np.stack([R_to_q(Rw[i]) for i in range(NF)])

    # state estimates
    sp, sr, sv = cfg.state_noise
    ba_est = ba_true + rng.normal(0, 0.005, 3); bg_est = bg_true + rng.normal(0, 0.0005, 3)
    para_pose = np.zeros((NF, 7)); para_sb = np.zeros((NF, 9))
    para_pose[:, :3] = Pw + rng.normal(0, sp, (NF, 3))
    dq = q_exp(rng.normal(0, sr, (NF, 3)))
    qe = q_mul(Qw, dq)
    para_pose[:, 3:] = qe / np.linalg.norm(qe, axis=-1, keepdims=True)
    para_sb[:, :3] = Vw + rng.normal(0, sv, (NF, 3))
    para_sb[:, 3:6] = ba_est + rng.normal(0, 0.001, (NF, 3))
    para_sb[:, 6:9] = bg_est + rng.normal(0, 0.0001, (NF, 3))

    # IMU pre-integration per interval j = 1..NF-1
    K = NF - 1
    acc_k = np.stack([acc_m[(j - 1) * S: j * S + 1] for j in range(1, NF)])
    gyr_k = np.stack([gyr_m[(j - 1) * S: j * S + 1] for j in range(1, NF)])
    lin_ba = ba_est + rng.normal(0, 0.003, (K, 3)); lin_bg = bg_est + rng.normal(0, 0.0003, (K, 3))
    imu = np.zeros((NF, IMU_DOUBLES))
    imu[1:] = preintegrate(acc_k, gyr_k, dt, lin_ba, lin_bg)
    imu[0, IMU_OFF["delta_q"][0] + 3] = 1.0

    # camera poses (truth)
    Rc = Rw @ RIC
    Pc = Pw + np.einsum('kij,j->ki', Rw, TIC)

    # features
    F = cfg.n_features
    starts, offs, pts, depth_true = [], [0], [], []
    n_try = 0
    while len(starts) < F and n_try < 50 * F:
        n_try += 1
        s0 = min(int(8 * rng.uniform() ** 2), NF - 4)                    # start_frame < WINDOW_SIZE-2 (estimator.cpp:755)
        u, v = rng.uniform(0, IMG_W), rng.uniform(0, IMG_H)
        d = rng.uniform(5.0, 50.0)
        ray = np.array([(u - CX) / FX, (v - CY) / FY, 1.0])
        Xw = Rc[s0] @ (d * ray) + Pc[s0]
        end = NF - 1
        if rng.uniform() < cfg.early_end_fraction:
            end = rng.integers(s0 + 1, NF)
        obs = []
        for j in range(s0, end + 1):
            pc = Rc[j].T @ (Xw - Pc[j])
            if pc[2] < 1.0 or abs(pc[0] / pc[2]) > 1.3 or abs(pc[1] / pc[2]) > 0.6:
                break
            obs.append([pc[0] / pc[2] + rng.normal(0, cfg.pixel_sigma), pc[1] / pc[2] + rng.normal(0, cfg.pixel_sigma), 1.0])
        if len(obs) < 2:
            continue
        starts.append(s0); pts.extend(obs); offs.append(len(pts)); depth_true.append(d)
    F = len(starts)
    depth_true = np.array(depth_true)
    feature_const = (rng.uniform(size=F) < cfg.const_fraction).astype(np.uint8)
    depth_est = np.where(feature_const == 1, depth_true + rng.normal(0, 0.05, F), depth_true * (1 + 0.1 * rng.normal(size=F)))
    depth_est = np.maximum(depth_est, 0.5)
    para_feature = 1.0 / depth_est

    # LiDAR between-constraints: true relative LiDAR pose (+ noise), lidar_factor.h:28-36
    lidar = np.zeros((NF, 7)); lidar[:, 3] = 1.0
    if cfg.use_lidar:
        Ril = RIC @ RCL; til = RIC @ TCL + TIC
        for j in range(1, NF):
            Rij = Rw[j - 1].T @ Rw[j]
            Pij = Rw[j - 1].T @ (Pw[j] - Pw[j - 1])
            Rl = Ril.T @ Rij @ Ril
            tl = Ril.T @ (Rij @ til + Pij - til)
            ql = q_mul(R_to_q(Rl), q_exp(rng.normal(0, np.deg2rad(0.1), 3)))
            lidar[j, :4] = ql / np.linalg.norm(ql)
            lidar[j, 4:] = tl + rng.normal(0, 0.02, 3)

    ex = np.concatenate([TIC, R_to_q(RIC)])
    win = Window(para_pose, para_sb, ex, para_feature, feature_const, np.array(starts, dtype=np.int32),
                 np.array(offs, dtype=np.int32), np.array(pts), imu, lidar=lidar,
                 marginalization_flag=cfg.marginalization_flag)

    prior = None
    if cfg.with_prior:
        prior = make_synthetic_prior(rng, win, NF)
    truth = dict(P=Pw, R=Rw, Q=Qw, V=Vw, ba=ba_true, bg=bg_true, depth=depth_true)
    return win, prior, truth


def make_synthetic_prior(rng, win, NF):
    """Dense prior over Pose[0..NF-2], SpeedBias[0], Ex_Pose (n = 6*(NF-1) + 9 + 6 = 75 for NF = 11): an SPD
    information matrix (gauge-fixing weights + random relative-pose style couplings), J0 = chol^T, small r0."""
    blocks = []
    idx = 0
    for i in range(NF - 1):
        x0 = win.para_pose[i].copy()
        x0[:3] += rng.normal(0, 0.02, 3)
        q = q_mul(x0[3:], q_exp(rng.normal(0, 0.003, 3))); x0[3:] = q / np.linalg.norm(q)
        blocks.append(dict(id=i, size=7, idx=idx, x0=x0)); idx += 6
    x0 = win.para_speed_bias[0].copy() + np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3), rng.normal(0, 0.0002, 3)])
    blocks.append(dict(id=NF + 0, size=9, idx=idx, x0=x0)); idx += 9
    blocks.append(dict(id=2 * NF, size=7, idx=idx, x0=win.para_ex_pose.copy())); idx += 6
    n = idx
    d = np.zeros(n)
    for i in range(NF - 1):
        wgt = 1e4 if i == 0 else 1e2
        d[6 * i:6 * i + 6] = wgt
    o = 6 * (NF - 1)
    d[o:o + 3] = 400.0; d[o + 3:o + 6] = 2500.0; d[o + 6:o + 9] = 2.5e5
    d[o + 9:o + 15] = 1e6
    Lam = np.diag(d)
    for _ in range(40):
        a = np.zeros(n)
        i = rng.integers(0, NF - 2)
        a[6 * i:6 * i + 6] = rng.normal(0, 1, 6)
        a[6 * (i + 1):6 * (i + 1) + 6] = -a[6 * i:6 * i + 6] + rng.normal(0, 0.1, 6)
        if rng.uniform() < 0.3:
            a[o:o + 9] = rng.normal(0, 1, 9) * np.array([1, 1, 1, 3, 3, 3, 30, 30, 30])
        Lam += rng.uniform(10, 300) * np.outer(a, a)
    J0 = np.linalg.cholesky(Lam).T
    r0 = rng.normal(0, 0.3, n)
    return make_prior(J0, r0, blocks, m=0)


def make_batch(seed, n_windows, opts, cfg=None, distinct=None):
    """`n_windows` windows from `distinct` (default: all) different seeds, tiled. Returns (windows, priors)."""
    distinct = n_windows if distinct is None else min(distinct, n_windows)
    base = [make_window(seed * 100003 + i, opts, cfg) for i in range(distinct)]
    wins = [base[i % distinct][0] for i in range(n_windows)]
    priors = [base[i % distinct][1] for i in range(n_windows)]
    return wins, priors


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic LiDAR scene for the scan-to-map path: ground plane, 4 walls and vertical poles in a box,
# a 64-ring / 1800-azimuth scan pattern ray-cast from the moving sensor. Pole hits are handed out as "edge" features, ground /
# wall hits as "surf" features (the LOAM curvature front-end that does this classification in the reference,
# featureExtraction.hpp, is out of scope: its outputs are the inputs of the hot path).
class LidarScene:
    def __init__(self, seed=0, half=100.0, n_poles=40, sensor_height=1.73, rings=64, azimuths=1800, max_range=90.0, min_range=3.0, noise=0.02):
        rng = np.random.default_rng(seed)
        self.rng = rng
        self.half, self.h = half, sensor_height
        self.poles = np.column_stack([rng.uniform(-half * 0.9, half * 0.9, n_poles), rng.uniform(-half * 0.9, half * 0.9, n_poles)])
        self.pole_r = rng.uniform(0.1, 0.25, n_poles)
        if rings == 64:      # HDL-64E: 1/3 deg spacing above -8.83 deg, 1/2 deg below — the model featureExtraction.hpp:92-101 inverts
            el = np.deg2rad(np.array([2.0 - r / 3.0 if r <= 32 else -8.83 - (r - 32) / 2.0 for r in range(64)]))
        else:
            el = np.deg2rad(np.linspace(-24.8, 2.0, rings))
        az = np.linspace(-np.pi, np.pi, azimuths, endpoint=False)
        E, A = np.meshgrid(el, az, indexing="ij")
        self.dirs = np.stack([np.cos(E) * np.cos(A), np.cos(E) * np.sin(A), np.sin(E)], -1).reshape(-1, 3)
        self.max_range, self.min_range, self.noise = max_range, min_range, noise

    def scan(self, R_wl, t_wl, edge_keep=0.5, surf_keep=0.06):
        """ray-cast one scan; returns (edge_xyzi, surf_xyzi) float32 in the LiDAR frame"""
        rng = self.rng
        d = self.dirs @ R_wl.T                     # world directions
        o = t_wl
        n = d.shape[0]
        best = np.full(n, np.inf); kind = np.zeros(n, dtype=np.int8)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (0.0 - o[2]) / d[:, 2]            # ground z = 0
            ok = (tg > 0) & (tg < best); best[ok] = tg[ok]; kind[ok] = 1
            for axis in (0, 1):
                for sgn in (-1.0, 1.0):
                    tw = (sgn * self.half - o[axis]) / d[:, axis]
                    hit = o + tw[:, None] * d
                    ok = (tw > 0) & (tw < best) & (hit[:, 2] > 0) & (hit[:, 2] < 12.0) & (np.abs(hit[:, 1 - axis]) <= self.half)
                    best[ok] = tw[ok]; kind[ok] = 1
            # vertical cylinders (poles), height 0..8 m
            dxy2 = d[:, 0] ** 2 + d[:, 1] ** 2
            for (px, py), r in zip(self.poles, self.pole_r):
                ox, oy = o[0] - px, o[1] - py
                bq = ox * d[:, 0] + oy * d[:, 1]
                cq = ox * ox + oy * oy - r * r
                disc = bq * bq - dxy2 * cq
                tc = (-bq - np.sqrt(np.where(disc > 0, disc, np.nan))) / dxy2
                z = o[2] + tc * d[:, 2]
                ok = (disc > 0) & (tc > 0) & (tc < best) & (z > 0) & (z < 8.0)
                best[ok] = tc[ok]; kind[ok] = 2
        ok = np.isfinite(best) & (best > self.min_range) & (best < self.max_range)
        rngs = best + rng.normal(0, self.noise, n)
        pts_l = self.dirs * rngs[:, None]          # LiDAR frame
        e_idx = np.where(ok & (kind == 2))[0]
        s_idx = np.where(ok & (kind == 1))[0]
        e_idx = e_idx[rng.uniform(size=e_idx.size) < edge_keep]
        s_idx = s_idx[rng.uniform(size=s_idx.size) < surf_keep]
        mk = lambda idx: np.ascontiguousarray(np.column_stack([pts_l[idx], np.ones(idx.size)]).astype(np.float32))
        return mk(e_idx), mk(s_idx)


    def scan_raw(self, R_wl, t_wl):
        """ray-cast one full scan; returns every valid return (xyzi float32, LiDAR frame) in firing order (ring-major, azimuth
        ascending) — the input of the LOAM feature extraction (featureExtraction.hpp)."""
        rng = self.rng
        d = self.dirs @ R_wl.T
        o = t_wl
        n = d.shape[0]
        best = np.full(n, np.inf)
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = (0.0 - o[2]) / d[:, 2]
            ok = (tg > 0) & (tg < best); best[ok] = tg[ok]
            for axis in (0, 1):
                for sgn in (-1.0, 1.0):
                    tw = (sgn * self.half - o[axis]) / d[:, axis]
                    hit = o + tw[:, None] * d
                    ok = (tw > 0) & (tw < best) & (hit[:, 2] > 0) & (hit[:, 2] < 12.0) & (np.abs(hit[:, 1 - axis]) <= self.half)
                    best[ok] = tw[ok]
            dxy2 = d[:, 0] ** 2 + d[:, 1] ** 2
            for (px, py), r in zip(self.poles, self.pole_r):
                ox, oy = o[0] - px, o[1] - py
                bq = ox * d[:, 0] + oy * d[:, 1]
                disc = bq * bq - dxy2 * (ox * ox + oy * oy - r * r)
                tc = (-bq - np.sqrt(np.where(disc > 0, disc, np.nan))) / dxy2
                z = o[2] + tc * d[:, 2]
                ok = (disc > 0) & (tc > 0) & (tc < best) & (z > 0) & (z < 8.0)
                best[ok] = tc[ok]
        ok = np.isfinite(best) & (best > self.min_range) & (best < self.max_range)
        pts = self.dirs * (best + rng.normal(0, self.noise, n))[:, None]
        idx = np.where(ok)[0]
        return np.ascontiguousarray(np.column_stack([pts[idx], np.ones(idx.size)]).astype(np.float32))

    def sample_map(self, R_wl, t_wl, ground_step=0.8, wall_step=0.8, pole_dz=0.25, pole_az=1, noise=0.02):
        """A dense local map of the scene as a long mapping run would have accumulated it (ground lattice, walls, pole surfaces),
        expressed in the LiDAR frame (R_wl, t_wl) — the frame scan-to-map calls "world" when that pose is its first one.
        Returns (edge_xyzi, surf_xyzi) float32."""
        rng = self.rng
        h = self.half
        g = np.arange(-h + 0.5 * ground_step, h, ground_step)
        gx, gy = np.meshgrid(g, g, indexing="ij")
        ground = np.column_stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)])
        wz = np.arange(0.4, 12.0, wall_step)
        wl = np.arange(-h + 0.5 * wall_step, h, wall_step)
        a, z = np.meshgrid(wl, wz, indexing="ij")
        walls = []
        for sgn in (-1.0, 1.0):
            walls.append(np.column_stack([np.full(a.size, sgn * h), a.ravel(), z.ravel()]))
            walls.append(np.column_stack([a.ravel(), np.full(a.size, sgn * h), z.ravel()]))
        surf = np.concatenate([ground] + walls)
        surf = surf + rng.normal(0, noise, surf.shape) + rng.uniform(-0.25, 0.25, surf.shape) * np.array([1.0, 1.0, 0.0]) * (np.arange(len(surf)) < len(ground))[:, None]
        pz = np.arange(0.2, 8.0, pole_dz)
        ang = np.linspace(0, 2 * np.pi, pole_az, endpoint=False)
        edge = []
        for (px, py), r in zip(self.poles, self.pole_r):
            for th in ang + rng.uniform(0, 2 * np.pi):
                edge.append(np.column_stack([np.full(pz.size, px + r * np.cos(th)), np.full(pz.size, py + r * np.sin(th)), pz]))
        edge = np.concatenate(edge) + rng.normal(0, noise, (len(edge) * pz.size, 3))
        to_l = lambda P: (P - t_wl) @ R_wl                      # R^T (p - t)
        mk = lambda P: np.ascontiguousarray(np.column_stack([to_l(P), np.ones(len(P))]).astype(np.float32))
        return mk(edge), mk(surf)


def make_lidar_bench_case(seed, n_poles=1500, edge_keep=0.11, surf_keep=0.25, speed=8.0, dt=0.1):
    """SURVEY.md §8(d) config 3 LiDAR stage. Returns (map_edge, map_surf, scans, pose_last_qt):
      * map_*: a dense raw local map in the frame of pose 0 (where scan-to-map starts with the identity pose) — localMapInited input;
      * scans[0]: the 64-ring scan one frame later — the WARM-UP step that turns the raw map into the steady-state voxelised local
        map (≈ 30 k edge + ≈ 60 k surf points, ascending leaf order) and gives the constant-velocity model its two poses;
      * scans[1]: the scan of the frame after that — the measured step (≈ 1.2 k edge + ≈ 2.8 k surf queries after the 0.4 / 0.8 m grids);
      * pose_last_qt: the pose one frame before pose 0 [qx qy qz qw tx ty tz] (globalOdom_last of the first prediction)."""
    scene = LidarScene(seed, n_poles=n_poles)
    pose = lambda k: (euler_R(np.array(0.05 * k * dt * 10), np.array(0.0), np.array(0.0)), np.array([-30.0 + speed * dt * k, 5.0 * np.sin(0.1 * k), scene.h]))
    (R0, t0), (Rm, tm) = pose(0), pose(-1)
    me, ms = scene.sample_map(R0, t0, ground_step=0.55, pole_dz=0.2)
    scans = [scene.scan(*pose(k), edge_keep=edge_keep, surf_keep=surf_keep) for k in (1, 2)]
    pose_last = np.concatenate([R_to_q(R0.T @ Rm), R0.T @ (tm - t0)])
    return me, ms, scans, pose_last


def make_lidar_sequence(seed, n_frames, speed=8.0, dt=0.1, yaw_rate=0.05, **kw):
    """returns (scans [(edge, surf)], poses [(R_wl, t_wl)]) for a vehicle driving through a LidarScene"""
    scene = LidarScene(seed, **kw)
    scans, poses = [], []
    for k in range(n_frames):
        yaw = yaw_rate * k * dt * 10
        R = euler_R(np.array(yaw), np.array(0.0), np.array(0.0))
        t = np.array([-30.0 + speed * dt * k, 5.0 * np.sin(0.1 * k), scene.h])
        scans.append(scene.scan(R, t))
        poses.append((R, t))
    return scans, poses


def with_td_inputs(win, seed, td_true=0.004):
    """ProjectionTdFactor inputs (projection_td_factor.cpp:6-21) for a synthetic window: pixel velocities on the normalised plane, zero per-observation td, image
    rows; the observations are shifted as a camera running `td_true` seconds late would have seen them."""
    from . import abi
    rng = np.random.default_rng(seed)
    vel = rng.normal(0.0, 0.4, (win.n_obs, 2))
    pts = win.obs_point.copy()
    pts[:, :2] += td_true * vel
    return abi.Window(win.para_pose, win.para_speed_bias, win.para_ex_pose, win.para_feature, win.feature_const, win.feature_start_frame,
                      win.feature_obs_offset, pts, win.imu, win.lidar, para_td=0.0, marginalization_flag=win.marginalization_flag,
                      obs_velocity=vel, obs_cur_td=np.zeros(win.n_obs), obs_row=rng.uniform(0.0, 370.0, win.n_obs))

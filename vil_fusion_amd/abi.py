"""ctypes mirror of include/vilfusion.h (POD structs only; no torch types).

The same structs are fed to the HIP library (libvilfusion_hip.so) and — in tests only — to the CPU oracle
(liboracle_vilf.so), so both see byte-identical inputs.
"""
import ctypes as C
import numpy as np

VILF_MAX_FRAMES = 11
VILF_MAX_FEATURES = 1000
VILF_PRIOR_MAX_DIM = 160
VILF_PRIOR_MAX_BLOCKS = 24

VILF_OK = 0
VILF_ERR_INVALID_ARGUMENT = -1
VILF_ERR_DEVICE = -2
VILF_ERR_UNSUPPORTED = -3
VILF_ERR_NO_GPU = -4

MARGIN_OLD = 0
MARGIN_SECOND_NEW = 1

TERM_NAMES = {0: "NO_CONVERGENCE", 1: "CONVERGENCE_FUNCTION", 2: "CONVERGENCE_PARAMETER", 3: "CONVERGENCE_GRADIENT", 4: "FAILURE"}

c_double_p = C.POINTER(C.c_double)


class Options(C.Structure):
    _fields_ = [
        ("window_size", C.c_int), ("max_num_iterations", C.c_int), ("max_solver_time", C.c_double),
        ("focal_length", C.c_double), ("cauchy_a", C.c_double), ("G", C.c_double * 3),
        ("estimate_extrinsic", C.c_int), ("estimate_td", C.c_int), ("use_lidar_const", C.c_int),
        ("RIC", C.c_double * 9), ("TIC", C.c_double * 3), ("RCL", C.c_double * 9), ("TCL", C.c_double * 3),
        ("TR", C.c_double), ("ROW", C.c_double), ("init_depth", C.c_double),
        ("edge_leaf_size", C.c_double), ("surf_leaf_size", C.c_double), ("huber_a", C.c_double),
        ("s2m_outer_iterations", C.c_int), ("s2m_max_iterations", C.c_int), ("s2m_crop_half", C.c_double),
    ]


class ImuPreint(C.Structure):
    _fields_ = [
        ("sum_dt", C.c_double), ("delta_p", C.c_double * 3), ("delta_q", C.c_double * 4), ("delta_v", C.c_double * 3),
        ("linearized_ba", C.c_double * 3), ("linearized_bg", C.c_double * 3),
        ("jacobian", C.c_double * 225), ("covariance", C.c_double * 225),
    ]


IMU_DOUBLES = 1 + 3 + 4 + 3 + 3 + 3 + 225 + 225  # 467
assert C.sizeof(ImuPreint) == 8 * IMU_DOUBLES
# numpy view of ImuPreint: a row of 467 doubles
IMU_OFF = {"sum_dt": (0, 1), "delta_p": (1, 4), "delta_q": (4, 8), "delta_v": (8, 11), "linearized_ba": (11, 14),
           "linearized_bg": (14, 17), "jacobian": (17, 242), "covariance": (242, 467)}


class LidarConstraint(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3)]


class WindowIn(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int), ("para_pose", c_double_p), ("para_speed_bias", c_double_p),
        ("para_ex_pose", C.c_double * 7), ("para_td", C.c_double),
        ("n_features", C.c_int), ("para_feature", c_double_p), ("feature_const", C.POINTER(C.c_uint8)),
        ("feature_start_frame", C.POINTER(C.c_int32)), ("feature_obs_offset", C.POINTER(C.c_int32)),
        ("n_obs", C.c_int), ("obs_point", c_double_p), ("obs_velocity", c_double_p), ("obs_cur_td", c_double_p),
        ("obs_row", c_double_p), ("imu", C.POINTER(ImuPreint)), ("lidar", C.POINTER(LidarConstraint)),
        ("marginalization_flag", C.c_int), ("gauge_R0", c_double_p), ("gauge_P0", c_double_p),
    ]


class Summary(C.Structure):
    _fields_ = [
        ("num_iterations", C.c_int), ("num_successful_steps", C.c_int), ("num_linear_solves", C.c_int),
        ("termination", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double),
        ("final_radius", C.c_double), ("usec_solve", C.c_double),
    ]


class WindowOut(C.Structure):
    _fields_ = [
        ("para_pose", c_double_p), ("para_speed_bias", c_double_p), ("para_feature", c_double_p),
        ("Ps", c_double_p), ("Rs", c_double_p), ("Vs", c_double_p), ("Bas", c_double_p), ("Bgs", c_double_p),
        ("tic", C.c_double * 3), ("ric", C.c_double * 9), ("td", C.c_double), ("summary", Summary),
    ]


class Prior(C.Structure):
    _fields_ = [
        ("valid", C.c_int), ("n", C.c_int), ("m", C.c_int), ("n_blocks", C.c_int),
        ("block_id", C.c_int * VILF_PRIOR_MAX_BLOCKS), ("block_size", C.c_int * VILF_PRIOR_MAX_BLOCKS),
        ("block_idx", C.c_int * VILF_PRIOR_MAX_BLOCKS), ("block_x0", (C.c_double * 9) * VILF_PRIOR_MAX_BLOCKS),
        ("linearized_residuals", C.c_double * VILF_PRIOR_MAX_DIM),
        ("linearized_jacobians", C.c_double * (VILF_PRIOR_MAX_DIM * VILF_PRIOR_MAX_DIM)),
    ]


class PgEdge(C.Structure):
    _fields_ = [("i", C.c_int), ("j", C.c_int), ("q", C.c_double * 4), ("t", C.c_double * 3), ("sigma", C.c_double * 6), ("robust", C.c_int), ("pad_", C.c_int)]


class ImuNoise(C.Structure):
    _fields_ = [("acc_n", C.c_double), ("gyr_n", C.c_double), ("acc_w", C.c_double), ("gyr_w", C.c_double)]


class Scan2MapResult(C.Structure):
    _fields_ = [
        ("pose_qt", C.c_double * 7), ("rel_q", C.c_double * 4), ("rel_t", C.c_double * 3),
        ("n_edge_ds", C.c_int), ("n_surf_ds", C.c_int), ("n_edge_factors", C.c_int * 2), ("n_surf_factors", C.c_int * 2),
        ("iterations", C.c_int * 2), ("final_cost", C.c_double * 2), ("map_edge_size", C.c_int), ("map_surf_size", C.c_int),
    ]


def dptr(a):
    """pointer to a C-contiguous float64 numpy array (None -> NULL)."""
    if a is None:
        return c_double_p()
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(c_double_p)


class Window:
    """One sliding-window snapshot ≙ the Estimator members optimization() reads (estimator.h:70-146).

    All arrays are numpy, C-contiguous, in the reference's layouts (see include/vilfusion.h).
    """

    def __init__(self, para_pose, para_speed_bias, para_ex_pose, para_feature, feature_const, feature_start_frame,
                 feature_obs_offset, obs_point, imu, lidar=None, para_td=0.0, marginalization_flag=MARGIN_OLD,
                 obs_velocity=None, obs_cur_td=None, obs_row=None, gauge_R0=None, gauge_P0=None):
        f64 = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float64)
        self.para_pose = f64(para_pose)
        self.para_speed_bias = f64(para_speed_bias)
        self.para_ex_pose = f64(para_ex_pose)
        self.para_td = float(para_td)
        self.para_feature = f64(para_feature)
        self.feature_const = np.ascontiguousarray(feature_const, dtype=np.uint8)
        self.feature_start_frame = np.ascontiguousarray(feature_start_frame, dtype=np.int32)
        self.feature_obs_offset = np.ascontiguousarray(feature_obs_offset, dtype=np.int32)
        self.obs_point = f64(obs_point)
        self.obs_velocity = f64(obs_velocity)
        self.obs_cur_td = f64(obs_cur_td)
        self.obs_row = f64(obs_row)
        self.imu = f64(imu)          # (n_frames, 467) rows ≙ vilf_imu_preint; row 0 unused
        self.lidar = f64(lidar)      # (n_frames, 7) [qx qy qz qw tx ty tz]; row 0 unused
        self.marginalization_flag = int(marginalization_flag)
        self.gauge_R0 = f64(gauge_R0)
        self.gauge_P0 = f64(gauge_P0)
        self.n_frames = self.para_pose.shape[0]
        assert self.para_pose.shape == (self.n_frames, 7)
        assert self.para_speed_bias.shape == (self.n_frames, 9)
        assert self.imu.shape == (self.n_frames, IMU_DOUBLES)
        assert self.feature_obs_offset.shape[0] == self.n_features + 1
        assert self.obs_point.shape == (self.n_obs, 3)

    @property
    def n_features(self):
        return int(self.para_feature.shape[0])

    @property
    def n_obs(self):
        return int(self.obs_point.shape[0])

    @property
    def n_factors(self):
        return self.n_obs - self.n_features

    def as_struct(self):
        """vilf_window_in pointing into this object's arrays (keep `self` alive while the struct is used)."""
        w = WindowIn()
        w.n_frames = self.n_frames
        w.para_pose = dptr(self.para_pose)
        w.para_speed_bias = dptr(self.para_speed_bias)
        for i in range(7):
            w.para_ex_pose[i] = self.para_ex_pose[i]
        w.para_td = self.para_td
        w.n_features = self.n_features
        w.para_feature = dptr(self.para_feature)
        w.feature_const = self.feature_const.ctypes.data_as(C.POINTER(C.c_uint8))
        w.feature_start_frame = self.feature_start_frame.ctypes.data_as(C.POINTER(C.c_int32))
        w.feature_obs_offset = self.feature_obs_offset.ctypes.data_as(C.POINTER(C.c_int32))
        w.n_obs = self.n_obs
        w.obs_point = dptr(self.obs_point)
        w.obs_velocity = dptr(self.obs_velocity)
        w.obs_cur_td = dptr(self.obs_cur_td)
        w.obs_row = dptr(self.obs_row)
        w.imu = C.cast(self.imu.ctypes.data, C.POINTER(ImuPreint))
        w.lidar = C.cast(self.lidar.ctypes.data, C.POINTER(LidarConstraint)) if self.lidar is not None else C.POINTER(LidarConstraint)()
        w.marginalization_flag = self.marginalization_flag
        w.gauge_R0 = dptr(self.gauge_R0)
        w.gauge_P0 = dptr(self.gauge_P0)
        return w


class WindowResult:
    """Caller-owned output buffers for vilf_window_out."""

    def __init__(self, n_frames, n_features):
        self.para_pose = np.zeros((n_frames, 7))
        self.para_speed_bias = np.zeros((n_frames, 9))
        self.para_feature = np.zeros(max(n_features, 1))[:n_features]
        self.Ps = np.zeros((n_frames, 3))
        self.Rs = np.zeros((n_frames, 3, 3))
        self.Vs = np.zeros((n_frames, 3))
        self.Bas = np.zeros((n_frames, 3))
        self.Bgs = np.zeros((n_frames, 3))
        self._feat_buf = np.zeros(max(n_features, 1))
        self.struct = WindowOut()
        s = self.struct
        s.para_pose = dptr(self.para_pose)
        s.para_speed_bias = dptr(self.para_speed_bias)
        s.para_feature = dptr(self._feat_buf)
        s.Ps, s.Rs, s.Vs, s.Bas, s.Bgs = dptr(self.Ps), dptr(self.Rs), dptr(self.Vs), dptr(self.Bas), dptr(self.Bgs)
        self._nfeat = n_features

    def finish(self):
        self.para_feature = self._feat_buf[:self._nfeat].copy()
        self.tic = np.array(self.struct.tic[:])
        self.ric = np.array(self.struct.ric[:]).reshape(3, 3)
        self.td = self.struct.td
        s = self.struct.summary
        self.summary = dict(num_iterations=s.num_iterations, num_successful_steps=s.num_successful_steps,
                            num_linear_solves=s.num_linear_solves, termination=s.termination,
                            initial_cost=s.initial_cost, final_cost=s.final_cost, final_radius=s.final_radius,
                            usec_solve=s.usec_solve)
        return self


def prior_to_numpy(p):
    """(J0 [n,n], r0 [n], blocks list) from a Prior struct."""
    n = p.n
    J = np.frombuffer(p.linearized_jacobians, dtype=np.float64, count=n * n).reshape(n, n).copy()
    r = np.frombuffer(p.linearized_residuals, dtype=np.float64, count=n).copy()
    blocks = [dict(id=p.block_id[i], size=p.block_size[i], idx=p.block_idx[i], x0=np.array(p.block_x0[i][:p.block_size[i]]))
              for i in range(p.n_blocks)]
    return J, r, blocks


def make_prior(J0, r0, blocks, m=0):
    """Prior struct from numpy parts; blocks = list of dict(id,size,idx,x0)."""
    p = Prior()
    n = J0.shape[0]
    assert J0.shape == (n, n) and n <= VILF_PRIOR_MAX_DIM and len(blocks) <= VILF_PRIOR_MAX_BLOCKS
    p.valid, p.n, p.m, p.n_blocks = 1, n, m, len(blocks)
    flat = np.ascontiguousarray(J0, dtype=np.float64).ravel()
    C.memmove(p.linearized_jacobians, flat.ctypes.data, 8 * n * n)
    rr = np.ascontiguousarray(r0, dtype=np.float64)
    C.memmove(p.linearized_residuals, rr.ctypes.data, 8 * n)
    for i, b in enumerate(blocks):
        p.block_id[i], p.block_size[i], p.block_idx[i] = int(b["id"]), int(b["size"]), int(b["idx"])
        for k in range(b["size"]):
            p.block_x0[i][k] = float(b["x0"][k])
    return p

"""Host mirror of the global_fusion pose-graph node (src/global_fusion/poseGraphOptimization.cpp) — SURVEY.md §8(f) N2.

What the reference's threads do around gtsam, restated as plain host code over the device solve (vilf_posegraph_optimize):
  * poseGraphOptimiza() :446-596   odometry in -> key-frame gate (2 m / 10 deg accumulated) -> prior / odometry BetweenFactor
  * icpCalculation()    :376-443   an accepted ICP result -> robust loop BetweenFactor(prev, curr)
  * isamUpdate()        :349-374   isam->update(); calculateEstimate(); updatePoses()
  * saveTUMTrajOdometry :88-110    `t x y z qx qy qz qw`, precision 9 / 5
ScanContext loop detection and the ICP itself are not part of this row: loop candidates come in as (prev, curr, ICP transform).
"""
import numpy as np

from . import synth

KEYFRAME_TRANS_TH = 2.0                      # /keyframe_trans_th (:647)
KEYFRAME_ROT_TH = np.deg2rad(10.0)           # /keyframe_rotate_th (:648-649)
PRIOR_SIGMA = np.sqrt(np.full(6, 1e-12))     # initNoises() :121-139, gtsam order [rot, trans]
ODOM_SIGMA = np.sqrt(np.array([1e-6, 1e-6, 1e-6, 1e-4, 1e-4, 1e-4]))
LOOP_SIGMA = np.sqrt(np.full(6, 0.5))


def inverse(p):
    qi = synth.q_conj(p[:4])
    return np.concatenate([qi, -(synth.q_to_R(qi) @ p[4:])])


def compose(a, b):
    q = synth.q_mul(a[:4], b[:4])
    return np.concatenate([q / np.linalg.norm(q), synth.q_to_R(a[:4]) @ b[4:] + a[4:]])


def between(a, b):
    """gtsam Pose3::between: a^-1 * b   (poses are [qx qy qz qw tx ty tz])"""
    return compose(inverse(a), b)


def max_rotation_difference(a, b):
    """largest relative rotation angle (rad) between two pose arrays"""
    worst = 0.0
    for pa, pb in zip(a, b):
        d = synth.q_mul(synth.q_conj(pa[:4]), pb[:4])
        worst = max(worst, 2.0 * np.arctan2(np.linalg.norm(d[:3]), abs(d[3])))
    return worst


def rpy_from_q(q):
    """tf::Matrix3x3(q).getRPY (getOdom, :140-151)"""
    R = synth.q_to_R(q)
    return np.array([np.arctan2(R[2, 1], R[2, 2]), np.arcsin(np.clip(-R[2, 0], -1.0, 1.0)), np.arctan2(R[1, 0], R[0, 0])])


def q_from_rpy(rpy):
    """gtsam::Rot3::RzRyRx(roll, pitch, yaw) / tf::createQuaternionFromRPY"""
    return synth.R_to_q(synth.euler_R(np.array(rpy[2]), np.array(rpy[1]), np.array(rpy[0])))


def diff_transformation(p1, p2):
    """diffTransformation (:153-164): |translation| and |euler angles| of p1^-1 p2, in float like pcl::getTransformation. p = (x y z roll pitch yaw)"""
    def T(p):
        M = np.eye(4, dtype=np.float32)
        M[:3, :3] = synth.euler_R(np.array(p[5]), np.array(p[4]), np.array(p[3])).astype(np.float32)
        M[:3, 3] = np.asarray(p[:3], dtype=np.float32)
        return M
    D = (np.linalg.inv(T(p1)) @ T(p2)).astype(np.float32)
    roll, pitch, yaw = np.arctan2(D[2, 1], D[2, 2]), np.arcsin(np.clip(-D[2, 0], -1, 1)), np.arctan2(D[1, 0], D[0, 0])
    return np.abs(np.array([D[0, 3], D[1, 3], D[2, 3], roll, pitch, yaw], dtype=np.float64))


class PoseGraph:
    """`backend(poses[n,7], prior_sigma, edges) -> poses[n,7]` runs the solve (HIP: estimator.posegraph_optimize; tests: the oracle)."""

    def __init__(self, backend):
        self.backend = backend
        self.translate_acc, self.rotation_acc = 1000000.0, 100000.0          # :50-51
        self.prev = np.zeros(6); self.curr = np.zeros(6)
        self.nodes = []               # key frames: dict(stamp, pose6d, updated6d)
        self.edges = []               # (i, j, q, t, sigma, robust)

    def add_odometry(self, stamp, pose_qt):
        """one synchronised odometry message (:483-587). Returns True if it became a key frame."""
        pose_qt = np.asarray(pose_qt, dtype=np.float64)
        cur = np.concatenate([pose_qt[4:], rpy_from_q(pose_qt[:4])])
        self.prev, self.curr = self.curr, cur
        rel = diff_transformation(self.prev, self.curr)
        self.translate_acc += float(np.float32(np.sqrt(rel[0] ** 2 + rel[1] ** 2 + rel[2] ** 2)))     # poseDistance() returns float (common.h:64)
        self.rotation_acc += rel[3] + rel[4] + rel[5]
        if not (self.translate_acc > KEYFRAME_TRANS_TH or self.rotation_acc > KEYFRAME_ROT_TH):
            return False
        self.translate_acc = self.rotation_acc = 0.0
        self.nodes.append(dict(stamp=float(stamp), pose=cur.copy(), updated=cur.copy()))
        k = len(self.nodes) - 1
        if k > 0:                                                           # BetweenFactor(prev, curr, poseFrom.between(poseTo), odomNoise) :577-584
            rel_qt = between(self._qt(self.nodes[k - 1]["pose"]), self._qt(cur))
            self.edges.append((k - 1, k, rel_qt[:4], rel_qt[4:], ODOM_SIGMA, 0))
        return True

    def add_loop(self, prev, curr, icp_qt):
        """an accepted ICP alignment of key frame `curr` onto `prev` (:420-436): measured = poseFrom.between(identity) = poseFrom^-1"""
        m = inverse(np.asarray(icp_qt, dtype=np.float64))
        self.edges.append((int(prev), int(curr), m[:4], m[4:], LOOP_SIGMA, 1))

    @staticmethod
    def _qt(p6):
        return np.concatenate([q_from_rpy(p6[3:]), p6[:3]])

    def update(self):
        """isamUpdate + updatePoses (:217-237, :349-374): the current estimate of every key frame"""
        x0 = np.array([self._qt(n["updated"]) for n in self.nodes])
        x = self.backend(x0, PRIOR_SIGMA, self.edges)
        for n, p in zip(self.nodes, x):
            n["updated"] = np.concatenate([p[4:], rpy_from_q(p[:4])])
        return x

    def save_tum(self, path):
        with open(path, "w") as fh:
            for n in self.nodes:
                p = n["updated"]; q = q_from_rpy(p[3:])
                fh.write("%.9f %.5f %.5f %.5f %.5f %.5f %.5f %.5f\n" % (n["stamp"], p[0], p[1], p[2], q[0], q[1], q[2], q[3]))


def make_synthetic_graph(seed, K, loops=(), odom_noise=(0.002, 0.02), loop_noise=(0.0005, 0.005), step=2.0):
    """K key frames along a closed, gently climbing loop (2 m apart, like the reference's key-frame spacing): truth poses, the dead-reckoned
    initial estimate built from noisy odometry edges (rotation / translation sigma), and loop edges between the given key-frame pairs."""
    rng = np.random.default_rng(seed)
    radius = max(K * step / (2 * np.pi), 1.0)
    truth = []
    for k in range(K):
        a = 2 * np.pi * k / max(K, 1)
        R = synth.euler_R(np.array(a + np.pi / 2), np.array(0.05 * np.sin(3 * a)), np.array(0.03 * np.cos(2 * a)))
        truth.append(np.concatenate([synth.R_to_q(R), [radius * np.cos(a), radius * np.sin(a), 2.0 * np.sin(a)]]))
    truth = np.array(truth)
    noisy = lambda rel, s: compose(rel, np.concatenate([synth.q_exp(rng.normal(0, s[0], 3)), rng.normal(0, s[1], 3)]))
    edges, x0 = [], [truth[0].copy()]
    for k in range(1, K):
        m = noisy(between(truth[k - 1], truth[k]), odom_noise)
        edges.append((k - 1, k, m[:4], m[4:], ODOM_SIGMA, 0))
        x0.append(compose(x0[-1], m))
    for (i, j) in loops:
        m = noisy(between(truth[i], truth[j]), loop_noise)
        edges.append((i, j, m[:4], m[4:], LOOP_SIGMA, 1))
    return truth, np.array(x0), edges

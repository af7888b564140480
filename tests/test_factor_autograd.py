"""Pins the oracle's analytic Jacobians against AUTOMATIC differentiation of an independent restatement of every residual (torch, fp64): tighter than the
forward differences of tests/test_oracle_factors.py (1e-10 instead of ~1e-6) and independent of the oracle's code — the residual formulas below are written from the
reference's sources (cited per factor), the derivative is taken along the SAME local perturbation the solver uses (pose: p += dp, q = (q ⊗ deltaQ(dθ)).normalized(),
pose_local_parameterization.cpp:3-19; F-LOAM pose: left SE(3) perturbation, EstimationMapping.hpp:34-49). The reference's own recipe for this check is
ProjectionFactor::check() (projection_factor.cpp:176-224, forward differences, eps 1e-6). CPU only.

Where the reference's analytic Jacobian is itself an approximation the test says so and pins the exact part:
  * IMUFactor: d r_q / d bg_i uses delta_q, not corrected_delta_q (imu_factor.h:124-126) — exact only at Bg_i = linearized_bg;
  * MarginalizationFactor: the Jacobian is J0 itself (marginalization_factor.cpp:364-376), i.e. d(2 vec(q0^-1 q)) / dθ = I — exact only at q = q0;
  * lidarFactor: the Jacobians are those of the UNWEIGHTED residual (lidar_factor.h:39-42 vs :44-75).
"""
import ctypes as C
import numpy as np
import pytest
import torch
from vil_fusion_amd import abi, synth

torch.set_default_dtype(torch.float64)
T = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64))


# ---- quaternions as [x y z w] tensors ------------------------------------------------------------------------------
def qmul(a, b):
    ax, ay, az, aw = a[0], a[1], a[2], a[3]
    bx, by, bz, bw = b[0], b[1], b[2], b[3]
    return torch.stack([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])


def qinv(q):
    return torch.stack([-q[0], -q[1], -q[2], q[3]]) / (q @ q)


def qrot(q, v):
    return qmul(qmul(q, torch.cat([v, torch.zeros(1)])), qinv(q))[:3]


def skew(v):
    z = torch.zeros(())
    return torch.stack([torch.stack([z, -v[2], v[1]]), torch.stack([v[2], z, -v[0]]), torch.stack([-v[1], v[0], z])])


def pose_plus(x, d):
    """PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19) with Utility::deltaQ = (1, θ/2) (utility.h:16-29)"""
    q = qmul(x[3:], torch.cat([d[3:] / 2, torch.ones(1)]))
    return torch.cat([x[:3] + d[:3], q / torch.linalg.norm(q)])


def se3_plus(x, d):
    """LocalSE3Parameterization::Plus (EstimationMapping.hpp:34-49): q+ = exp(ω) ⊗ q, t+ = exp(ω) t + J(ω) υ. Second-order series of exp and first-order of J: exact first
    derivative at d = 0, and no 0/0 for autograd at the origin."""
    w, u = d[:3], d[3:]
    K = skew(w)
    R = torch.eye(3) + K + 0.5 * K @ K
    dq = torch.cat([w / 2, torch.ones(1)]); dq = dq / torch.linalg.norm(dq)
    return torch.cat([qmul(dq, x[:4]), R @ x[4:] + (torch.eye(3) + 0.5 * K) @ u])


def jac_blocks(f, sizes):
    """autograd Jacobian of f(list of local perturbations) at zero, one block per parameter"""
    z = torch.zeros(sum(sizes))
    J = torch.autograd.functional.jacobian(lambda d: f(list(torch.split(d, sizes))), z)
    return [b.numpy() for b in torch.split(J, sizes, dim=1)]


def close(a, b, tol=1e-10):
    return np.abs(a - b).max() <= tol * max(1.0, np.abs(b).max())


def rand_pose(rng, scale=1.0):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(0, scale, 3), q])


# ---- ProjectionFactor / ProjectionTdFactor -----------------------------------------------------------------------------------
def projection_residual(Pi, Pj, ex, lam, pts_i, pts_j, sqrt_info):
    """projection_factor.cpp:36-54"""
    pc_i = pts_i / lam
    p_imu_i = qrot(ex[3:], pc_i) + ex[:3]
    p_w = qrot(Pi[3:], p_imu_i) + Pi[:3]
    p_imu_j = qrot(qinv(Pj[3:]), p_w - Pj[:3])
    pc_j = qrot(qinv(ex[3:]), p_imu_j - ex[:3])
    return sqrt_info * (pc_j[:2] / pc_j[2] - pts_j[:2])


def _proj_case(rng, opts):
    Pi = rand_pose(rng); Pj = rand_pose(rng); Pj[:3] = Pi[:3] + rng.normal(0, 0.5, 3)
    Pj[3:] = synth.q_mul(Pi[3:], synth.q_exp(rng.normal(0, 0.1, 3)))
    ex = np.concatenate([np.array(opts.TIC[:]), synth.R_to_q(np.array(opts.RIC[:]).reshape(3, 3))])
    lam = np.array([1.0 / rng.uniform(4, 30)])
    pi = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.2), 1.0]); pj = np.array([rng.uniform(-0.5, 0.5), rng.uniform(-0.2, 0.2), 1.0])
    return [Pi, Pj, ex, lam], pi, pj


def test_projection_factor_jacobians_by_autograd(oracle, opts):
    rng = np.random.default_rng(100)
    si = float(opts.focal_length) / 1.5
    for _ in range(10):
        params, pi, pj = _proj_case(rng, opts)
        r, J = oracle.eval_factor("projection", opts, params, pi, pj, sizes=[7, 7, 7, 1], nres=2)
        f = lambda d: projection_residual(pose_plus(T(params[0]), d[0]), pose_plus(T(params[1]), d[1]), pose_plus(T(params[2]), d[2]), T(params[3]) + d[3], T(pi), T(pj), si)
        assert close(f([torch.zeros(6)] * 3 + [torch.zeros(1)]).numpy(), r, 1e-12)
        for a, b, loc in zip(J, jac_blocks(f, [6, 6, 6, 1]), [6, 6, 6, 1]):
            assert close(a[:, :loc], b), (a[:, :loc], b)


def test_projection_td_factor_jacobians_by_autograd(oracle):
    import test_oracle_factors as tof
    rng = np.random.default_rng(101)
    o = oracle.default_options(); o.TR = 0.02
    si = float(o.focal_length) / 1.5
    for _ in range(10):
        params, pi, pj, vi, vj, tdi, tdj, ri, rj = tof._td_case(rng, o)
        r, J = oracle.eval_factor("projection_td", o, params, pi, pj, vi, vj, tdi, tdj, ri, rj, sizes=[7, 7, 7, 1, 1], nres=2)
        vi3, vj3 = T(np.append(vi, 0.0)), T(np.append(vj, 0.0))

        def f(d):          # projection_td_factor.cpp:52-67 (row_* - ROW / 2: constructor :18-19)
            td = T(params[4]) + d[4]
            pit = T(pi) - (td - tdi + o.TR / o.ROW * (ri - o.ROW / 2)) * vi3
            pjt = T(pj) - (td - tdj + o.TR / o.ROW * (rj - o.ROW / 2)) * vj3
            return projection_residual(pose_plus(T(params[0]), d[0]), pose_plus(T(params[1]), d[1]), pose_plus(T(params[2]), d[2]), T(params[3]) + d[3], pit, pjt, si)
        assert close(f([torch.zeros(6)] * 3 + [torch.zeros(1)] * 2).numpy(), r, 1e-12)
        for a, b, loc in zip(J, jac_blocks(f, [6, 6, 6, 1, 1]), [6, 6, 6, 1, 1]):
            assert close(a[:, :loc], b), (a[:, :loc], b)


# ---- IMUFactor --------------------------------------------------------------------------------------------------------------
def imu_residual(Pi, SBi, Pj, SBj, pre, G):
    """IntegrationBase::evaluate (integration_base.h:160-186), before sqrt_info"""
    Jm = T(np.array(pre.jacobian[:]).reshape(15, 15))
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = Jm[0:3, 9:12], Jm[0:3, 12:15], Jm[3:6, 12:15], Jm[6:9, 9:12], Jm[6:9, 12:15]
    dt = float(pre.sum_dt)
    dba = SBi[3:6] - T(pre.linearized_ba[:]); dbg = SBi[6:9] - T(pre.linearized_bg[:])
    cdq = qmul(T(pre.delta_q[:]), torch.cat([dq_dbg @ dbg / 2, torch.ones(1)]))
    cdv = T(pre.delta_v[:]) + dv_dba @ dba + dv_dbg @ dbg
    cdp = T(pre.delta_p[:]) + dp_dba @ dba + dp_dbg @ dbg
    Qi_inv = qinv(Pi[3:])
    rp = qrot(Qi_inv, 0.5 * G * dt * dt + Pj[:3] - Pi[:3] - SBi[:3] * dt) - cdp
    rq = 2 * qmul(qinv(cdq), qmul(Qi_inv, Pj[3:]))[:3]
    rv = qrot(Qi_inv, G * dt + SBj[:3] - SBi[:3]) - cdv
    return torch.cat([rp, rq, rv, SBj[3:6] - SBi[3:6], SBj[6:9] - SBi[6:9]])


def test_imu_factor_jacobians_by_autograd(oracle, opts):
    import test_oracle_factors as tof
    rng = np.random.default_rng(102)
    G = T(opts.G[:])
    for trial in range(6):
        win, j, pre, params = tof._imu_setup(rng, opts)
        params = [p.copy() for p in params]
        at_lin = trial % 2 == 0
        if at_lin:
            params[1][6:9] = np.array(pre.linearized_bg[:])          # dbg = 0: every block of the reference's Jacobian is exact
        f = lambda d: imu_residual(pose_plus(T(params[0]), d[0]), T(params[1]) + d[1], pose_plus(T(params[2]), d[2]), T(params[3]) + d[3], pre, G)
        r, J = oracle.eval_factor("imu_raw", opts, params, pre, sizes=[7, 9, 7, 9], nres=15)
        assert close(f([torch.zeros(6), torch.zeros(9), torch.zeros(6), torch.zeros(9)]).numpy(), r, 1e-12)
        Ja = jac_blocks(f, [6, 9, 6, 9])
        # away from the linearisation point corrected_delta_q = delta_q ⊗ (1, θ/2), θ = dq_dbg dbg, is not a unit quaternion: the residual uses its inverse
        # (integration_base.h:178), the Jacobians its conjugate (Qright(corrected_delta_q), imu_factor.h:97,150) — an O(θ²) difference the reference carries
        th = np.array(pre.jacobian[:]).reshape(15, 15)[3:6, 12:15] @ (params[1][6:9] - np.array(pre.linearized_bg[:]))
        tol = 1e-10 + 2.0 * float(th @ th)
        for k, (a, b, loc) in enumerate(zip(J, Ja, [6, 9, 6, 9])):
            a = a[:, :loc].copy(); b = b.copy()
            if k == 1 and not at_lin:
                # imu_factor.h:124-126: the block d r_q / d bg_i is built with delta_q instead of corrected_delta_q — first order in dbg away from the exact derivative
                assert np.abs(a[3:6, 6:9] - b[3:6, 6:9]).max() < 5.0 * np.linalg.norm(params[1][6:9] - np.array(pre.linearized_bg[:])) + 1e-12
                a[3:6, 6:9] = 0; b[3:6, 6:9] = 0
            assert close(a, b, tol), (trial, k, np.abs(a - b).max(), tol)
        # whitened: sqrt_info (the oracle's) times the raw parts
        S = np.zeros(225); oracle.lib().vilo_imu_sqrt_info(C.byref(pre), abi.dptr(S)); S = S.reshape(15, 15)
        rw, Jw = oracle.eval_factor("imu", opts, params, pre, sizes=[7, 9, 7, 9], nres=15)
        assert np.abs(rw - S @ r).max() <= 1e-12 * max(1.0, np.abs(rw).max())
        for a, b in zip(Jw, J):
            assert np.abs(a - S @ b).max() <= 1e-12 * max(1.0, np.abs(a).max())


# ---- lidarFactor (between-factor of the sliding window) -------------------------------------------------------------------------
def test_lidar_between_factor_jacobians_by_autograd(oracle, opts):
    rng = np.random.default_rng(103)
    RIC = np.array(opts.RIC[:]).reshape(3, 3); RCL = np.array(opts.RCL[:]).reshape(3, 3)
    qil = T(synth.R_to_q(RIC @ RCL)); til = T(RIC @ np.array(opts.TCL[:]) + np.array(opts.TIC[:]))
    qli = qinv(qil); tli = -qrot(qli, til)
    for _ in range(10):
        Pi = rand_pose(rng); Pj = rand_pose(rng)
        Pj[3:] = synth.q_mul(Pi[3:], synth.q_exp(rng.normal(0, 0.05, 3))); Pj[:3] = Pi[:3] + rng.normal(0, 1, 3)
        c = abi.LidarConstraint()
        lq = synth.q_exp(rng.normal(0, 0.05, 3)); lt = rng.normal(0, 1, 3)
        for k in range(4):
            c.q[k] = lq[k]
        for k in range(3):
            c.t[k] = lt[k]

        def f(d):          # lidar_factor.h:34-35, unweighted
            A, B = pose_plus(T(Pi), d[0]), pose_plus(T(Pj), d[1])
            Qi_inv = qinv(A[3:])
            corr = qmul(qmul(qil, T(lq)), qli)
            rp = qrot(qli, qrot(Qi_inv, B[:3] - A[:3]) - til - qrot(qmul(qil, T(lq)), tli)) - T(lt)
            rq = 2 * qmul(qinv(corr), qmul(Qi_inv, B[3:]))[:3]
            return torch.cat([rp, rq])
        r, J = oracle.eval_factor("lidar_between", opts, [Pi, Pj], c, sizes=[7, 7], nres=6)
        w = np.array([10, 10, 10, 100, 100, 100.0])
        assert close(w * f([torch.zeros(6), torch.zeros(6)]).numpy(), r, 1e-11), "the RESIDUAL is weighted (lidar_factor.h:37-42)"
        for a, b in zip(J, jac_blocks(f, [6, 6])):
            assert close(a[:, :6], b), "the Jacobians are those of the unweighted residual (:44-75)"


# ---- F-LOAM edge / plane factors ---------------------------------------------------------------------------------------------
def test_edge_and_surf_factor_jacobians_by_autograd(oracle):
    rng = np.random.default_rng(104)
    L = oracle.lib()
    L.vilo_eval_surf.argtypes = [abi.c_double_p, abi.c_double_p, abi.c_double_p, C.c_double, abi.c_double_p, abi.c_double_p]
    for _ in range(10):
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        pose = np.concatenate([q, rng.normal(0, 2, 3)])
        cp = rng.normal(0, 5, 3); a = rng.normal(0, 5, 3); b = a + rng.normal(0, 0.2, 3)
        r = np.zeros(3); J = np.zeros((3, 7))
        L.vilo_eval_edge(abi.dptr(pose), abi.dptr(cp), abi.dptr(a), abi.dptr(b), abi.dptr(r), abi.dptr(J))

        def fe(d):         # lidarFactor.hpp:23-32
            x = se3_plus(T(pose), d[0])
            lp = qrot(x[:4], T(cp)) + x[4:]
            return torch.linalg.cross(lp - T(a), lp - T(b)) / torch.linalg.norm(T(a) - T(b))
        assert close(fe([torch.zeros(6)]).numpy(), r, 1e-12) and close(J[:, :6], jac_blocks(fe, [6])[0]) and np.all(J[:, 6] == 0)
        n = rng.normal(size=3); n /= np.linalg.norm(n); dd = float(rng.normal())
        rs = np.zeros(1); Js = np.zeros((1, 7))
        L.vilo_eval_surf(abi.dptr(pose), abi.dptr(cp), abi.dptr(n), dd, abi.dptr(rs), abi.dptr(Js))

        def fs(d):         # lidarFactor.hpp:81-84
            x = se3_plus(T(pose), d[0])
            return (T(n) @ (qrot(x[:4], T(cp)) + x[4:]) + dd).reshape(1)
        assert close(fs([torch.zeros(6)]).numpy(), rs, 1e-12) and close(Js[:, :6], jac_blocks(fs, [6])[0])


# ---- MarginalizationFactor -----------------------------------------------------------------------------------------------------
def test_prior_factor_jacobian_by_autograd_at_the_linearisation_point(oracle):
    o = oracle.default_options()
    win, prior, _ = synth.make_window(5, o)
    J0, r0, blocks = abi.prior_to_numpy(prior)
    sizes = [b["size"] for b in blocks]; loc = [6 if s == 7 else s for s in sizes]

    def f(d):              # marginalization_factor.cpp:343-363 at x = x0 (+) d
        dx = []
        for b, di in zip(blocks, d):
            x0 = T(b["x0"])
            if b["size"] == 7:
                x = pose_plus(x0, di)
                dq = qmul(qinv(x0[3:]), x[3:])
                dx += [x[:3] - x0[:3], 2 * dq[:3]]               # q0^-1 q has w > 0 next to the identity: no sign flip (:356-361)
            else:
                dx.append(di)
        return T(r0) + T(J0) @ torch.cat(dx)
    r, J = oracle.eval_factor("prior", None, [b["x0"] for b in blocks], prior, sizes=sizes, nres=prior.n)
    assert close(f([torch.zeros(k) for k in loc]).numpy(), r, 1e-12)
    for b, a, c, k in zip(blocks, J, jac_blocks(f, loc), loc):
        assert close(a[:, :k], c, 1e-10)
        assert np.array_equal(a[:, :k], J0[:, b["idx"]:b["idx"] + k]), "the reference's Jacobian IS the J0 block — at any x (exact only at x0)"

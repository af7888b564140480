// ORACLE — TEST INFRASTRUCTURE ONLY. "parity unpinned" vs Ceres (see omath.hpp).
#include "factors.hpp"
#include <cstdio>

namespace ora {

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };  // vins_estimator/parameters.h:67-74

// ---- loss functions (Ceres 2.0 loss_function.cc semantics) ------------------------------------------
void CauchyLoss::Evaluate(double s, double rho[3]) const {
    const double sum = 1.0 + s * c;
    const double inv = 1.0 / sum;
    rho[0] = b * std::log(sum);
    rho[1] = std::max(std::numeric_limits<double>::min(), inv);
    rho[2] = -c * (inv * inv);
}
void HuberLoss::Evaluate(double s, double rho[3]) const {
    if (s > b) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

// marginalization_factor.cpp:37-68
void apply_corrector(const LossFunction *loss, int nres, double *residuals, int nblocks, const int *block_sizes,
                     double **jacobians, double *rho_out) {
    double sq_norm = 0, rho[3];
    for (int i = 0; i < nres; i++) sq_norm += residuals[i] * residuals[i];
    loss->Evaluate(sq_norm, rho);
    if (rho_out) { rho_out[0] = rho[0]; rho_out[1] = rho[1]; rho_out[2] = rho[2]; }
    const double sqrt_rho1 = std::sqrt(rho[1]);
    double residual_scaling, alpha_sq_norm;
    if (sq_norm == 0.0 || rho[2] <= 0.0) {
        residual_scaling = sqrt_rho1;
        alpha_sq_norm = 0.0;
    } else {
        const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
        const double alpha = 1.0 - std::sqrt(D);
        residual_scaling = sqrt_rho1 / (1 - alpha);
        alpha_sq_norm = alpha / sq_norm;
    }
    if (jacobians) {
        for (int b = 0; b < nblocks; b++) {
            double *J = jacobians[b];
            if (!J) continue;
            const int nc = block_sizes[b];
            // J = sqrt_rho1 * (J - alpha_sq_norm * r * (r^T J))
            for (int c = 0; c < nc; c++) {
                double rtj = 0;
                for (int r = 0; r < nres; r++) rtj += residuals[r] * J[r * nc + c];
                for (int r = 0; r < nres; r++) J[r * nc + c] = sqrt_rho1 * (J[r * nc + c] - alpha_sq_norm * residuals[r] * rtj);
            }
        }
    }
    for (int i = 0; i < nres; i++) residuals[i] *= residual_scaling;
}

// ---- local parameterisations --------------------------------------------------------------------------
void pose_plus(const double *x, const double *delta, double *xp) {
    Q4 q = Q4::from_xyzw(x + 3);
    Q4 dq = deltaQ(V3(delta + 3));
    for (int i = 0; i < 3; i++) xp[i] = x[i] + delta[i];
    Q4 r = normalized(q * dq);
    r.to_xyzw(xp + 3);
}
void getTransformFromSe3(const double se3[6], Q4 &q, V3 &t) {
    V3 omega(se3), upsilon(se3 + 3);
    M3 Omega = skew(omega);
    double theta = norm(omega);
    double half_theta = 0.5 * theta;
    double imag_factor;
    double real_factor = std::cos(half_theta);
    if (theta < 1e-10) {
        double theta_sq = theta * theta;
        double theta_po4 = theta_sq * theta_sq;
        imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
        imag_factor = std::sin(half_theta) / theta;
    }
    q = Q4(real_factor, imag_factor * omega.x, imag_factor * omega.y, imag_factor * omega.z);
    M3 J;
    if (theta < 1e-10) {
        J = toR(q);
    } else {
        M3 Omega2 = Omega * Omega;
        J = M3::Identity() + ((1 - std::cos(theta)) / (theta * theta)) * Omega + ((theta - std::sin(theta)) / std::pow(theta, 3)) * Omega2;
    }
    t = J * upsilon;
}
void se3_plus(const double *x, const double *delta, double *xp) {
    Q4 dq; V3 dt;
    getTransformFromSe3(delta, dq, dt);
    Q4 q = Q4::from_xyzw(x);
    V3 t(x + 4);
    Q4 qp = dq * q;
    V3 tp = dq * t + dt;
    qp.to_xyzw(xp);
    xp[4] = tp.x; xp[5] = tp.y; xp[6] = tp.z;
}

// ---- helpers -----------------------------------------------------------------------------------------
static inline void set_block3(double *J, int ncols, int r0, int c0, const M3 &m) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) J[(r0 + i) * ncols + c0 + j] = m(i, j);
}
// J(rows x ncols) = S(rows x rows) * J
static void left_mul(const double *S, int rows, double *J, int ncols) {
    std::vector<double> tmp((size_t)rows * ncols);
    for (int i = 0; i < rows; i++)
        for (int j = 0; j < ncols; j++) {
            double s = 0;
            for (int k = 0; k < rows; k++) s += S[i * rows + k] * J[k * ncols + j];
            tmp[i * ncols + j] = s;
        }
    std::memcpy(J, tmp.data(), sizeof(double) * rows * ncols);
}

// ---- ProjectionFactor ----------------------------------------------------------------------------------
ProjectionFactor::ProjectionFactor(V3 pi, V3 pj, double si) : pts_i(pi), pts_j(pj), sqrt_info(si) {
    num_residuals = 2;
    block_sizes = {7, 7, 7, 1};
}
struct ProjGeom {
    V3 pts_camera_i, pts_imu_i, pts_w, pts_imu_j, pts_camera_j;
    M3 Ri, Rj, ric;
};
static void proj_common(const double *const *p, V3 pts_i_eff, ProjGeom &g, Q4 &Qi, Q4 &Qj, Q4 &qic, V3 &Pi, V3 &Pj, V3 &tic) {
    Pi = V3(p[0]); Qi = Q4::from_xyzw(p[0] + 3);
    Pj = V3(p[1]); Qj = Q4::from_xyzw(p[1] + 3);
    tic = V3(p[2]); qic = Q4::from_xyzw(p[2] + 3);
    double inv_dep_i = p[3][0];
    g.pts_camera_i = pts_i_eff / inv_dep_i;
    g.pts_imu_i = qic * g.pts_camera_i + tic;
    g.pts_w = Qi * g.pts_imu_i + Pi;
    g.pts_imu_j = inverse(Qj) * (g.pts_w - Pj);
    g.pts_camera_j = inverse(qic) * (g.pts_imu_j - tic);
}
static void proj_jacobians(const ProjGeom &g, double sqrt_info, V3 Pi, V3 Pj, V3 tic, V3 pts_i_eff, double inv_dep_i,
                           double **jac, double reduce[6]) {
    const double dep_j = g.pts_camera_j.z;
    // reduce (2x3)
    reduce[0] = sqrt_info * (1. / dep_j); reduce[1] = 0; reduce[2] = sqrt_info * (-g.pts_camera_j.x / (dep_j * dep_j));
    reduce[3] = 0; reduce[4] = sqrt_info * (1. / dep_j); reduce[5] = sqrt_info * (-g.pts_camera_j.y / (dep_j * dep_j));
    const M3 &Ri = g.Ri, &Rj = g.Rj, &ric = g.ric;
    M3 ricT = transpose(ric), RjT = transpose(Rj);
    auto reduce_mul = [&](const M3 &L, const M3 &Rr, double *J /*2x7*/) {
        for (int r = 0; r < 2; r++) {
            for (int c = 0; c < 3; c++) {
                double s = 0, t = 0;
                for (int k = 0; k < 3; k++) { s += reduce[3 * r + k] * L(k, c); t += reduce[3 * r + k] * Rr(k, c); }
                J[7 * r + c] = s; J[7 * r + 3 + c] = t;
            }
            J[7 * r + 6] = 0;
        }
    };
    if (jac[0]) {
        M3 L = ricT * RjT;
        M3 Rr = ricT * RjT * Ri * (-skew(g.pts_imu_i));
        reduce_mul(L, Rr, jac[0]);
    }
    if (jac[1]) {
        M3 L = ricT * (-RjT);
        M3 Rr = ricT * skew(g.pts_imu_j);
        reduce_mul(L, Rr, jac[1]);
    }
    if (jac[2]) {
        M3 L = ricT * (RjT * Ri - M3::Identity());
        M3 tmp_r = ricT * RjT * Ri * ric;
        M3 Rr = -tmp_r * skew(g.pts_camera_i) + skew(tmp_r * g.pts_camera_i) +
                skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
        reduce_mul(L, Rr, jac[2]);
    }
    if (jac[3]) {
        M3 M = ricT * RjT * Ri * ric;
        V3 v = M * pts_i_eff;
        for (int r = 0; r < 2; r++)
            jac[3][r] = (reduce[3 * r] * v.x + reduce[3 * r + 1] * v.y + reduce[3 * r + 2] * v.z) * -1.0 / (inv_dep_i * inv_dep_i);
    }
}
bool ProjectionFactor::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    ProjGeom g; Q4 Qi, Qj, qic; V3 Pi, Pj, tic;
    proj_common(p, pts_i, g, Qi, Qj, qic, Pi, Pj, tic);
    double dep_j = g.pts_camera_j.z;
    residuals[0] = sqrt_info * (g.pts_camera_j.x / dep_j - pts_j.x);
    residuals[1] = sqrt_info * (g.pts_camera_j.y / dep_j - pts_j.y);
    if (jacobians) {
        g.Ri = toR(Qi); g.Rj = toR(Qj); g.ric = toR(qic);
        double reduce[6];
        proj_jacobians(g, sqrt_info, Pi, Pj, tic, pts_i, p[3][0], jacobians, reduce);
    }
    return true;
}

// ---- ProjectionTdFactor ----------------------------------------------------------------------------------
ProjectionTdFactor::ProjectionTdFactor(V3 pi, V3 pj, const double vel_i[2], const double vel_j[2], double tdi, double tdj,
                                       double rowi, double rowj, double si, double TR_, double ROW_)
    : pts_i(pi), pts_j(pj), velocity_i(vel_i[0], vel_i[1], 0), velocity_j(vel_j[0], vel_j[1], 0), td_i(tdi), td_j(tdj),
      row_i(rowi - ROW_ / 2), row_j(rowj - ROW_ / 2), sqrt_info(si), TR(TR_), ROW(ROW_) {
    num_residuals = 2;
    block_sizes = {7, 7, 7, 1, 1};
}
bool ProjectionTdFactor::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    double td = p[4][0];
    V3 pts_i_td = pts_i - (td - td_i + TR / ROW * row_i) * velocity_i;
    V3 pts_j_td = pts_j - (td - td_j + TR / ROW * row_j) * velocity_j;
    ProjGeom g; Q4 Qi, Qj, qic; V3 Pi, Pj, tic;
    proj_common(p, pts_i_td, g, Qi, Qj, qic, Pi, Pj, tic);
    double dep_j = g.pts_camera_j.z;
    residuals[0] = sqrt_info * (g.pts_camera_j.x / dep_j - pts_j_td.x);
    residuals[1] = sqrt_info * (g.pts_camera_j.y / dep_j - pts_j_td.y);
    if (jacobians) {
        g.Ri = toR(Qi); g.Rj = toR(Qj); g.ric = toR(qic);
        double reduce[6];
        proj_jacobians(g, sqrt_info, Pi, Pj, tic, pts_i_td, p[3][0], jacobians, reduce);
        if (jacobians[4]) {
            M3 M = transpose(g.ric) * transpose(g.Rj) * g.Ri * g.ric;
            V3 v = M * velocity_i;
            double inv_dep_i = p[3][0];
            for (int r = 0; r < 2; r++) {
                double a = (reduce[3 * r] * v.x + reduce[3 * r + 1] * v.y + reduce[3 * r + 2] * v.z) / inv_dep_i * -1.0;
                double b = sqrt_info * (r == 0 ? velocity_j.x : velocity_j.y);
                jacobians[4][r] = a + b;
            }
        }
    }
    return true;
}

// ---- IMUFactor ----------------------------------------------------------------------------------------------
IMUFactor::IMUFactor(const vilf_imu_preint *p, V3 G_) : pre(p), G(G_) {
    num_residuals = 15;
    block_sizes = {7, 9, 7, 9};
}
void IMUFactor::sqrt_info(const vilf_imu_preint *p, double out[225]) {
    Mat cov(15, 15), inv;
    for (int i = 0; i < 225; i++) cov.d[i] = p->covariance[i];
    inverse_pplu(cov, inv);
    cholesky_lower(inv);  // inv now holds L (lower)
    for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) out[i * 15 + j] = inv(j, i);  // L^T
}
static inline M3 jblock(const vilf_imu_preint *p, int r0, int c0) {
    M3 m;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m(i, j) = p->jacobian[(r0 + i) * 15 + c0 + j];
    return m;
}
bool IMUFactor::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    V3 Pi(p[0]); Q4 Qi = Q4::from_xyzw(p[0] + 3);
    V3 Vi(p[1]), Bai(p[1] + 3), Bgi(p[1] + 6);
    V3 Pj(p[2]); Q4 Qj = Q4::from_xyzw(p[2] + 3);
    V3 Vj(p[3]), Baj(p[3] + 3), Bgj(p[3] + 6);

    const double sum_dt = pre->sum_dt;
    M3 dp_dba = jblock(pre, O_P, O_BA), dp_dbg = jblock(pre, O_P, O_BG), dq_dbg = jblock(pre, O_R, O_BG);
    M3 dv_dba = jblock(pre, O_V, O_BA), dv_dbg = jblock(pre, O_V, O_BG);
    V3 lin_ba(pre->linearized_ba), lin_bg(pre->linearized_bg);
    V3 delta_p(pre->delta_p), delta_v(pre->delta_v);
    Q4 delta_q = Q4::from_xyzw(pre->delta_q);

    // IntegrationBase::evaluate (integration_base.h:160-186)
    V3 dba = Bai - lin_ba, dbg = Bgi - lin_bg;
    Q4 corrected_delta_q = delta_q * deltaQ(dq_dbg * dbg);
    V3 corrected_delta_v = delta_v + dv_dba * dba + dv_dbg * dbg;
    V3 corrected_delta_p = delta_p + dp_dba * dba + dp_dbg * dbg;
    Q4 Qi_inv = inverse(Qi);
    V3 rp = Qi_inv * (0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p;
    V3 rq = 2.0 * (inverse(corrected_delta_q) * (Qi_inv * Qj)).vec();
    V3 rv = Qi_inv * (G * sum_dt + Vj - Vi) - corrected_delta_v;
    V3 rba = Baj - Bai, rbg = Bgj - Bgi;
    double r[15] = {rp.x, rp.y, rp.z, rq.x, rq.y, rq.z, rv.x, rv.y, rv.z, rba.x, rba.y, rba.z, rbg.x, rbg.y, rbg.z};

    double S[225];
    if (whiten) sqrt_info(pre, S);
    else for (int i = 0; i < 225; i++) S[i] = (i % 16 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 15; i++) { double s = 0; for (int k = 0; k < 15; k++) s += S[i * 15 + k] * r[k]; residuals[i] = s; }

    if (jacobians) {
        M3 RiT = toR(Qi_inv);
        if (jacobians[0]) {
            double *J = jacobians[0];
            std::memset(J, 0, sizeof(double) * 15 * 7);
            set_block3(J, 7, O_P, O_P, -RiT);
            set_block3(J, 7, O_P, O_R, skew(Qi_inv * (0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
            Q4 cdq = delta_q * deltaQ(dq_dbg * (Bgi - lin_bg));
            set_block3(J, 7, O_R, O_R, -bottomRight3(Qleft(inverse(Qj) * Qi) * Qright(cdq)));
            set_block3(J, 7, O_V, O_R, skew(Qi_inv * (G * sum_dt + Vj - Vi)));
            left_mul(S, 15, J, 7);
        }
        if (jacobians[1]) {
            double *J = jacobians[1];
            std::memset(J, 0, sizeof(double) * 15 * 9);
            set_block3(J, 9, O_P, O_V - O_V, -RiT * sum_dt);
            set_block3(J, 9, O_P, O_BA - O_V, -dp_dba);
            set_block3(J, 9, O_P, O_BG - O_V, -dp_dbg);
            set_block3(J, 9, O_R, O_BG - O_V, -bottomRight3(Qleft(inverse(Qj) * Qi * delta_q)) * dq_dbg);
            set_block3(J, 9, O_V, O_V - O_V, -RiT);
            set_block3(J, 9, O_V, O_BA - O_V, -dv_dba);
            set_block3(J, 9, O_V, O_BG - O_V, -dv_dbg);
            set_block3(J, 9, O_BA, O_BA - O_V, -M3::Identity());
            set_block3(J, 9, O_BG, O_BG - O_V, -M3::Identity());
            left_mul(S, 15, J, 9);
        }
        if (jacobians[2]) {
            double *J = jacobians[2];
            std::memset(J, 0, sizeof(double) * 15 * 7);
            set_block3(J, 7, O_P, O_P, RiT);
            Q4 cdq = delta_q * deltaQ(dq_dbg * (Bgi - lin_bg));
            set_block3(J, 7, O_R, O_R, bottomRight3(Qleft(inverse(cdq) * Qi_inv * Qj)));
            left_mul(S, 15, J, 7);
        }
        if (jacobians[3]) {
            double *J = jacobians[3];
            std::memset(J, 0, sizeof(double) * 15 * 9);
            set_block3(J, 9, O_V, O_V - O_V, RiT);
            set_block3(J, 9, O_BA, O_BA - O_V, M3::Identity());
            set_block3(J, 9, O_BG, O_BG - O_V, M3::Identity());
            left_mul(S, 15, J, 9);
        }
    }
    return true;
}

// ---- LidarFactor ----------------------------------------------------------------------------------------------
LidarFactor::LidarFactor(const vilf_lidar_constraint *c, const vilf_options *o) {
    num_residuals = 6;
    block_sizes = {7, 7};
    lidar_q = Q4::from_xyzw(c->q);
    lidar_t = V3(c->t);
    RIC = M3::from(o->RIC); RCL = M3::from(o->RCL);
    TIC = V3(o->TIC); TCL = V3(o->TCL);
}
bool LidarFactor::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    V3 Pi(p[0]); Q4 Qi = Q4::from_xyzw(p[0] + 3);
    V3 Pj(p[1]); Q4 Qj = Q4::from_xyzw(p[1] + 3);
    Q4 qil = fromR(RIC * RCL);
    V3 til = RIC * TCL + TIC;
    Q4 qli = inverse(qil);
    V3 tli = -(inverse(qil) * til);
    V3 rp = qli * (inverse(Qi) * (Pj - Pi) - til - ((qil * lidar_q) * tli)) - lidar_t;
    V3 rq = 2.0 * (inverse(qil * lidar_q * qli) * (inverse(Qi) * Qj)).vec();
    // residual is weighted, jacobians are NOT (lidar_factor.h:39-42 vs :44-75) — reference behaviour, reproduced.
    residuals[0] = 10.0 * rp.x; residuals[1] = 10.0 * rp.y; residuals[2] = 10.0 * rp.z;
    residuals[3] = 100.0 * rq.x; residuals[4] = 100.0 * rq.y; residuals[5] = 100.0 * rq.z;
    if (jacobians) {
        if (jacobians[0]) {
            double *J = jacobians[0];
            std::memset(J, 0, sizeof(double) * 6 * 7);
            set_block3(J, 7, O_P, O_P, -toR(qli * inverse(Qi)));
            set_block3(J, 7, O_P, O_R, toR(qli) * skew(inverse(Qi) * (Pj - Pi)));
            Q4 cdq = qil * lidar_q * qli;
            set_block3(J, 7, O_R, O_R, -bottomRight3(Qleft(inverse(Qj) * Qi) * Qright(cdq)));
        }
        if (jacobians[1]) {
            double *J = jacobians[1];
            std::memset(J, 0, sizeof(double) * 6 * 7);
            set_block3(J, 7, O_P, O_P, toR(qli * inverse(Qi)));
            Q4 cdq = qil * lidar_q * qli;
            set_block3(J, 7, O_R, O_R, bottomRight3(Qleft(inverse(cdq) * inverse(Qi) * Qj)));
        }
    }
    return true;
}

// ---- MarginalizationFactor -------------------------------------------------------------------------------------
MarginalizationFactor::MarginalizationFactor(const vilf_prior *p) : prior(p) {
    num_residuals = p->n;
    for (int i = 0; i < p->n_blocks; i++) block_sizes.push_back(p->block_size[i]);
}
bool MarginalizationFactor::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    const int n = prior->n;
    std::vector<double> dx(n, 0.0);
    for (int i = 0; i < prior->n_blocks; i++) {
        int size = prior->block_size[i];
        int idx = prior->block_idx[i];
        const double *x = p[i];
        const double *x0 = prior->block_x0[i];
        if (size != 7) {
            for (int k = 0; k < size; k++) dx[idx + k] = x[k] - x0[k];
        } else {
            for (int k = 0; k < 3; k++) dx[idx + k] = x[k] - x0[k];
            Q4 dq = inverse(Q4::from_xyzw(x0 + 3)) * Q4::from_xyzw(x + 3);
            V3 v = 2.0 * dq.vec();
            if (!(dq.w >= 0)) v = 2.0 * (-dq.vec());
            dx[idx + 3] = v.x; dx[idx + 4] = v.y; dx[idx + 5] = v.z;
        }
    }
    for (int i = 0; i < n; i++) {
        double s = prior->linearized_residuals[i];
        const double *Jr = prior->linearized_jacobians + (size_t)i * n;
        for (int k = 0; k < n; k++) s += Jr[k] * dx[k];
        residuals[i] = s;
    }
    if (jacobians) {
        for (int i = 0; i < prior->n_blocks; i++) {
            if (!jacobians[i]) continue;
            int size = prior->block_size[i], local = (size == 7 ? 6 : size), idx = prior->block_idx[i];
            double *J = jacobians[i];
            std::memset(J, 0, sizeof(double) * n * size);
            for (int r = 0; r < n; r++)
                for (int c = 0; c < local; c++) J[r * size + c] = prior->linearized_jacobians[(size_t)r * n + idx + c];
        }
    }
    return true;
}

// ---- Edge / Surf (F-LOAM scan-to-map) --------------------------------------------------------------------------
EdgeCostFunction::EdgeCostFunction(V3 c, V3 a, V3 b) : curr_point(c), point_a(a), point_b(b) {
    num_residuals = 3; block_sizes = {7};
}
bool EdgeCostFunction::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    Q4 q = Q4::from_xyzw(p[0]);
    V3 t(p[0] + 4);
    V3 lp = q * curr_point + t;
    V3 nu = cross(lp - point_a, lp - point_b);
    V3 ab = point_a - point_b;
    double ab_norm = norm(ab);
    residuals[0] = nu.x / ab_norm; residuals[1] = nu.y / ab_norm; residuals[2] = nu.z / ab_norm;
    if (jacobians && jacobians[0]) {
        double *J = jacobians[0];
        std::memset(J, 0, sizeof(double) * 21);
        M3 A = -skew(ab);
        M3 left = A * (-skew(lp));  // rotation part
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { J[7 * i + j] = left(i, j) / ab_norm; J[7 * i + 3 + j] = A(i, j) / ab_norm; }
    }
    return true;
}
SurfCostFunction::SurfCostFunction(V3 c, V3 n, double d) : curr_point(c), nrm(n), negative_OA_dot_norm(d) {
    num_residuals = 1; block_sizes = {7};
}
bool SurfCostFunction::Evaluate(double const *const *p, double *residuals, double **jacobians) const {
    Q4 q = Q4::from_xyzw(p[0]);
    V3 t(p[0] + 4);
    V3 pw = q * curr_point + t;
    residuals[0] = dot(nrm, pw) + negative_OA_dot_norm;
    if (jacobians && jacobians[0]) {
        double *J = jacobians[0];
        M3 ns = -skew(pw);
        for (int j = 0; j < 3; j++) J[j] = nrm.x * ns(0, j) + nrm.y * ns(1, j) + nrm.z * ns(2, j);
        J[3] = nrm.x; J[4] = nrm.y; J[5] = nrm.z; J[6] = 0;
    }
    return true;
}

// ---- IntegrationBase ----------------------------------------------------------------------------------------------
void imu_preintegrate(const vilf_imu_noise *nz, const double acc0[3], const double gyr0[3], const double ba_[3],
                      const double bg_[3], int n, const double *dts, const double *accs, const double *gyrs, vilf_imu_preint *out) {
    V3 acc_0(acc0), gyr_0(gyr0), ba(ba_), bg(bg_);
    V3 delta_p, delta_v; Q4 delta_q;
    double sum_dt = 0;
    Mat jac(15, 15), cov(15, 15), noise(18, 18);
    for (int i = 0; i < 15; i++) jac(i, i) = 1.0;
    for (int i = 0; i < 3; i++) {
        noise(i, i) = nz->acc_n * nz->acc_n; noise(3 + i, 3 + i) = nz->gyr_n * nz->gyr_n;
        noise(6 + i, 6 + i) = nz->acc_n * nz->acc_n; noise(9 + i, 9 + i) = nz->gyr_n * nz->gyr_n;
        noise(12 + i, 12 + i) = nz->acc_w * nz->acc_w; noise(15 + i, 15 + i) = nz->gyr_w * nz->gyr_w;
    }
    auto setb = [](Mat &M, int r0, int c0, const M3 &m) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M(r0 + i, c0 + j) = m(i, j); };
    for (int s = 0; s < n; s++) {
        double _dt = dts[s];
        V3 _acc_1(accs + 3 * s), _gyr_1(gyrs + 3 * s);
        // midPointIntegration (integration_base.h:54-128)
        V3 un_acc_0 = delta_q * (acc_0 - ba);
        V3 un_gyr = 0.5 * (gyr_0 + _gyr_1) - bg;
        Q4 result_delta_q = delta_q * Q4(1, un_gyr.x * _dt / 2, un_gyr.y * _dt / 2, un_gyr.z * _dt / 2);
        V3 un_acc_1 = result_delta_q * (_acc_1 - ba);
        V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
        V3 result_delta_p = delta_p + delta_v * _dt + 0.5 * un_acc * _dt * _dt;
        V3 result_delta_v = delta_v + un_acc * _dt;

        V3 w_x = 0.5 * (gyr_0 + _gyr_1) - bg;
        V3 a_0_x = acc_0 - ba, a_1_x = _acc_1 - ba;
        M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
        M3 I3 = M3::Identity();
        M3 Rd = toR(delta_q), Rr = toR(result_delta_q);
        Mat F(15, 15), V(15, 18);
        setb(F, 0, 0, I3);
        setb(F, 0, 3, -0.25 * Rd * R_a_0_x * _dt * _dt + -0.25 * Rr * R_a_1_x * (I3 - R_w_x * _dt) * _dt * _dt);
        setb(F, 0, 6, I3 * _dt);
        setb(F, 0, 9, -0.25 * (Rd + Rr) * _dt * _dt);
        setb(F, 0, 12, -0.25 * Rr * R_a_1_x * _dt * _dt * -_dt);
        setb(F, 3, 3, I3 - R_w_x * _dt);
        setb(F, 3, 12, -1.0 * I3 * _dt);
        setb(F, 6, 3, -0.5 * Rd * R_a_0_x * _dt + -0.5 * Rr * R_a_1_x * (I3 - R_w_x * _dt) * _dt);
        setb(F, 6, 6, I3);
        setb(F, 6, 9, -0.5 * (Rd + Rr) * _dt);
        setb(F, 6, 12, -0.5 * Rr * R_a_1_x * _dt * -_dt);
        setb(F, 9, 9, I3);
        setb(F, 12, 12, I3);
        M3 V03 = 0.25 * (-Rr) * R_a_1_x * _dt * _dt * 0.5 * _dt;
        setb(V, 0, 0, 0.25 * Rd * _dt * _dt);
        setb(V, 0, 3, V03);
        setb(V, 0, 6, 0.25 * Rr * _dt * _dt);
        setb(V, 0, 9, V03);
        setb(V, 3, 3, 0.5 * I3 * _dt);
        setb(V, 3, 9, 0.5 * I3 * _dt);
        setb(V, 6, 0, 0.5 * Rd * _dt);
        M3 V63 = 0.5 * (-Rr) * R_a_1_x * _dt * 0.5 * _dt;
        setb(V, 6, 3, V63);
        setb(V, 6, 6, 0.5 * Rr * _dt);
        setb(V, 6, 9, V63);
        setb(V, 9, 12, I3 * _dt);
        setb(V, 12, 15, I3 * _dt);
        jac = matmul(F, jac);
        cov = matmul(matmul(F, cov), transpose(F));
        Mat vnv = matmul(matmul(V, noise), transpose(V));
        for (int i = 0; i < 225; i++) cov.d[i] += vnv.d[i];
        // propagate (integration_base.h:130-158)
        delta_p = result_delta_p;
        delta_q = normalized(result_delta_q);
        delta_v = result_delta_v;
        sum_dt += _dt;
        acc_0 = _acc_1;
        gyr_0 = _gyr_1;
    }
    out->sum_dt = sum_dt;
    out->delta_p[0] = delta_p.x; out->delta_p[1] = delta_p.y; out->delta_p[2] = delta_p.z;
    delta_q.to_xyzw(out->delta_q);
    out->delta_v[0] = delta_v.x; out->delta_v[1] = delta_v.y; out->delta_v[2] = delta_v.z;
    for (int i = 0; i < 3; i++) { out->linearized_ba[i] = ba_[i]; out->linearized_bg[i] = bg_[i]; }
    for (int i = 0; i < 225; i++) { out->jacobian[i] = jac.d[i]; out->covariance[i] = cov.d[i]; }
}

}  // namespace ora

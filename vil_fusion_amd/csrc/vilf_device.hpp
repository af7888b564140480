// vilf_device.hpp — fp64 device math for the sliding-window back-end (gfx950).
//
// Restates, for one GPU thread, the arithmetic of the reference's factors (paths relative to
// src/visual_inertial_lidar/vins_estimator/):
//   utility/utility.h:16-143           deltaQ (un-normalised), Qleft/Qright, R2ypr/ypr2R
//   factor/pose_local_parameterization.cpp:3-19
//   factor/projection_factor.cpp:21-121, factor/imu_factor.h:19-179 (+ integration_base.h:160-186),
//   factor/lidar_factor.h:19-78, feature_tracker/include/lidarFactor.hpp:21-102, common.h:137-176
// Quaternions are stored x y z w (Eigen coeffs order); matrices row-major.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define VD __device__ __forceinline__

// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global load and store in flight (s_waitcnt vmcnt(0)), which ends any prefetch
// at the next barrier. Use only where nothing in GLOBAL memory written before it is read by another thread after it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Sum over the 64 lanes of a wave, the total in every lane, as a fixed binary tree: lane pairs, quads, half rows, rows (DPP: quad_perm, row_half_mirror, row_mirror),
// then the four row sums as (r0 + r1) + (r2 + r3) through v_readlane. At every stage all lanes of a group hold the same group sum, so every lane ends with the same
// bits. ~150 cycles instead of ~800 for a butterfly of six ds_bpermute round trips (three of these per Householder step of the marginalization, 28 per evaluation
// of the scan-to-map LM solve ...). ALL 64 lanes must be active.
template <int CTRL>
VD double vilf_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
VD double vilf_wave_sum64(double v) {
    v += vilf_dpp_f64<0xB1>(v);        // quad_perm [1 0 3 2]: lane ^ 1
    v += vilf_dpp_f64<0x4E>(v);        // quad_perm [2 3 0 1]: lane ^ 2
    v += vilf_dpp_f64<0x141>(v);       // row_half_mirror: the other quad of the 8-lane half row
    v += vilf_dpp_f64<0x140>(v);       // row_mirror: the other half of the 16-lane row
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
    return (r0 + r1) + (r2 + r3);
}
VD double vilf_wave_max64(double v) {           // the same tree with fmax
    v = fmax(v, vilf_dpp_f64<0xB1>(v)); v = fmax(v, vilf_dpp_f64<0x4E>(v)); v = fmax(v, vilf_dpp_f64<0x141>(v)); v = fmax(v, vilf_dpp_f64<0x140>(v));
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 0), __builtin_amdgcn_readlane(__double2loint(v), 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 16), __builtin_amdgcn_readlane(__double2loint(v), 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 32), __builtin_amdgcn_readlane(__double2loint(v), 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 48), __builtin_amdgcn_readlane(__double2loint(v), 48));
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

namespace vd {

struct Q { double x, y, z, w; };

VD Q q_load(const double *p) { return Q{p[0], p[1], p[2], p[3]}; }
VD void q_store(double *p, const Q &q) { p[0] = q.x; p[1] = q.y; p[2] = q.z; p[3] = q.w; }
VD Q q_mul(const Q &a, const Q &b) {
    return Q{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
             a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
             a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x,
             a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
VD double q_sqnorm(const Q &q) { return q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w; }
// Eigen QuaternionBase::inverse(): conj / squaredNorm (also for non-unit quaternions)
VD Q q_inv(const Q &q) { double n = 1.0 / q_sqnorm(q); return Q{-q.x * n, -q.y * n, -q.z * n, q.w * n}; }
VD Q q_normalized(const Q &q) { double n = 1.0 / sqrt(q_sqnorm(q)); return Q{q.x * n, q.y * n, q.z * n, q.w * n}; }
// Utility::deltaQ: (1, theta/2), NOT normalised (utility.h:16-29)
VD Q q_delta(double tx, double ty, double tz) { return Q{tx * 0.5, ty * 0.5, tz * 0.5, 1.0}; }
// Eigen toRotationMatrix (no normalisation)
VD void q_toR(const Q &q, double *R) {
    const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// Eigen _transformVector
VD void q_rot(const Q &q, const double *v, double *o) {
    double ux = 2 * (q.y * v[2] - q.z * v[1]), uy = 2 * (q.z * v[0] - q.x * v[2]), uz = 2 * (q.x * v[1] - q.y * v[0]);
    o[0] = v[0] + q.w * ux + (q.y * uz - q.z * uy);
    o[1] = v[1] + q.w * uy + (q.z * ux - q.x * uz);
    o[2] = v[2] + q.w * uz + (q.x * uy - q.y * ux);
}
// Eigen Quaterniond(Matrix3d)
VD Q q_fromR(const double *m) {
    Q q;
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q.w = 0.5 * t;
        t = 0.5 / t;
        q.x = (m[7] - m[5]) * t; q.y = (m[2] - m[6]) * t; q.z = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double v[3];
        t = sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        v[i] = 0.5 * t;
        t = 0.5 / t;
        q.w = (m[3 * k + j] - m[3 * j + k]) * t;
        v[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        v[k] = (m[3 * k + i] + m[3 * i + k]) * t;
        q.x = v[0]; q.y = v[1]; q.z = v[2];
    }
    return q;
}

VD void m3_mul(const double *a, const double *b, double *c) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j];
}
VD void m3_mulT(const double *a, const double *b, double *c) {  // c = a^T b
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c[3 * i + j] = a[i] * b[j] + a[3 + i] * b[3 + j] + a[6 + i] * b[6 + j];
}
VD void m3_vec(const double *a, const double *v, double *o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a[3 * i] * v[0] + a[3 * i + 1] * v[1] + a[3 * i + 2] * v[2];
}
VD void m3T_vec(const double *a, const double *v, double *o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a[i] * v[0] + a[3 + i] * v[1] + a[6 + i] * v[2];
}
VD void skew3(const double *v, double *s) {
    s[0] = 0; s[1] = -v[2]; s[2] = v[1];
    s[3] = v[2]; s[4] = 0; s[5] = -v[0];
    s[6] = -v[1]; s[7] = v[0]; s[8] = 0;
}
// bottom-right 3x3 of Qleft(q) = w I + [v]x ; of Qright(p) = w I - [v]x  (utility.h:51-69)
VD void qleft_br(const Q &q, double *m) {
    m[0] = q.w; m[1] = -q.z; m[2] = q.y; m[3] = q.z; m[4] = q.w; m[5] = -q.x; m[6] = -q.y; m[7] = q.x; m[8] = q.w;
}
// bottom-right 3x3 of Qleft(a) * Qright(b) (4x4 product, rows/cols 1..3)
VD void qleft_qright_br(const Q &a, const Q &b, double *m) {
    // L = [[aw, -av^T],[av, aw I + [av]x]], R = [[bw, -bv^T],[bv, bw I - [bv]x]]
    // (L R)_br = av (-bv^T) + (aw I + [av]x)(bw I - [bv]x)
    double La[9], Rb[9], t[9];
    qleft_br(a, La);
    Rb[0] = b.w; Rb[1] = b.z; Rb[2] = -b.y; Rb[3] = -b.z; Rb[4] = b.w; Rb[5] = b.x; Rb[6] = b.y; Rb[7] = -b.x; Rb[8] = b.w;
    m3_mul(La, Rb, t);
    const double av[3] = {a.x, a.y, a.z}, bv[3] = {b.x, b.y, b.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) m[3 * i + j] = t[3 * i + j] - av[i] * bv[j];
}

// PoseLocalParameterization::Plus
VD void pose_plus(const double *x, const double *d, double *o) {
    o[0] = x[0] + d[0]; o[1] = x[1] + d[1]; o[2] = x[2] + d[2];
    Q q = q_normalized(q_mul(q_load(x + 3), q_delta(d[3], d[4], d[5])));
    q_store(o + 3, q);
}

// Utility::R2ypr / ypr2R (degrees)
VD void R2ypr(const double *R, double *ypr) {
    const double PI = 3.14159265358979323846;
    double y = atan2(R[3], R[0]);
    double p = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
    double r = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
    ypr[0] = y / PI * 180.0; ypr[1] = p / PI * 180.0; ypr[2] = r / PI * 180.0;
}
VD void ypr2R(const double *ypr, double *R) {
    const double PI = 3.14159265358979323846;
    double y = ypr[0] / 180.0 * PI, p = ypr[1] / 180.0 * PI, r = ypr[2] / 180.0 * PI;
    double Rz[9] = {cos(y), -sin(y), 0, sin(y), cos(y), 0, 0, 0, 1};
    double Ry[9] = {cos(p), 0, sin(p), 0, 1, 0, -sin(p), 0, cos(p)};
    double Rx[9] = {1, 0, 0, 0, cos(r), -sin(r), 0, sin(r), cos(r)};
    double t[9];
    m3_mul(Rz, Ry, t);
    m3_mul(t, Rx, R);
}

// ---- robust losses (Ceres loss_function.cc) -----------------------------------------------------------
// Cauchy(a): rho0 = b log(1+s/b), rho1 = 1/(1+s/b); rho2 < 0 always => the Corrector reduces to sqrt(rho1) scaling
// (marginalization_factor.cpp:45-49 / ceres corrector.cc).
// 1/sqrt(x) in fp64: v_rsq_f64 seed (~2^-23 relative) + two Newton steps
// the same from ONE third-order step (e = 1 - x y^2; y (1 + e/2 + 3 e^2/8)): seed error 2^-23 cubed is below fp64 resolution and the dependent
// chain is four operations instead of eight — for the pivot chain of the tile Cholesky, where nothing else hides it
VD double rsqrt_h3(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(0.375, e, 0.5), y);
}
VD double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    return y;
}
// broadcast lane `src` (compile-time constant after unrolling) of a double through SGPRs (v_readlane_b32 x2)
VD double readlane_f64(double v, int src) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
VD void cauchy(double s, double b, double &rho0, double &sqrt_rho1) {
    double sum = 1.0 + s / b;
    rho0 = b * log(sum);
    sqrt_rho1 = sqrt(1.0 / sum);
}
VD void huber(double s, double a, double &rho0, double &sqrt_rho1) {
    double b = a * a;
    if (s > b) { double r = sqrt(s); rho0 = 2.0 * a * r - b; sqrt_rho1 = sqrt(a / r); }
    else { rho0 = s; sqrt_rho1 = 1.0; }
}

// ---- ProjectionTdFactor (projection_td_factor.cpp:34-141): ProjectionFactor on observations shifted by the pixel velocity over the
// time offset (+ rolling-shutter row time), plus the 2x1 jacobian with respect to td. row_*_c = uv.y - ROW / 2 (:19-20).
template <bool JAC>
VD void projection_td_eval(const double *Pi, const double *Ri, const double *Pj, const double *Rj, const double *ric, const double *tic,
                           const double *pts_i, const double *pts_j, const double *vel_i, const double *vel_j, double td, double td_i, double td_j,
                           double row_i_c, double row_j_c, double tr_over_row, double inv_dep, double sqrt_info,
                           double *r, double *Ji, double *Jj, double *Jf, double *Jex, double *Jtd);

// ---- ProjectionFactor (projection_factor.cpp:21-121) --------------------------------------------------
// Ri, Rj, ric are rotation matrices of the (unit) parameter quaternions. Outputs the LOCAL (tangent) jacobians:
// Ji, Jj: 2x6 row-major [dp | dtheta], Jf: 2x1. JAC=false: residual only.
template <bool JAC>
VD void projection_eval(const double *Pi, const double *Ri, const double *Pj, const double *Rj, const double *ric, const double *tic,
                        const double *pts_i, const double *pts_j, double inv_dep, double sqrt_info,
                        double *r, double *Ji, double *Jj, double *Jf, double *Jex = nullptr, double *M2out = nullptr) {
    double pc_i[3] = {pts_i[0] / inv_dep, pts_i[1] / inv_dep, pts_i[2] / inv_dep};
    double p_imu_i[3], pw[3], d[3], p_imu_j[3], e[3], pc_j[3];
    m3_vec(ric, pc_i, p_imu_i);
    p_imu_i[0] += tic[0]; p_imu_i[1] += tic[1]; p_imu_i[2] += tic[2];
    m3_vec(Ri, p_imu_i, pw);
    d[0] = pw[0] + Pi[0] - Pj[0]; d[1] = pw[1] + Pi[1] - Pj[1]; d[2] = pw[2] + Pi[2] - Pj[2];
    m3T_vec(Rj, d, p_imu_j);
    e[0] = p_imu_j[0] - tic[0]; e[1] = p_imu_j[1] - tic[1]; e[2] = p_imu_j[2] - tic[2];
    m3T_vec(ric, e, pc_j);
    const double inv_z = 1.0 / pc_j[2];
    r[0] = sqrt_info * (pc_j[0] * inv_z - pts_j[0]);
    r[1] = sqrt_info * (pc_j[1] * inv_z - pts_j[1]);
    if (JAC) {
        // reduce = sqrt_info * [1/z 0 -x/z^2; 0 1/z -y/z^2]
        const double r00 = sqrt_info * inv_z, r02 = -sqrt_info * pc_j[0] * inv_z * inv_z, r12 = -sqrt_info * pc_j[1] * inv_z * inv_z;
        // A = ric^T Rj^T ; M = reduce * A (2x3)
        double A[9];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) A[3 * i + j] = ric[i] * Rj[3 * j] + ric[3 + i] * Rj[3 * j + 1] + ric[6 + i] * Rj[3 * j + 2];
        double M[6];
#pragma unroll
        for (int j = 0; j < 3; j++) { M[j] = r00 * A[j] + r02 * A[6 + j]; M[3 + j] = r00 * A[3 + j] + r12 * A[6 + j]; }
        // M2 = M * Ri
        double M2[6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            M2[j] = M[0] * Ri[j] + M[1] * Ri[3 + j] + M[2] * Ri[6 + j];
            M2[3 + j] = M[3] * Ri[j] + M[4] * Ri[3 + j] + M[5] * Ri[6 + j];
        }
        if (M2out) for (int k = 0; k < 6; k++) M2out[k] = M2[k];       // reduce * ric^T Rj^T Ri (ProjectionTdFactor needs it for d/dtd)
        // N = reduce * ric^T
        double N[6];
#pragma unroll
        for (int j = 0; j < 3; j++) { N[j] = r00 * ric[3 * j] + r02 * ric[3 * j + 2]; N[3 + j] = r00 * ric[3 * j + 1] + r12 * ric[3 * j + 2]; }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const double *m = M + 3 * rr, *m2 = M2 + 3 * rr, *n = N + 3 * rr;
            double *ji = Ji + 6 * rr, *jj = Jj + 6 * rr;
            ji[0] = m[0]; ji[1] = m[1]; ji[2] = m[2];
            // m2 * (-skew(p)) : row vector a, a * (-[p]x) = (p x a)^T ... (a [p]x)_k = sum_i a_i [p]x_{ik}
            // [p]x = [0 -pz py; pz 0 -px; -py px 0]; a*[p]x = (a1 pz - a2 py, -a0 pz + a2 px, a0 py - a1 px)
            ji[3] = -(m2[1] * p_imu_i[2] - m2[2] * p_imu_i[1]);
            ji[4] = -(-m2[0] * p_imu_i[2] + m2[2] * p_imu_i[0]);
            ji[5] = -(m2[0] * p_imu_i[1] - m2[1] * p_imu_i[0]);
            jj[0] = -m[0]; jj[1] = -m[1]; jj[2] = -m[2];
            jj[3] = n[1] * p_imu_j[2] - n[2] * p_imu_j[1];
            jj[4] = -n[0] * p_imu_j[2] + n[2] * p_imu_j[0];
            jj[5] = n[0] * p_imu_j[1] - n[1] * p_imu_j[0];
        }
        // Jf = M2 * (ric * pts_i) * (-1/lambda^2)
        double rp[3];
        m3_vec(ric, pts_i, rp);
        const double s = -1.0 / (inv_dep * inv_dep);
        Jf[0] = (M2[0] * rp[0] + M2[1] * rp[1] + M2[2] * rp[2]) * s;
        Jf[1] = (M2[3] * rp[0] + M2[4] * rp[1] + M2[5] * rp[2]) * s;
        if (Jex) {   // projection_factor.cpp:98-107 (extrinsic block; needed by the marginalization, where Ex_Pose is never constant)
            double RjtRi[9], L[9], tmp_r[9], T1[9], S1[9], S2[9], S3[9], v[3], u[3], wv[3];
            m3_mulT(Rj, Ri, RjtRi);
            double Dm[9];
            for (int k = 0; k < 9; k++) Dm[k] = RjtRi[k];
            Dm[0] -= 1; Dm[4] -= 1; Dm[8] -= 1;
            m3_mulT(ric, Dm, L);                         // ric^T (Rj^T Ri - I)
            m3_mul(RjtRi, ric, T1);
            m3_mulT(ric, T1, tmp_r);                     // ric^T Rj^T Ri ric
            skew3(pc_i, S1);
            m3_mul(tmp_r, S1, T1);                       // tmp_r * skew(pts_camera_i)
            m3_vec(tmp_r, pc_i, v); skew3(v, S2);
            m3_vec(Ri, tic, u);
            u[0] += Pi[0] - Pj[0]; u[1] += Pi[1] - Pj[1]; u[2] += Pi[2] - Pj[2];
            m3T_vec(Rj, u, wv);
            wv[0] -= tic[0]; wv[1] -= tic[1]; wv[2] -= tic[2];
            m3T_vec(ric, wv, v); skew3(v, S3);
            double Rr[9];
            for (int k = 0; k < 9; k++) Rr[k] = -T1[k] + S2[k] + S3[k];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Jex[j] = r00 * L[j] + r02 * L[6 + j];           Jex[6 + j] = r00 * L[3 + j] + r12 * L[6 + j];
                Jex[3 + j] = r00 * Rr[j] + r02 * Rr[6 + j];     Jex[9 + j] = r00 * Rr[3 + j] + r12 * Rr[6 + j];
            }
        }
    }
}

// ---- ProjectionFactor through per-frame-pair geometry tables ---------------------------------------------------------
// Everything in projection_factor.cpp:36-40,66-110 that depends only on the frame pair (i, j) and the extrinsic is hoisted
// into a 42-double table per pair: A = ric^T Rj^T, ARi = A Ri, Rji = Rj^T Ri, tji = Rj^T (Pi - Pj), Rc = ric^T Rji ric,
// tc = ric^T (Rji tic + tji - tic). A factor is then pts_camera_j = Rc (pts_i / lambda) + tc plus a few 2x3 products.
#define PT_LD 42
VD void pair_table(const double *Pi, const double *Ri, const double *Pj, const double *Rj, const double *ric, const double *tic, double *pt) {
    double A[9], ARi[9], Rji[9], tji[3], T[9], Rc[9], d[3], u[3], tc[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) A[3 * i + j] = ric[i] * Rj[3 * j] + ric[3 + i] * Rj[3 * j + 1] + ric[6 + i] * Rj[3 * j + 2];
    m3_mul(A, Ri, ARi);
    m3_mulT(Rj, Ri, Rji);
    d[0] = Pi[0] - Pj[0]; d[1] = Pi[1] - Pj[1]; d[2] = Pi[2] - Pj[2];
    m3T_vec(Rj, d, tji);
    m3_mul(Rji, ric, T);
    m3_mulT(ric, T, Rc);
    m3_vec(Rji, tic, u);
    u[0] += tji[0] - tic[0]; u[1] += tji[1] - tic[1]; u[2] += tji[2] - tic[2];
    m3T_vec(ric, u, tc);
#pragma unroll
    for (int k = 0; k < 9; k++) { pt[k] = A[k]; pt[9 + k] = ARi[k]; pt[18 + k] = Rji[k]; pt[30 + k] = Rc[k]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { pt[27 + k] = tji[k]; pt[39 + k] = tc[k]; }
}
template <bool JAC>
VD void projection_eval_pair(const double *pt, const double *ric, const double *tic, const double *pts_i, const double *pts_j, double inv_dep,
                             double sqrt_info, double *r, double *Ji, double *Jj, double *Jf) {
    const double il = 1.0 / inv_dep;
    const double pc_i[3] = {pts_i[0] * il, pts_i[1] * il, pts_i[2] * il};
    double pc_j[3];
    m3_vec(pt + 30, pc_i, pc_j);
    pc_j[0] += pt[39]; pc_j[1] += pt[40]; pc_j[2] += pt[41];
    const double inv_z = 1.0 / pc_j[2];
    r[0] = sqrt_info * (pc_j[0] * inv_z - pts_j[0]);
    r[1] = sqrt_info * (pc_j[1] * inv_z - pts_j[1]);
    if (JAC) {
        const double r00 = sqrt_info * inv_z, r02 = -sqrt_info * pc_j[0] * inv_z * inv_z, r12 = -sqrt_info * pc_j[1] * inv_z * inv_z;
        const double *A = pt, *ARi = pt + 9, *Rji = pt + 18;
        double p_imu_i[3], p_imu_j[3];
        m3_vec(ric, pc_i, p_imu_i);
        p_imu_i[0] += tic[0]; p_imu_i[1] += tic[1]; p_imu_i[2] += tic[2];
        m3_vec(Rji, p_imu_i, p_imu_j);
        p_imu_j[0] += pt[27]; p_imu_j[1] += pt[28]; p_imu_j[2] += pt[29];
        double M[6], M2[6], N[6];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            M[j] = r00 * A[j] + r02 * A[6 + j]; M[3 + j] = r00 * A[3 + j] + r12 * A[6 + j];
            M2[j] = r00 * ARi[j] + r02 * ARi[6 + j]; M2[3 + j] = r00 * ARi[3 + j] + r12 * ARi[6 + j];
            N[j] = r00 * ric[3 * j] + r02 * ric[3 * j + 2]; N[3 + j] = r00 * ric[3 * j + 1] + r12 * ric[3 * j + 2];
        }
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
            const double *m = M + 3 * rr, *m2 = M2 + 3 * rr, *n = N + 3 * rr;
            double *ji = Ji + 6 * rr, *jj = Jj + 6 * rr;
            ji[0] = m[0]; ji[1] = m[1]; ji[2] = m[2];
            ji[3] = -(m2[1] * p_imu_i[2] - m2[2] * p_imu_i[1]);
            ji[4] = -(-m2[0] * p_imu_i[2] + m2[2] * p_imu_i[0]);
            ji[5] = -(m2[0] * p_imu_i[1] - m2[1] * p_imu_i[0]);
            jj[0] = -m[0]; jj[1] = -m[1]; jj[2] = -m[2];
            jj[3] = n[1] * p_imu_j[2] - n[2] * p_imu_j[1];
            jj[4] = -n[0] * p_imu_j[2] + n[2] * p_imu_j[0];
            jj[5] = n[0] * p_imu_j[1] - n[1] * p_imu_j[0];
        }
        // Jf = reduce * (ric^T Rj^T Ri ric) * pts_i * (-1/lambda^2) = reduce * Rc * pts_i * (-1/lambda^2)
        double rp[3];
        m3_vec(pt + 30, pts_i, rp);
        const double s = -il * il;
        Jf[0] = (r00 * rp[0] + r02 * rp[2]) * s;
        Jf[1] = (r00 * rp[1] + r12 * rp[2]) * s;
    }
}

// ---- lidarFactor (lidar_factor.h:19-78) -----------------------------------------------------------------
// consts: qil (unit), til, lidar_q, lidar_t. Jacobians LOCAL 6x6 each, NOT weighted (reference quirk), residual weighted.
template <bool JAC>
VD void lidar_between_eval(const double *posei, const double *posej, const Q &qil, const double *til, const Q &lq, const double *lt,
                           double *r, double *Ji, double *Jj) {
    Q Qi = q_load(posei + 3), Qj = q_load(posej + 3);
    Q Qi_inv = q_inv(Qi), qli = q_inv(qil);
    double dP[3] = {posej[0] - posei[0], posej[1] - posei[1], posej[2] - posei[2]};
    double a[3], tli[3], b[3], c[3], rp[3];
    q_rot(Qi_inv, dP, a);                       // Qi^-1 (Pj - Pi)
    q_rot(qli, til, tli); tli[0] = -tli[0]; tli[1] = -tli[1]; tli[2] = -tli[2];
    Q qm = q_mul(qil, lq);
    q_rot(qm, tli, b);
    c[0] = a[0] - til[0] - b[0]; c[1] = a[1] - til[1] - b[1]; c[2] = a[2] - til[2] - b[2];
    q_rot(qli, c, rp);
    Q cdq = q_mul(qm, qli);                     // qil * lidar_q * qli
    Q rq = q_mul(q_inv(cdq), q_mul(Qi_inv, Qj));
    r[0] = 10.0 * (rp[0] - lt[0]); r[1] = 10.0 * (rp[1] - lt[1]); r[2] = 10.0 * (rp[2] - lt[2]);
    r[3] = 100.0 * 2.0 * rq.x; r[4] = 100.0 * 2.0 * rq.y; r[5] = 100.0 * 2.0 * rq.z;
    if (JAC) {
#pragma unroll
        for (int k = 0; k < 36; k++) { Ji[k] = 0; Jj[k] = 0; }
        double Rm[9], Rl[9], S[9], T[9];
        q_toR(q_mul(qli, Qi_inv), Rm);
        q_toR(qli, Rl);
        skew3(a, S);
        m3_mul(Rl, S, T);
        double B[9], Cm[9];
        qleft_qright_br(q_mul(q_inv(Qj), Qi), cdq, B);
        qleft_br(q_mul(q_inv(cdq), q_mul(Qi_inv, Qj)), Cm);
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                Ji[6 * i + j] = -Rm[3 * i + j];
                Ji[6 * i + 3 + j] = T[3 * i + j];
                Ji[6 * (3 + i) + 3 + j] = -B[3 * i + j];
                Jj[6 * i + j] = Rm[3 * i + j];
                Jj[6 * (3 + i) + 3 + j] = Cm[3 * i + j];
            }
    }
}

// ---- IMUFactor raw part (imu_factor.h:19-179): residual and jacobians BEFORE the sqrt_info multiplication --------
// imu record layout (288 doubles): [0] sum_dt, [1..3] dp, [4..7] dq(xyzw), [8..10] dv, [11..13] lin_ba, [14..16] lin_bg,
// [17..25] dp_dba, [26..34] dp_dbg, [35..43] dq_dbg, [44..52] dv_dba, [53..61] dv_dbg, [62..286] sqrt_info 15x15, [287] valid
#define IMU_REC 288
#define IMU_SQRT 62
// Jraw: 15 x LD row-major (LD >= 30), columns [pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9]; ZERO: clear the 15 x 30 block first
template <bool JAC, int LD = 30, bool ZERO = true>
VD void imu_raw_eval(const double *posei, const double *sbi, const double *posej, const double *sbj, const double *rec, const double *G,
                     double *r, double *Jraw) {
    const double dt = rec[0];
    const double *dp = rec + 1, *dv = rec + 8, *lba = rec + 11, *lbg = rec + 14;
    const double *dp_dba = rec + 17, *dp_dbg = rec + 26, *dq_dbg = rec + 35, *dv_dba = rec + 44, *dv_dbg = rec + 53;
    Q Qi = q_load(posei + 3), Qj = q_load(posej + 3), delta_q = q_load(rec + 4);
    Q Qi_inv = q_inv(Qi);
    double dba[3] = {sbi[3] - lba[0], sbi[4] - lba[1], sbi[5] - lba[2]};
    double dbg[3] = {sbi[6] - lbg[0], sbi[7] - lbg[1], sbi[8] - lbg[2]};
    double th[3];
    m3_vec(dq_dbg, dbg, th);
    Q cdq = q_mul(delta_q, q_delta(th[0], th[1], th[2]));
    double t1[3], t2[3], cdv[3], cdp[3];
    m3_vec(dv_dba, dba, t1); m3_vec(dv_dbg, dbg, t2);
    for (int k = 0; k < 3; k++) cdv[k] = dv[k] + t1[k] + t2[k];
    m3_vec(dp_dba, dba, t1); m3_vec(dp_dbg, dbg, t2);
    for (int k = 0; k < 3; k++) cdp[k] = dp[k] + t1[k] + t2[k];
    double a[3], b[3], ra[3], rb[3];
    for (int k = 0; k < 3; k++) {
        a[k] = 0.5 * G[k] * dt * dt + posej[k] - posei[k] - sbi[k] * dt;
        b[k] = G[k] * dt + sbj[k] - sbi[k];
    }
    q_rot(Qi_inv, a, ra);
    q_rot(Qi_inv, b, rb);
    Q QiQj = q_mul(Qi_inv, Qj);
    Q rq = q_mul(q_inv(cdq), QiQj);
    r[0] = ra[0] - cdp[0]; r[1] = ra[1] - cdp[1]; r[2] = ra[2] - cdp[2];
    r[3] = 2 * rq.x; r[4] = 2 * rq.y; r[5] = 2 * rq.z;
    r[6] = rb[0] - cdv[0]; r[7] = rb[1] - cdv[1]; r[8] = rb[2] - cdv[2];
    for (int k = 0; k < 3; k++) { r[9 + k] = sbj[3 + k] - sbi[3 + k]; r[12 + k] = sbj[6 + k] - sbi[6 + k]; }
    if (JAC) {
        if (ZERO) for (int k = 0; k < 450; k++) Jraw[k] = 0;
        double RiT[9], S[9], B[9];
        q_toR(Qi_inv, RiT);
        auto put = [&](int r0, int c0, const double *m, double s) {
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Jraw[(r0 + i) * LD + c0 + j] = s * m[3 * i + j];
        };
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        // pose_i (cols 0..5)
        put(0, 0, RiT, -1.0);
        skew3(ra, S); put(0, 3, S, 1.0);
        qleft_qright_br(q_mul(q_inv(Qj), Qi), cdq, B); put(3, 3, B, -1.0);
        skew3(rb, S); put(6, 3, S, 1.0);
        // speedbias_i (cols 6..14)
        put(0, 6, RiT, -dt); put(0, 9, dp_dba, -1.0); put(0, 12, dp_dbg, -1.0);
        {
            double L[9], T[9];
            qleft_br(q_mul(q_mul(q_inv(Qj), Qi), delta_q), L);
            m3_mul(L, dq_dbg, T);
            put(3, 12, T, -1.0);
        }
        put(6, 6, RiT, -1.0); put(6, 9, dv_dba, -1.0); put(6, 12, dv_dbg, -1.0);
        put(9, 9, I3, -1.0); put(12, 12, I3, -1.0);
        // pose_j (cols 15..20)
        put(0, 15, RiT, 1.0);
        qleft_br(q_mul(q_inv(cdq), QiQj), B); put(3, 18, B, 1.0);
        // speedbias_j (cols 21..29)
        put(6, 21, RiT, 1.0); put(9, 24, I3, 1.0); put(12, 27, I3, 1.0);
    }
}

// ---- F-LOAM edge / surf factors (lidarFactor.hpp:21-102); pose = [qx qy qz qw tx ty tz]; J local 3x6 / 1x6 (rot | trans) ----
template <bool JAC>
VD void edge_eval(const double *pose, const double *cp, const double *pa, const double *pb, double *r, double *J) {
    Q q = q_load(pose);
    double lp[3];
    q_rot(q, cp, lp);
    lp[0] += pose[4]; lp[1] += pose[5]; lp[2] += pose[6];
    double u[3] = {lp[0] - pa[0], lp[1] - pa[1], lp[2] - pa[2]}, v[3] = {lp[0] - pb[0], lp[1] - pb[1], lp[2] - pb[2]};
    double ab[3] = {pa[0] - pb[0], pa[1] - pb[1], pa[2] - pb[2]};
    double inv = 1.0 / sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    r[0] = (u[1] * v[2] - u[2] * v[1]) * inv;
    r[1] = (u[2] * v[0] - u[0] * v[2]) * inv;
    r[2] = (u[0] * v[1] - u[1] * v[0]) * inv;
    if (JAC) {
        double Sab[9], Slp[9], T[9];
        skew3(ab, Sab); skew3(lp, Slp);
        m3_mul(Sab, Slp, T);  // (-Sab)(-Slp) = Sab Slp
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { J[6 * i + j] = T[3 * i + j] * inv; J[6 * i + 3 + j] = -Sab[3 * i + j] * inv; }
    }
}
template <bool JAC>
VD void surf_eval(const double *pose, const double *cp, const double *n, double d, double *r, double *J) {
    Q q = q_load(pose);
    double pw[3];
    q_rot(q, cp, pw);
    pw[0] += pose[4]; pw[1] += pose[5]; pw[2] += pose[6];
    r[0] = n[0] * pw[0] + n[1] * pw[1] + n[2] * pw[2] + d;
    if (JAC) {
        // n^T (-[pw]x) = (pw x n)^T ... -(n^T [pw]x): (n [p]x)_k as above
        J[0] = -(n[1] * pw[2] - n[2] * pw[1]);
        J[1] = -(-n[0] * pw[2] + n[2] * pw[0]);
        J[2] = -(n[0] * pw[1] - n[1] * pw[0]);
        J[3] = n[0]; J[4] = n[1]; J[5] = n[2];
    }
}
// LocalSE3Parameterization::Plus + getTransformFromSe3 (EstimationMapping.hpp:34-49, common.h:137-176)
VD void se3_plus(const double *x, const double *d, double *o) {
    double th = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    double half = 0.5 * th, imag, real = cos(half);
    if (th < 1e-10) { double t2 = th * th, t4 = t2 * t2; imag = 0.5 - 0.0208333 * t2 + 0.000260417 * t4; }
    else imag = sin(half) / th;
    Q dq{imag * d[0], imag * d[1], imag * d[2], real};
    double Jm[9];
    if (th < 1e-10) q_toR(dq, Jm);
    else {
        double O[9], O2[9];
        skew3(d, O); m3_mul(O, O, O2);
        double a = (1 - cos(th)) / (th * th), b = (th - sin(th)) / (th * th * th);
        for (int k = 0; k < 9; k++) Jm[k] = a * O[k] + b * O2[k];
        Jm[0] += 1; Jm[4] += 1; Jm[8] += 1;
    }
    double dt[3], rt[3];
    m3_vec(Jm, d + 3, dt);
    Q qn = q_mul(dq, q_load(x));
    q_rot(dq, x + 4, rt);
    q_store(o, qn);
    o[4] = rt[0] + dt[0]; o[5] = rt[1] + dt[1]; o[6] = rt[2] + dt[2];
}


template <bool JAC>
VD void projection_td_eval(const double *Pi, const double *Ri, const double *Pj, const double *Rj, const double *ric, const double *tic,
                           const double *pts_i, const double *pts_j, const double *vel_i, const double *vel_j, double td, double td_i, double td_j,
                           double row_i_c, double row_j_c, double tr_over_row, double inv_dep, double sqrt_info,
                           double *r, double *Ji, double *Jj, double *Jf, double *Jex, double *Jtd) {
    const double si = td - td_i + tr_over_row * row_i_c, sj = td - td_j + tr_over_row * row_j_c;
    const double pi_td[3] = {pts_i[0] - si * vel_i[0], pts_i[1] - si * vel_i[1], pts_i[2]};
    const double pj_td[3] = {pts_j[0] - sj * vel_j[0], pts_j[1] - sj * vel_j[1], pts_j[2]};
    double M2[6];
    projection_eval<JAC>(Pi, Ri, Pj, Rj, ric, tic, pi_td, pj_td, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex, M2);
    if (JAC) {
        const double v3[3] = {vel_i[0], vel_i[1], 0.0};
        double rv[3];
        m3_vec(ric, v3, rv);
        Jtd[0] = (M2[0] * rv[0] + M2[1] * rv[1] + M2[2] * rv[2]) / inv_dep * -1.0 + sqrt_info * vel_j[0];
        Jtd[1] = (M2[3] * rv[0] + M2[4] * rv[1] + M2[5] * rv[2]) / inv_dep * -1.0 + sqrt_info * vel_j[1];
    }
}
}  // namespace vd

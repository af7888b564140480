// ORACLE — TEST INFRASTRUCTURE ONLY ("parity unpinned", see oracle_api.h).
// CPU restatement of the visual-inertial alignment that precedes the first window solve:
//   VisualIMUAlignment   vins_estimator/initial/initial_aligment.cpp:199-207
//   solveGyroscopeBias   :3-37    3x3 normal equations over consecutive frames, LDLT, repropagate(0, Bgs[0])
//   LinearAlignment      :125-197 [v_0 .. v_{n-1}, g, s] normal equations (6x10 blocks), x1000, LDLT, |g| gate
//   RefineGravity        :55-123  4 sweeps on the 2-d tangent space of |g| = |G| (TangentBasis :40-53)
// The dense solves are Eigen's A.ldlt().solve(b). Eigen is a system dependency of the reference (eigen3, unpinned; ROS melodic /
// noetic ship 3.3.4 / 3.3.7) and is absent from this image, so ldlt_solve() below restates the published algorithm of
// Eigen 3.3 LDLT (Cholesky/LDLT.h, unblocked lower variant): left-looking, symmetric pivoting on the largest |diagonal| of the
// not-yet-eliminated part AS STORED (left-looking: those entries are still the original ones), first maximum wins.
#include <cmath>
#include <vector>
#include "factors.hpp"
#include "omath.hpp"
#include "oracle_api.h"

using namespace ora;

namespace {

// x = A^-1 b through P A P^T = L D L^T. A is n x n row-major (only the lower triangle is read), destroyed.
void ldlt_solve(int n, std::vector<double> &A, const double *b, double *x) {
    auto a = [&](int i, int j) -> double & { return A[(size_t)i * n + j]; };
    std::vector<int> tr(n);
    std::vector<double> tmp(n);
    for (int k = 0; k < n; k++) {
        int p = k;
        double big = std::fabs(a(k, k));
        for (int i = k + 1; i < n; i++) if (std::fabs(a(i, i)) > big) { big = std::fabs(a(i, i)); p = i; }
        tr[k] = p;
        if (p != k) {                                    // symmetric swap of rows / columns k and p inside the lower triangle
            for (int j = 0; j < k; j++) std::swap(a(k, j), a(p, j));
            for (int i = p + 1; i < n; i++) std::swap(a(i, k), a(i, p));
            std::swap(a(k, k), a(p, p));
            for (int i = k + 1; i < p; i++) std::swap(a(i, k), a(p, i));
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) tmp[j] = a(j, j) * a(k, j);
            double s = 0.0;
            for (int j = 0; j < k; j++) s += a(k, j) * tmp[j];
            a(k, k) -= s;
            for (int i = k + 1; i < n; i++) {
                double t = 0.0;
                for (int j = 0; j < k; j++) t += a(i, j) * tmp[j];
                a(i, k) -= t;
            }
        }
        const double d = a(k, k);
        if (k == 0 && !(std::fabs(d) > 0.0)) {           // zero matrix: Eigen stops with identity transpositions
            for (int j = 0; j < n; j++) tr[j] = j;
            break;
        }
        if (std::fabs(d) > 0.0) for (int i = k + 1; i < n; i++) a(i, k) /= d;
    }
    std::vector<double> y(b, b + n);
    for (int k = 0; k < n; k++) std::swap(y[k], y[tr[k]]);                     // P b
    for (int i = 0; i < n; i++) { double s = y[i]; for (int j = 0; j < i; j++) s -= a(i, j) * y[j]; y[i] = s; }
    const double tol = 1.0 / 1.7976931348623157e308;                           // 1 / NumTraits<double>::highest()
    for (int i = 0; i < n; i++) y[i] = std::fabs(a(i, i)) > tol ? y[i] / a(i, i) : 0.0;
    for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < n; j++) s -= a(j, i) * y[j]; y[i] = s; }
    for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[tr[k]]);                // P^T
    for (int i = 0; i < n; i++) x[i] = y[i];
}

M3 load_R(const double *p) { M3 r; for (int i = 0; i < 9; i++) r.m[i] = p[i]; return r; }
V3 load_V(const double *p) { return {p[0], p[1], p[2]}; }

// accumulate tmp_A^T tmp_A / tmp_A^T tmp_b of one 6 x (6 + T) block pair into the arrow-shaped system (:103-113, :165-175)
void scatter(int n_state, int i, int T, const double *tA /*6 x (6+T)*/, const double *tb, std::vector<double> &A, std::vector<double> &b) {
    const int W = 6 + T;
    std::vector<double> rA((size_t)W * W, 0.0), rb(W, 0.0);
    for (int r = 0; r < W; r++) {
        for (int c = 0; c < W; c++) { double s = 0; for (int k = 0; k < 6; k++) s += tA[k * W + r] * tA[k * W + c]; rA[(size_t)r * W + c] = s; }
        double s = 0; for (int k = 0; k < 6; k++) s += tA[k * W + r] * tb[k]; rb[r] = s;
    }
    auto gi = [&](int r) { return r < 6 ? i * 3 + r : n_state - T + (r - 6); };
    for (int r = 0; r < W; r++) {
        for (int c = 0; c < W; c++) A[(size_t)gi(r) * n_state + gi(c)] += rA[(size_t)r * W + c];
        b[gi(r)] += rb[r];
    }
}

void tangent_basis(V3 g0, V3 &b, V3 &c) {
    V3 a = g0 / norm(g0);
    V3 t{0, 0, 1};
    if (a.x == t.x && a.y == t.y && a.z == t.z) t = {1, 0, 0};
    V3 u = t - a * dot(a, t);
    b = u / norm(u);
    c = cross(a, b);
}

}  // namespace

extern "C" int vilo_ldlt_solve(int n, const double *A, const double *b, double *x) {
    if (n < 1 || !A || !b || !x) return VILF_ERR_INVALID_ARGUMENT;
    std::vector<double> a(A, A + (size_t)n * n);
    ldlt_solve(n, a, b, x);
    return VILF_OK;
}

extern "C" int vilo_visual_imu_alignment(const vilf_options *o, const vilf_imu_noise *nz, int n, const double *frame_R, const double *frame_T,
                                         const double *acc_0, const double *gyr_0, const double *lin_ba, const double *lin_bg, const int *n_samples,
                                         int max_samples, const double *dt, const double *acc, const double *gyr, const double bgs0[3],
                                         double delta_bg[3], double g_out[3], double *x_out, int *n_x, vilf_imu_preint *pre_out, int *ok) {
    if (!o || !nz || n < 2 || !frame_R || !frame_T || !bgs0 || !delta_bg || !g_out || !x_out || !n_x || !ok) return VILF_ERR_INVALID_ARGUMENT;
    const int m = n - 1;                                                      // interval k joins frame k and frame k + 1
    std::vector<vilf_imu_preint> pre(m);
    auto integrate = [&](int k, const double *ba, const double *bg) {
        imu_preintegrate(nz, acc_0 + 3 * k, gyr_0 + 3 * k, ba, bg, n_samples[k], dt + (size_t)k * max_samples, acc + (size_t)k * max_samples * 3,
                         gyr + (size_t)k * max_samples * 3, &pre[k]);
    };
    for (int k = 0; k < m; k++) integrate(k, lin_ba + 3 * k, lin_bg + 3 * k);

    // ---- solveGyroscopeBias (:3-37)
    {
        std::vector<double> A(9, 0.0);
        double b[3] = {0, 0, 0};
        for (int k = 0; k < m; k++) {
            M3 Ri = load_R(frame_R + 9 * k), Rj = load_R(frame_R + 9 * (k + 1));
            Q4 q_ij = fromR(transpose(Ri) * Rj);
            double J[9];
            for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) J[3 * r + c] = pre[k].jacobian[(3 + r) * 15 + 12 + c];   // block(O_R, O_BG)
            V3 e = 2.0 * (inverse(Q4::from_xyzw(pre[k].delta_q)) * q_ij).vec();
            const double ev[3] = {e.x, e.y, e.z};
            for (int r = 0; r < 3; r++) {
                for (int c = 0; c < 3; c++) { double s = 0; for (int t = 0; t < 3; t++) s += J[3 * t + r] * J[3 * t + c]; A[3 * r + c] += s; }
                double s = 0; for (int t = 0; t < 3; t++) s += J[3 * t + r] * ev[t]; b[r] += s;
            }
        }
        ldlt_solve(3, A, b, delta_bg);
    }
    const double zero[3] = {0, 0, 0};
    const double bg_new[3] = {bgs0[0] + delta_bg[0], bgs0[1] + delta_bg[1], bgs0[2] + delta_bg[2]};
    for (int k = 0; k < m; k++) integrate(k, zero, bg_new);                   // repropagate(Vector3d::Zero(), Bgs[0]) (:35)
    if (pre_out) for (int k = 0; k < m; k++) pre_out[k] = pre[k];

    // ---- LinearAlignment (:125-197)
    const V3 TIC = load_V(o->TIC);
    const double Gn = std::sqrt(o->G[0] * o->G[0] + o->G[1] * o->G[1] + o->G[2] * o->G[2]);
    int n_state = 3 * n + 4;
    std::vector<double> A((size_t)n_state * n_state, 0.0), b(n_state, 0.0), x(n_state, 0.0);
    for (int i = 0; i < m; i++) {
        M3 RiT = transpose(load_R(frame_R + 9 * i)), Rj = load_R(frame_R + 9 * (i + 1));
        V3 Ti = load_V(frame_T + 3 * i), Tj = load_V(frame_T + 3 * (i + 1));
        const double t = pre[i].sum_dt;
        double tA[60] = {0}, tb[6] = {0};
        M3 RR = RiT * Rj;
        M3 h = ((RiT * t) * t) * 0.5;                               // R^T * dt * dt / 2, evaluated left to right
        M3 h2 = RiT * t;
        V3 c9 = (RiT * (Tj - Ti)) / 100.0;
        V3 bp = load_V(pre[i].delta_p) + RR * TIC - TIC;
        const double c9v[3] = {c9.x, c9.y, c9.z}, bpv[3] = {bp.x, bp.y, bp.z};
        for (int r = 0; r < 3; r++) {
            tA[r * 10 + r] = -t;
            for (int c = 0; c < 3; c++) { tA[r * 10 + 6 + c] = h(r, c); tA[(3 + r) * 10 + 3 + c] = RR(r, c); tA[(3 + r) * 10 + 6 + c] = h2(r, c); }
            tA[r * 10 + 9] = c9v[r];
            tA[(3 + r) * 10 + r] = -1.0;
            tb[r] = bpv[r];
            tb[3 + r] = pre[i].delta_v[r];
        }
        scatter(n_state, i, 4, tA, tb, A, b);
    }
    for (auto &v : A) v *= 1000.0;
    for (auto &v : b) v *= 1000.0;
    ldlt_solve(n_state, A, b.data(), x.data());
    double s = x[n_state - 1] / 100.0;
    V3 g{x[n_state - 4], x[n_state - 3], x[n_state - 2]};
    for (int i = 0; i < n_state; i++) x_out[i] = x[i];
    *n_x = n_state;
    g_out[0] = g.x; g_out[1] = g.y; g_out[2] = g.z;
    if (std::fabs(norm(g) - Gn) > 1.0 || s < 0) { *ok = 0; return VILF_OK; }

    // ---- RefineGravity (:55-123)
    V3 g0 = g / norm(g) * Gn;
    n_state = 3 * n + 3;
    A.assign((size_t)n_state * n_state, 0.0); b.assign(n_state, 0.0); x.assign(n_state, 0.0);
    for (int sweep = 0; sweep < 4; sweep++) {
        V3 lx, ly;
        tangent_basis(g0, lx, ly);
        for (int i = 0; i < m; i++) {
            M3 RiT = transpose(load_R(frame_R + 9 * i)), Rj = load_R(frame_R + 9 * (i + 1));
            V3 Ti = load_V(frame_T + 3 * i), Tj = load_V(frame_T + 3 * (i + 1));
            const double t = pre[i].sum_dt;
            double tA[54] = {0}, tb[6] = {0};
            M3 RR = RiT * Rj;
            M3 h = ((RiT * t) * t) * 0.5;                               // R^T * dt * dt / 2, evaluated left to right
            M3 h2 = RiT * t;
            V3 hx = h * lx, hy = h * ly, h2x = h2 * lx, h2y = h2 * ly;
            V3 c8 = (RiT * (Tj - Ti)) / 100.0;
            V3 bp = load_V(pre[i].delta_p) + RR * TIC - TIC - h * g0;
            V3 bv = load_V(pre[i].delta_v) - h2 * g0;
            const double hxv[3] = {hx.x, hx.y, hx.z}, hyv[3] = {hy.x, hy.y, hy.z}, h2xv[3] = {h2x.x, h2x.y, h2x.z}, h2yv[3] = {h2y.x, h2y.y, h2y.z};
            const double c8v[3] = {c8.x, c8.y, c8.z}, bpv[3] = {bp.x, bp.y, bp.z}, bvv[3] = {bv.x, bv.y, bv.z};
            for (int r = 0; r < 3; r++) {
                tA[r * 9 + r] = -t;
                tA[r * 9 + 6] = hxv[r]; tA[r * 9 + 7] = hyv[r]; tA[r * 9 + 8] = c8v[r];
                tA[(3 + r) * 9 + r] = -1.0;
                for (int c = 0; c < 3; c++) tA[(3 + r) * 9 + 3 + c] = RR(r, c);
                tA[(3 + r) * 9 + 6] = h2xv[r]; tA[(3 + r) * 9 + 7] = h2yv[r];
                tb[r] = bpv[r]; tb[3 + r] = bvv[r];
            }
            scatter(n_state, i, 3, tA, tb, A, b);
        }
        // A and b are NOT cleared between sweeps (:63-66 are outside the loop): the system accumulates, x1000 each time
        for (auto &v : A) v *= 1000.0;
        for (auto &v : b) v *= 1000.0;
        std::vector<double> Ac = A;
        ldlt_solve(n_state, Ac, b.data(), x.data());
        V3 gn = g0 + lx * x[n_state - 3] + ly * x[n_state - 2];
        g0 = gn / norm(gn) * Gn;
    }
    s = x[n_state - 1] / 100.0;
    x[n_state - 1] = s;
    for (int i = 0; i < n_state; i++) x_out[i] = x[i];
    *n_x = n_state;
    g_out[0] = g0.x; g_out[1] = g0.y; g_out[2] = g0.z;
    *ok = s < 0.0 ? 0 : 1;
    return VILF_OK;
}

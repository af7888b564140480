// ORACLE — TEST INFRASTRUCTURE ONLY ("parity unpinned", see oracle_api.h).
// CPU restatement of the global_fusion pose-graph back-end, src/global_fusion/poseGraphOptimization.cpp:
//   graph    PriorFactor<Pose3> on key frame 0 (variances 1e-12)                                   :565-566, :123-126
//            BetweenFactor<Pose3>(k-1, k, odometry, variances 1e-6 rot / 1e-4 trans)              :577-584, :128-130
//            BetweenFactor<Pose3>(prev, curr, ICP, Robust(Cauchy(1), variances 0.5))               :433-436, :132-138
//   solve    isam->update(graph, initial); isam->update(); calculateEstimate()                    :349-374
// GTSAM is an un-vendored, unpinned system dependency of the reference and absent from this image. Restated from its published
// definitions (gtsam/geometry/Pose3.cpp, Rot3M.cpp, slam/BetweenFactor.h, linear/NoiseModel.cpp, LossFunctions.cpp; 4.1 defaults,
// GTSAM_POSE3_EXPMAP on): Pose3 tangent = [omega, v], retract = p * Expmap(delta), BetweenFactor error = Logmap(measured^-1 (p1^-1 p2))
// with H1 = -LogmapDerivative * Ad((p1^-1 p2)^-1), H2 = LogmapDerivative; Diagonal noise whitens by 1 / sigma; Robust(Cauchy k) reweights
// the whitened factor by sqrt(k^2 / (k^2 + |e|^2)). ISAM2 (incremental Gauss-Newton with a relinearisation threshold of 0.01) is restated
// as batch Gauss-Newton run to convergence on the whole graph — the fixed point ISAM2 tracks. The normal equations are block
// tridiagonal (prior + odometry chain) plus one rank-6 term per loop edge: block Cholesky of the chain, Woodbury for the loops.
#include <cmath>
#include <cstring>
#include <vector>
#include "omath.hpp"
#include "oracle_api.h"

using namespace ora;

namespace {

struct P3 { M3 R; V3 t; };

M3 I3() { M3 m; m(0, 0) = m(1, 1) = m(2, 2) = 1.0; return m; }
M3 so3_exp(V3 w) {
    const double th2 = dot(w, w), th = std::sqrt(th2);
    M3 W = skew(w);
    if (th < 1e-10) return I3() + W;
    return I3() + W * (std::sin(th) / th) + (W * W) * ((1.0 - std::cos(th)) / th2);
}
V3 so3_log(const M3 &R) {                       // Rot3::Logmap (matrix version)
    const double tr = R(0, 0) + R(1, 1) + R(2, 2);
    if (tr + 1.0 < 1e-10) {                     // angle = pi
        V3 o;
        if (std::fabs(R(2, 2) + 1.0) > 1e-5) o = V3{R(0, 2), R(1, 2), 1.0 + R(2, 2)} * (M_PI / std::sqrt(2.0 + 2.0 * R(2, 2)));
        else if (std::fabs(R(1, 1) + 1.0) > 1e-5) o = V3{R(0, 1), 1.0 + R(1, 1), R(2, 1)} * (M_PI / std::sqrt(2.0 + 2.0 * R(1, 1)));
        else o = V3{1.0 + R(0, 0), R(1, 0), R(2, 0)} * (M_PI / std::sqrt(2.0 + 2.0 * R(0, 0)));
        return o;
    }
    double mag;
    const double tr3 = tr - 3.0;
    if (tr3 < -1e-7) { const double th = std::acos((tr - 1.0) / 2.0); mag = th / (2.0 * std::sin(th)); }
    else mag = 0.5 - tr3 / 12.0;
    return V3{R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)} * mag;
}
M3 so3_log_derivative(V3 w) {                   // Rot3::LogmapDerivative = inverse right Jacobian
    const double th2 = dot(w, w);
    if (th2 <= 2.220446049250313e-16) return I3();
    const double th = std::sqrt(th2);
    M3 W = skew(w);
    return I3() + W * 0.5 + (W * W) * (1.0 / th2 - (1.0 + std::cos(th)) / (2.0 * th * std::sin(th)));
}
P3 compose(const P3 &a, const P3 &b) { return P3{a.R * b.R, a.R * b.t + a.t}; }
P3 inverse(const P3 &a) { M3 Rt = transpose(a.R); return P3{Rt, -(Rt * a.t)}; }
P3 se3_exp(const double xi[6]) {
    V3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
    M3 R = so3_exp(w);
    const double th2 = dot(w, w);
    if (th2 < 1e-20) return P3{R, v};
    V3 wv = cross(w, v), tpar = w * (dot(w, v) / th2);
    return P3{R, (wv - R * wv) / th2 + tpar};
}
void se3_log(const P3 &T, double xi[6]) {
    V3 w = so3_log(T.R);
    const double th = norm(w);
    V3 u;
    if (th < 1e-10) u = T.t;
    else {
        M3 W = skew(w / th);
        const double Tan = std::tan(0.5 * th);
        V3 Wt = W * T.t;
        u = T.t - Wt * (0.5 * th) + (W * Wt) * (1.0 - th / (2.0 * Tan));
    }
    xi[0] = w.x; xi[1] = w.y; xi[2] = w.z; xi[3] = u.x; xi[4] = u.y; xi[5] = u.z;
}
struct M6 { double m[36]; double &operator()(int r, int c) { return m[6 * r + c]; } double operator()(int r, int c) const { return m[6 * r + c]; } };
M6 zero6() { M6 a; std::memset(a.m, 0, sizeof(a.m)); return a; }
M6 mul6(const M6 &a, const M6 &b) { M6 c = zero6(); for (int i = 0; i < 6; i++) for (int k = 0; k < 6; k++) { const double v = a(i, k); for (int j = 0; j < 6; j++) c(i, j) += v * b(k, j); } return c; }
void setblk(M6 &a, int r0, int c0, const M3 &b, double s = 1.0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) a(r0 + i, c0 + j) = s * b(i, j); }
M6 adjoint(const P3 &T) { M6 a = zero6(); setblk(a, 0, 0, T.R); setblk(a, 3, 3, T.R); setblk(a, 3, 0, skew(T.t) * T.R); return a; }
M6 se3_log_derivative(const P3 &T) {            // Pose3::LogmapDerivative: [Jw 0; -Jw Q Jw, Jw], Q = computeQforExpmapDerivative(xi)
    double xi[6];
    se3_log(T, xi);
    V3 w{xi[0], xi[1], xi[2]}, v{xi[3], xi[4], xi[5]};
    M3 Jw = so3_log_derivative(w), V = skew(v), W = skew(w), Q;
    const double phi = norm(w);
    M3 WV = W * V, VW = V * W, WVW = WV * W, WW = W * W;
    M3 t1 = WV + VW - WVW, t2 = WW * V + VW * W - WVW * 3.0, t3 = WVW * W + WW * VW;
    if (phi > 1e-5) {
        const double s = std::sin(phi), c = std::cos(phi), p2 = phi * phi, p3 = p2 * phi, p4 = p3 * phi, p5 = p4 * phi;
        Q = V * -0.5 + t1 * ((phi - s) / p3) + t2 * ((1.0 - p2 / 2.0 - c) / p4) - t3 * (0.5 * ((1.0 - p2 / 2.0 - c) / p4 - 3.0 * (phi - s - p3 / 6.0) / p5));
    } else Q = V * -0.5 + t1 * (1.0 / 6.0) - t2 * (1.0 / 24.0) + t3 * (1.0 / 120.0);
    M3 Q2 = -(Jw * Q * Jw);
    M6 J = zero6();
    setblk(J, 0, 0, Jw); setblk(J, 3, 3, Jw); setblk(J, 3, 0, Q2);
    return J;
}
P3 from_qt(const double *p) { return P3{toR(normalized(Q4::from_xyzw(p))), V3{p[4], p[5], p[6]}}; }
void to_qt(const P3 &T, double *p) { Q4 q = fromR(T.R); q.to_xyzw(p); p[4] = T.t.x; p[5] = T.t.y; p[6] = T.t.z; }

// whitened (and robustly re-weighted) between factor: residual e[6], A = d e / d delta_i, B = d e / d delta_j; returns 0.5 rho
double between(const P3 &pi, const P3 &pj, const P3 &meas, const double sigma[6], int robust, double e[6], M6 &A, M6 &B) {
    P3 hx = compose(inverse(pi), pj);
    P3 d = compose(inverse(meas), hx);
    se3_log(d, e);
    M6 Hl = se3_log_derivative(d);
    M6 H1 = adjoint(inverse(hx));
    for (double &v : H1.m) v = -v;
    A = mul6(Hl, H1); B = Hl;
    double r2 = 0;
    for (int k = 0; k < 6; k++) { e[k] /= sigma[k]; r2 += e[k] * e[k]; for (int c = 0; c < 6; c++) { A(k, c) /= sigma[k]; B(k, c) /= sigma[k]; } }
    if (!robust) return 0.5 * r2;
    const double wgt = std::sqrt(1.0 / (1.0 + r2));                     // Cauchy k = 1: sqrt(k^2 / (k^2 + r^2))
    for (int k = 0; k < 6; k++) { e[k] *= wgt; for (int c = 0; c < 6; c++) { A(k, c) *= wgt; B(k, c) *= wgt; } }
    return 0.5 * std::log1p(r2);                                        // rho(r) = k^2 log(1 + r^2 / k^2) / 2 ... x 1/2 again as a cost
}

// 6x6 Cholesky helpers on row-major blocks
bool chol6(M6 &a) { for (int j = 0; j < 6; j++) { double s = a(j, j); for (int k = 0; k < j; k++) s -= a(j, k) * a(j, k); if (!(s > 0)) return false; const double l = std::sqrt(s); a(j, j) = l; for (int i = j + 1; i < 6; i++) { double t = a(i, j); for (int k = 0; k < j; k++) t -= a(i, k) * a(j, k); a(i, j) = t / l; } } return true; }
void chol6_solve(const M6 &L, double *b) { for (int i = 0; i < 6; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L(i, k) * b[k]; b[i] = s / L(i, i); } for (int i = 5; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < 6; k++) s -= L(k, i) * b[k]; b[i] = s / L(i, i); } }

}  // namespace

extern "C" int vilo_pg_between(const double pi_qt[7], const double pj_qt[7], const double meas_qt[7], const double sigma[6], int robust, double e[6], double A36[36], double B36[36], double *cost) {
    M6 A, B;
    const double c = between(from_qt(pi_qt), from_qt(pj_qt), from_qt(meas_qt), sigma, robust, e, A, B);
    std::memcpy(A36, A.m, sizeof(A.m)); std::memcpy(B36, B.m, sizeof(B.m));
    if (cost) *cost = c;
    return VILF_OK;
}
extern "C" int vilo_pg_retract(const double p_qt[7], const double delta[6], double out_qt[7]) { to_qt(compose(from_qt(p_qt), se3_exp(delta)), out_qt); return VILF_OK; }

extern "C" int vilo_posegraph_optimize(int K, double *poses_qt, const double prior_sigma[6], int n_edges, const vilf_pg_edge *edges, int max_iterations, double tol,
                                       int *iterations_out, double *final_cost) {
    if (K < 1 || !poses_qt || !prior_sigma || n_edges < 0 || (n_edges && !edges)) return VILF_ERR_INVALID_ARGUMENT;
    std::vector<P3> x(K);
    for (int k = 0; k < K; k++) x[k] = from_qt(poses_qt + 7 * k);
    const P3 prior = x[0];                                              // PriorFactor(0, poseOrigin): the first key frame pose as handed over
    std::vector<int> loops;
    for (int e = 0; e < n_edges; e++) {
        if (edges[e].i < 0 || edges[e].j < 0 || edges[e].i >= K || edges[e].j >= K || edges[e].i == edges[e].j) return VILF_ERR_INVALID_ARGUMENT;
        if (std::abs(edges[e].i - edges[e].j) != 1) loops.push_back(e);
    }
    const int L = (int)loops.size(), NL = 6 * L;
    int it = 0;
    double cost = 0;
    for (;; it++) {
        // ---- linearise: D[k] diagonal blocks, E[k] = block (k+1, k), g = -J^T r; loop edges keep their Jacobian rows (U^T)
        std::vector<M6> D(K, zero6()), E(std::max(K - 1, 1), zero6());
        std::vector<double> g(6 * (size_t)K, 0.0);
        std::vector<M6> UA(L), UB(L);
        std::vector<double> ue(6 * (size_t)L);
        cost = 0;
        {   // prior on node 0
            double e[6];
            P3 d = compose(inverse(prior), x[0]);
            se3_log(d, e);
            M6 H = se3_log_derivative(d);
            for (int r = 0; r < 6; r++) { e[r] /= prior_sigma[r]; for (int c = 0; c < 6; c++) H(r, c) /= prior_sigma[r]; cost += 0.5 * e[r] * e[r]; }
            for (int a = 0; a < 6; a++) { for (int b = 0; b < 6; b++) { double s = 0; for (int r = 0; r < 6; r++) s += H(r, a) * H(r, b); D[0](a, b) += s; } double s = 0; for (int r = 0; r < 6; r++) s += H(r, a) * e[r]; g[a] -= s; }
        }
        int li = 0;
        for (int ei = 0; ei < n_edges; ei++) {
            const vilf_pg_edge &ed = edges[ei];
            double e[6], mq[7] = {ed.q[0], ed.q[1], ed.q[2], ed.q[3], ed.t[0], ed.t[1], ed.t[2]};
            M6 A, B;
            cost += between(x[ed.i], x[ed.j], from_qt(mq), ed.sigma, ed.robust, e, A, B);
            for (int a = 0; a < 6; a++) { double sa = 0, sb = 0; for (int r = 0; r < 6; r++) { sa += A(r, a) * e[r]; sb += B(r, a) * e[r]; } g[6 * ed.i + a] -= sa; g[6 * ed.j + a] -= sb; }
            if (std::abs(ed.i - ed.j) == 1) {
                const int lo = std::min(ed.i, ed.j);
                const M6 &Jlo = ed.i < ed.j ? A : B, &Jhi = ed.i < ed.j ? B : A;
                for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) {
                    double sll = 0, shh = 0, shl = 0;
                    for (int r = 0; r < 6; r++) { sll += Jlo(r, a) * Jlo(r, b); shh += Jhi(r, a) * Jhi(r, b); shl += Jhi(r, a) * Jlo(r, b); }
                    D[lo](a, b) += sll; D[lo + 1](a, b) += shh; E[lo](a, b) += shl;
                }
            } else { UA[li] = A; UB[li] = B; for (int r = 0; r < 6; r++) ue[6 * li + r] = e[r]; li++; }
        }
        if (it >= max_iterations) break;
        // ---- block Cholesky of the chain: T = Lc Lc^T with diagonal factors C[k] (lower) and sub-diagonal blocks F[k] = E[k] C[k]^-T
        std::vector<M6> C(K), F(std::max(K - 1, 1));
        for (int k = 0; k < K; k++) {
            M6 S = D[k];
            if (k > 0) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int r = 0; r < 6; r++) s += F[k - 1](a, r) * F[k - 1](b, r); S(a, b) -= s; }
            if (!chol6(S)) return VILF_ERR_UNSUPPORTED;               // a key frame without an odometry link to its predecessor
            C[k] = S;
            if (k + 1 < K) for (int a = 0; a < 6; a++) {               // F = E C^-T: solve C y = E(a, :)^T per row
                double y[6];
                for (int c = 0; c < 6; c++) { double s = E[k](a, c); for (int r = 0; r < c; r++) s -= y[r] * S(c, r); y[c] = s / S(c, c); }
                for (int c = 0; c < 6; c++) F[k](a, c) = y[c];
            }
        }
        auto chain_solve = [&](double *b) {                            // b <- T^-1 b
            for (int k = 0; k < K; k++) {
                double *bk = b + 6 * k;
                if (k > 0) for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < 6; r++) s += F[k - 1](a, r) * b[6 * (k - 1) + r]; bk[a] -= s; }
                for (int a = 0; a < 6; a++) { double s = bk[a]; for (int r = 0; r < a; r++) s -= C[k](a, r) * bk[r]; bk[a] = s / C[k](a, a); }
            }
            for (int k = K - 1; k >= 0; k--) {
                double *bk = b + 6 * k;
                if (k + 1 < K) for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < 6; r++) s += F[k](r, a) * b[6 * (k + 1) + r]; bk[a] -= s; }
                for (int a = 5; a >= 0; a--) { double s = bk[a]; for (int r = a + 1; r < 6; r++) s -= C[k](r, a) * bk[r]; bk[a] = s / C[k](a, a); }
            }
        };
        std::vector<double> z = g;
        chain_solve(z.data());
        std::vector<double> delta = z;
        if (L > 0) {   // Woodbury: H = T + U U^T, U column (l, r) = row r of loop l's Jacobian [A at i, B at j]
            std::vector<std::vector<double>> Y(NL, std::vector<double>(6 * (size_t)K, 0.0));
            for (int l = 0; l < L; l++) for (int r = 0; r < 6; r++) {
                std::vector<double> &y = Y[6 * l + r];
                const vilf_pg_edge &ed = edges[loops[l]];
                for (int c = 0; c < 6; c++) { y[6 * ed.i + c] = UA[l](r, c); y[6 * ed.j + c] = UB[l](r, c); }
                chain_solve(y.data());
            }
            auto udot = [&](int l, int r, const double *v) { const vilf_pg_edge &ed = edges[loops[l]]; double s = 0; for (int c = 0; c < 6; c++) s += UA[l](r, c) * v[6 * ed.i + c] + UB[l](r, c) * v[6 * ed.j + c]; return s; };
            Mat Cm(NL, NL);
            std::vector<double> rhs(NL);
            for (int a = 0; a < NL; a++) { for (int b = 0; b < NL; b++) Cm(a, b) = udot(a / 6, a % 6, Y[b].data()) + (a == b ? 1.0 : 0.0); rhs[a] = udot(a / 6, a % 6, z.data()); }
            for (int a = 0; a < NL; a++) for (int b = a + 1; b < NL; b++) { const double s = 0.5 * (Cm(a, b) + Cm(b, a)); Cm(a, b) = Cm(b, a) = s; }
            if (!cholesky_lower(Cm)) return VILF_ERR_UNSUPPORTED;
            chol_solve(Cm, rhs.data());
            for (int a = 0; a < NL; a++) for (size_t k = 0; k < delta.size(); k++) delta[k] -= Y[a][k] * rhs[a];
        }
        double dmax = 0;
        for (int k = 0; k < K; k++) { x[k] = compose(x[k], se3_exp(delta.data() + 6 * k)); for (int c = 0; c < 6; c++) dmax = std::fmax(dmax, std::fabs(delta[6 * k + c])); }
        if (dmax < tol) { it++; break; }
    }
    for (int k = 0; k < K; k++) to_qt(x[k], poses_qt + 7 * k);
    if (iterations_out) *iterations_out = it;
    if (final_cost) *final_cost = cost;
    return VILF_OK;
}

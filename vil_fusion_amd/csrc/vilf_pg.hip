// vilf_pg.hip — the global_fusion pose-graph back-end on the device.
//   vilf_posegraph_optimize ≙ the gtsam graph of src/global_fusion/poseGraphOptimization.cpp (PriorFactor :565, odometry BetweenFactor :577-584,
//   robust loop BetweenFactor :433-436) + isamUpdate (:349-374), ISAM2's incremental Gauss-Newton run as batch Gauss-Newton to convergence.
// One Gauss-Newton iteration =
//   pg_linearize     one thread per factor: Pose3 between / prior error (full SE(3) Logmap), exact Jacobians, whitening, Cauchy re-weighting
//   pg_assemble      one thread per key frame: diagonal / sub-diagonal 6x6 blocks of the odometry chain and -J^T r, summed over the incident
//                    factors in factor order (host-built adjacency: no atomics, fixed summation order)
//   pg_chain_factor  block Cholesky of the block-tridiagonal chain matrix (prior + odometry), one wave walking the K key frames
//   pg_chain_solve   T^-1 applied to 1 + 6 L right-hand sides at once (the gradient and the 6 Jacobian rows of every loop edge), one
//                    thread per column, columns interleaved so a wave's loads coalesce; the chain factors are broadcast loads
//   pg_capacitance   the loop edges as a low-rank (6 L) update: C = I + U^T T^-1 U; dense Cholesky solve of C by the blocked MFMA Cholesky of vilf_lw.hip (vilf_lw_chol_solve)
//   pg_update        delta = z - Y C^-1 U^T z, retract p <- p * Expmap(delta), max |delta|
// The normal equations of a pose graph are block tridiagonal plus a few loop edges: the chain is factorised in O(K), the loops go
// through the Woodbury identity — no general sparse solver, no fill-in, no iteration count that grows with the chain length.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

struct PgCtx {
    DBuf x, prior, edges, rec, adj_off, adj_item, loop_ij, D, E, C, F, g, Y, Cm, rhs, scal, info;
    void release() {
        DBuf *all[] = {&x, &prior, &edges, &rec, &adj_off, &adj_item, &loop_ij, &D, &E, &C, &F, &g, &Y, &Cm, &rhs, &scal, &info};
        for (DBuf *b : all) b->release();
    }
};
void vilf_pg_release(vilf_handle *h) { if (h->pg) { h->pg->release(); delete h->pg; h->pg = nullptr; } }

namespace {
using namespace vd;
#define PG_REC 80                     // per factor: e[6] A[36] B[36] cost pad

// ---- SO(3) / SE(3) (gtsam Rot3 / Pose3 conventions, tangent = [omega, v], row-major 3x3) ---------------------------------------
VD void m3_mul3(const double *a, const double *b, double *c) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c[3 * i + j] = a[3 * i] * b[j] + a[3 * i + 1] * b[3 + j] + a[3 * i + 2] * b[6 + j]; }
VD void so3_exp(const double *w, double *R) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], th = sqrt(th2);
    double W[9], WW[9];
    skew3(w, W);
    double a = 1.0, b = 0.0;
    if (!(th < 1e-10)) { a = sin(th) / th; b = (1.0 - cos(th)) / th2; }
    m3_mul3(W, W, WW);
    for (int i = 0; i < 9; i++) R[i] = a * W[i] + b * WW[i];
    R[0] += 1.0; R[4] += 1.0; R[8] += 1.0;
}
VD void so3_log(const double *R, double *w) {
    const double tr = R[0] + R[4] + R[8];
    if (tr + 1.0 < 1e-10) {
        if (fabs(R[8] + 1.0) > 1e-5) { const double s = M_PI / sqrt(2.0 + 2.0 * R[8]); w[0] = s * R[2]; w[1] = s * R[5]; w[2] = s * (1.0 + R[8]); }
        else if (fabs(R[4] + 1.0) > 1e-5) { const double s = M_PI / sqrt(2.0 + 2.0 * R[4]); w[0] = s * R[1]; w[1] = s * (1.0 + R[4]); w[2] = s * R[7]; }
        else { const double s = M_PI / sqrt(2.0 + 2.0 * R[0]); w[0] = s * (1.0 + R[0]); w[1] = s * R[3]; w[2] = s * R[6]; }
        return;
    }
    double mag;
    const double tr3 = tr - 3.0;
    if (tr3 < -1e-7) { const double th = acos((tr - 1.0) / 2.0); mag = th / (2.0 * sin(th)); }
    else mag = 0.5 - tr3 / 12.0;
    w[0] = mag * (R[7] - R[5]); w[1] = mag * (R[2] - R[6]); w[2] = mag * (R[3] - R[1]);
}
VD void so3_log_derivative(const double *w, double *J) {
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    for (int i = 0; i < 9; i++) J[i] = 0.0;
    J[0] = J[4] = J[8] = 1.0;
    if (th2 <= 2.220446049250313e-16) return;
    const double th = sqrt(th2), c = 1.0 / th2 - (1.0 + cos(th)) / (2.0 * th * sin(th));
    double W[9], WW[9];
    skew3(w, W); m3_mul3(W, W, WW);
    for (int i = 0; i < 9; i++) J[i] += 0.5 * W[i] + c * WW[i];
}
struct P3 { double R[9], t[3]; };
VD void p3_from_qt(const double *p, P3 &T) { q_toR(q_normalized(q_load(p)), T.R); T.t[0] = p[4]; T.t[1] = p[5]; T.t[2] = p[6]; }
VD void p3_inv(const P3 &a, P3 &o) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) o.R[3 * i + j] = a.R[3 * j + i]; double t[3]; m3_vec(o.R, a.t, t); o.t[0] = -t[0]; o.t[1] = -t[1]; o.t[2] = -t[2]; }
VD void p3_mul(const P3 &a, const P3 &b, P3 &o) { m3_mul3(a.R, b.R, o.R); double t[3]; m3_vec(a.R, b.t, t); o.t[0] = t[0] + a.t[0]; o.t[1] = t[1] + a.t[1]; o.t[2] = t[2] + a.t[2]; }
VD void se3_exp(const double *xi, P3 &T) {
    so3_exp(xi, T.R);
    const double *w = xi, *v = xi + 3;
    const double th2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (th2 < 1e-20) { T.t[0] = v[0]; T.t[1] = v[1]; T.t[2] = v[2]; return; }
    const double wv[3] = {w[1] * v[2] - w[2] * v[1], w[2] * v[0] - w[0] * v[2], w[0] * v[1] - w[1] * v[0]};
    double Rwv[3];
    m3_vec(T.R, wv, Rwv);
    const double s = (w[0] * v[0] + w[1] * v[1] + w[2] * v[2]) / th2;
    for (int k = 0; k < 3; k++) T.t[k] = (wv[k] - Rwv[k]) / th2 + w[k] * s;
}
VD void se3_log(const P3 &T, double *xi) {
    so3_log(T.R, xi);
    const double th = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    if (th < 1e-10) { xi[3] = T.t[0]; xi[4] = T.t[1]; xi[5] = T.t[2]; return; }
    const double n[3] = {xi[0] / th, xi[1] / th, xi[2] / th};
    double W[9], Wt[3], WWt[3];
    skew3(n, W); m3_vec(W, T.t, Wt); m3_vec(W, Wt, WWt);
    const double c = 1.0 - th / (2.0 * tan(0.5 * th));
    for (int k = 0; k < 3; k++) xi[3 + k] = T.t[k] - 0.5 * th * Wt[k] + c * WWt[k];
}
// Pose3::LogmapDerivative(T) = [Jw 0; -Jw Q Jw, Jw] (6x6 row-major), Q = computeQforExpmapDerivative(Logmap(T)); also returns xi
VD void se3_log_derivative(const P3 &T, double *xi, double *J) {
    se3_log(T, xi);
    double Jw[9], V[9], W[9], WV[9], VW[9], WVW[9], WW[9], t2a[9], t2b[9], t3a[9], t3b[9], Q[9], tmp[9], Q2[9];
    so3_log_derivative(xi, Jw);
    skew3(xi + 3, V); skew3(xi, W);
    m3_mul3(W, V, WV); m3_mul3(V, W, VW); m3_mul3(WV, W, WVW); m3_mul3(W, W, WW);
    m3_mul3(WW, V, t2a); m3_mul3(VW, W, t2b); m3_mul3(WVW, W, t3a); m3_mul3(WW, VW, t3b);
    const double phi = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    double c1 = 1.0 / 6.0, c2 = -1.0 / 24.0, c3 = 1.0 / 120.0;
    if (phi > 1e-5) {
        const double s = sin(phi), c = cos(phi), p2 = phi * phi, p3 = p2 * phi, p4 = p3 * phi, p5 = p4 * phi;
        c1 = (phi - s) / p3; c2 = (1.0 - p2 / 2.0 - c) / p4; c3 = -0.5 * (c2 - 3.0 * (phi - s - p3 / 6.0) / p5);
    }
    for (int i = 0; i < 9; i++) Q[i] = -0.5 * V[i] + c1 * (WV[i] + VW[i] - WVW[i]) + c2 * (t2a[i] + t2b[i] - 3.0 * WVW[i]) + c3 * (t3a[i] + t3b[i]);
    m3_mul3(Jw, Q, tmp); m3_mul3(tmp, Jw, Q2);
    for (int i = 0; i < 36; i++) J[i] = 0.0;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { J[6 * i + j] = Jw[3 * i + j]; J[6 * (3 + i) + 3 + j] = Jw[3 * i + j]; J[6 * (3 + i) + j] = -Q2[3 * i + j]; }
}

struct PgEdgeDev { int i, j, robust, pad; double q[4], t[3], sigma[6]; };

// factor f = 0: the prior on node 0; f >= 1: edge f - 1. rec[f] = e[6], A[36], B[36], cost
__global__ void pg_linearize(int nF, const PgEdgeDev *edges, const double *x, const double *prior_qt, const double *prior_sigma, double *rec_all) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= nF) return;
    double *rec = rec_all + (size_t)f * PG_REC;
    double e[6], A[36], B[36], sig[6];
    int robust = 0;
    if (f == 0) {
        P3 pr, x0, pri, d;
        p3_from_qt(prior_qt, pr); p3_from_qt(x, x0);
        p3_inv(pr, pri); p3_mul(pri, x0, d);
        se3_log_derivative(d, e, B);
        for (int k = 0; k < 36; k++) A[k] = 0.0;
        for (int k = 0; k < 6; k++) sig[k] = prior_sigma[k];
    } else {
        const PgEdgeDev &ed = edges[f - 1];
        P3 pi, pj, m, pii, hx, mi, d, hxi;
        p3_from_qt(x + 7 * ed.i, pi); p3_from_qt(x + 7 * ed.j, pj);
        const double mq[7] = {ed.q[0], ed.q[1], ed.q[2], ed.q[3], ed.t[0], ed.t[1], ed.t[2]};
        p3_from_qt(mq, m);
        p3_inv(pi, pii); p3_mul(pii, pj, hx);
        p3_inv(m, mi); p3_mul(mi, hx, d);
        se3_log_derivative(d, e, B);
        // H1 = -Ad(hx^-1): [R 0; [t]x R, R]
        p3_inv(hx, hxi);
        double tx[9], txR[9], H1[36];
        skew3(hxi.t, tx); m3_mul3(tx, hxi.R, txR);
        for (int k = 0; k < 36; k++) H1[k] = 0.0;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { H1[6 * r + c] = -hxi.R[3 * r + c]; H1[6 * (3 + r) + 3 + c] = -hxi.R[3 * r + c]; H1[6 * (3 + r) + c] = -txR[3 * r + c]; }
        for (int r = 0; r < 6; r++) for (int c = 0; c < 6; c++) { double s = 0; for (int k = 0; k < 6; k++) s += B[6 * r + k] * H1[6 * k + c]; A[6 * r + c] = s; }
        for (int k = 0; k < 6; k++) sig[k] = ed.sigma[k];
        robust = ed.robust;
    }
    double r2 = 0;
    for (int k = 0; k < 6; k++) { e[k] /= sig[k]; r2 += e[k] * e[k]; for (int c = 0; c < 6; c++) { A[6 * k + c] /= sig[k]; B[6 * k + c] /= sig[k]; } }
    double cost = 0.5 * r2;
    if (robust) {
        const double wgt = sqrt(1.0 / (1.0 + r2));
        for (int k = 0; k < 6; k++) { e[k] *= wgt; for (int c = 0; c < 6; c++) { A[6 * k + c] *= wgt; B[6 * k + c] *= wgt; } }
        cost = 0.5 * log1p(r2);
    }
    for (int k = 0; k < 6; k++) rec[k] = e[k];
    for (int k = 0; k < 36; k++) { rec[6 + k] = A[k]; rec[42 + k] = B[k]; }
    rec[78] = cost;
}

// adjacency item: factor id | side << 28 (0: the node is the factor's i -> A, 1: j -> B) | kind << 29 (0 prior, 1 chain & node is the lower
// index, 2 chain & node is the upper index, 3 loop)
__global__ void pg_assemble(int K, const int *adj_off, const int *adj_item, const double *rec_all, double *D, double *E, double *g) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double Dk[36], Ek[36], gk[6];
    for (int i = 0; i < 36; i++) { Dk[i] = 0.0; Ek[i] = 0.0; }
    for (int i = 0; i < 6; i++) gk[i] = 0.0;
    for (int it = adj_off[k]; it < adj_off[k + 1]; it++) {
        const int item = adj_item[it], f = item & 0x0fffffff, side = (item >> 28) & 1, kind = (item >> 29) & 3;
        const double *rec = rec_all + (size_t)f * PG_REC;
        const double *J = rec + (side ? 42 : 6), *Jo = rec + (side ? 6 : 42);
        for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < 6; r++) s += J[6 * r + a] * rec[r]; gk[a] -= s; }
        if (kind == 3) continue;
        for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int r = 0; r < 6; r++) s += J[6 * r + a] * J[6 * r + b]; Dk[6 * a + b] += s; }
        if (kind == 1) for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) { double s = 0; for (int r = 0; r < 6; r++) s += Jo[6 * r + a] * J[6 * r + b]; Ek[6 * a + b] += s; }   // E[k](a: upper, b: lower)
    }
    for (int i = 0; i < 36; i++) { D[(size_t)k * 36 + i] = Dk[i]; E[(size_t)k * 36 + i] = Ek[i]; }
    for (int i = 0; i < 6; i++) g[6 * k + i] = gk[i];
}
__global__ void pg_cost(int nF, const double *rec_all, double *scal) {
    if (blockIdx.x == 0 && threadIdx.x == 0) { double c = 0; for (int f = 0; f < nF; f++) c += rec_all[(size_t)f * PG_REC + 78]; scal[1] = c; scal[0] = 0.0; }
}

// block Cholesky of the chain: C[k] = chol(D[k] - F[k-1] F[k-1]^T) (lower), F[k] = E[k] C[k]^-T. One wave; lane (a, b) = (lane / 6, lane % 6)
__global__ void __launch_bounds__(64) pg_chain_factor(int K, const double *D, const double *E, double *C, double *F, int *status) {
    __shared__ double sS[36], sF[36];
    const int lane = threadIdx.x, a = lane / 6, b = lane % 6;
    const bool on = lane < 36;
    if (on) sF[lane] = 0.0;
    double dn = on ? D[lane] : 0.0, en = on ? E[lane] : 0.0;
    __syncthreads();
    for (int k = 0; k < K; k++) {
        const double d = dn, e = en;
        if (on && k + 1 < K) { dn = D[(size_t)(k + 1) * 36 + lane]; en = E[(size_t)(k + 1) * 36 + lane]; }     // next blocks in flight
        if (on) { double s = 0; for (int r = 0; r < 6; r++) s += sF[6 * a + r] * sF[6 * b + r]; sS[lane] = d - s; }
        __syncthreads();
        if (lane == 0) {                                                        // 6x6 Cholesky in place (lower)
            bool ok = true;
            for (int j = 0; j < 6 && ok; j++) {
                double s = sS[7 * j];
                for (int r = 0; r < j; r++) s -= sS[6 * j + r] * sS[6 * j + r];
                if (!(s > 0)) { ok = false; break; }
                const double l = sqrt(s);
                sS[7 * j] = l;
                for (int i = j + 1; i < 6; i++) { double t = sS[6 * i + j]; for (int r = 0; r < j; r++) t -= sS[6 * i + r] * sS[6 * j + r]; sS[6 * i + j] = t / l; }
            }
            if (!ok) *status = 1;
        }
        __syncthreads();
        if (*status) return;
        if (on) C[(size_t)k * 36 + lane] = (b <= a) ? sS[lane] : 0.0;
        if (on) sF[lane] = e;                                                   // E[k] rows -> solve C y = E(a, :)^T per row (lanes 0..5)
        __syncthreads();
        if (lane < 6 && k + 1 < K) {
            double y[6];
            for (int c = 0; c < 6; c++) { double s = sF[6 * lane + c]; for (int r = 0; r < c; r++) s -= y[r] * sS[6 * c + r]; y[c] = s / sS[7 * c]; }
            for (int c = 0; c < 6; c++) sF[6 * lane + c] = y[c];
        }
        __syncthreads();
        if (on && k + 1 < K) F[(size_t)k * 36 + lane] = sF[lane];
    }
}

// Y [6 K][NC] (row-major: the NC right-hand sides of one scalar row are adjacent): column 0 = g, column 1 + 6 l + r = U column (l, r)
__global__ void pg_build_rhs(int K, int L, int NC, const int *loop_ij, const int *loop_f, const double *rec_all, const double *g, double *Y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 6 * K) Y[(size_t)t * NC] = g[t];
    if (t < 36 * L) {
        const int l = t / 36, r = (t % 36) / 6, c = t % 6;
        const double *rec = rec_all + (size_t)loop_f[l] * PG_REC;
        Y[(size_t)(6 * loop_ij[2 * l] + c) * NC + 1 + 6 * l + r] = rec[6 + 6 * r + c];
        Y[(size_t)(6 * loop_ij[2 * l + 1] + c) * NC + 1 + 6 * l + r] = rec[42 + 6 * r + c];
    }
}
__global__ void pg_chain_solve(int K, int NC, const double *C, const double *F, double *Y) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= NC) return;
    double prev[6] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < K; k++) {
        double b[6];
        for (int a = 0; a < 6; a++) b[a] = Y[(size_t)(6 * k + a) * NC + col];
        if (k > 0) { const double *Fk = F + (size_t)(k - 1) * 36; for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < 6; r++) s += Fk[6 * a + r] * prev[r]; b[a] -= s; } }
        const double *Ck = C + (size_t)k * 36;
        for (int a = 0; a < 6; a++) { double s = b[a]; for (int r = 0; r < a; r++) s -= Ck[6 * a + r] * b[r]; b[a] = s / Ck[7 * a]; }
        for (int a = 0; a < 6; a++) { Y[(size_t)(6 * k + a) * NC + col] = b[a]; prev[a] = b[a]; }
    }
    for (int k = K - 1; k >= 0; k--) {
        double b[6];
        for (int a = 0; a < 6; a++) b[a] = Y[(size_t)(6 * k + a) * NC + col];
        if (k + 1 < K) { const double *Fk = F + (size_t)k * 36; for (int a = 0; a < 6; a++) { double s = 0; for (int r = 0; r < 6; r++) s += Fk[6 * r + a] * prev[r]; b[a] -= s; } }
        const double *Ck = C + (size_t)k * 36;
        for (int a = 5; a >= 0; a--) { double s = b[a]; for (int r = a + 1; r < 6; r++) s -= Ck[6 * r + a] * b[r]; b[a] = s / Ck[7 * a]; }
        for (int a = 0; a < 6; a++) { Y[(size_t)(6 * k + a) * NC + col] = b[a]; prev[a] = b[a]; }
    }
}
// capacitance matrix Cm = I + sym(U^T Y) (NL x NL) and rhs = U^T z
__global__ void pg_capacitance(int L, int NC, const int *loop_ij, const int *loop_f, const double *rec_all, const double *Y, double *Cm, double *rhs) {
    const int NL = 6 * L;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)NL * (NL + 1)) return;
    const int a = (int)(t / (NL + 1)), b = (int)(t % (NL + 1));
    auto udot = [&](int row, int col) {                       // U column `row` . Y column `col`
        const int l = row / 6, r = row % 6;
        const double *rec = rec_all + (size_t)loop_f[l] * PG_REC;
        const int i = loop_ij[2 * l], j = loop_ij[2 * l + 1];
        double s = 0;
        for (int c = 0; c < 6; c++) s += rec[6 + 6 * r + c] * Y[(size_t)(6 * i + c) * NC + col] + rec[42 + 6 * r + c] * Y[(size_t)(6 * j + c) * NC + col];
        return s;
    };
    if (b == NL) { rhs[a] = udot(a, 0); return; }
    Cm[(size_t)b * NL + a] = 0.5 * (udot(a, 1 + b) + udot(b, 1 + a)) + (a == b ? 1.0 : 0.0);
}
__global__ void pg_update(int K, int NC, int NL, const double *Y, const double *rhs, double *x, double *scal) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double d[6], dm = 0;
    for (int a = 0; a < 6; a++) {
        const double *row = Y + (size_t)(6 * k + a) * NC;
        double s = row[0];
        for (int b = 0; b < NL; b++) s -= row[1 + b] * rhs[b];
        d[a] = s; dm = fmax(dm, fabs(s));
    }
    P3 T, E, O;
    p3_from_qt(x + 7 * k, T); se3_exp(d, E); p3_mul(T, E, O);
    q_store(x + 7 * k, q_fromR(O.R));
    x[7 * k + 4] = O.t[0]; x[7 * k + 5] = O.t[1]; x[7 * k + 6] = O.t[2];
    atomicMax(reinterpret_cast<unsigned long long *>(scal), (unsigned long long)__double_as_longlong(dm));    // non-negative doubles order like their bit patterns
}
}  // namespace

extern "C" int vilf_posegraph_optimize(vilf_handle *h, int K, double *poses_qt, const double prior_sigma[6], int n_edges, const vilf_pg_edge *edges, int max_iterations,
                                       double tol, int *iterations_out, double *final_cost) {
    if (!h || K < 1 || !poses_qt || !prior_sigma || n_edges < 0 || (n_edges && !edges) || max_iterations < 0) return VILF_ERR_INVALID_ARGUMENT;
    if (n_edges >= (1 << 28) - 1) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    if (!h->pg) h->pg = new PgCtx();
    PgCtx *c = h->pg;
    // ---- host: device edge records, adjacency (factor order), loop list
    std::vector<PgEdgeDev> ed(std::max(n_edges, 1));
    std::vector<std::vector<int>> adj(K);
    std::vector<int> loop_ij, loop_f;
    std::vector<char> linked(std::max(K - 1, 1), 0);
    adj[0].push_back(0);                                                       // the prior: factor 0, side 1 (its Jacobian lives in B), kind 0
    adj[0][0] = 0 | (1 << 28) | (0 << 29);
    for (int e = 0; e < n_edges; e++) {
        const vilf_pg_edge &s = edges[e];
        if (s.i < 0 || s.j < 0 || s.i >= K || s.j >= K || s.i == s.j) { h->err = "posegraph: edge endpoint out of range"; return VILF_ERR_INVALID_ARGUMENT; }
        for (int k = 0; k < 6; k++) if (!(s.sigma[k] > 0)) { h->err = "posegraph: sigma must be positive"; return VILF_ERR_INVALID_ARGUMENT; }
        PgEdgeDev &d = ed[e];
        d.i = s.i; d.j = s.j; d.robust = s.robust; d.pad = 0;
        std::memcpy(d.q, s.q, 32); std::memcpy(d.t, s.t, 24); std::memcpy(d.sigma, s.sigma, 48);
        const int f = e + 1;
        if (std::abs(s.i - s.j) == 1) {
            const int lo = std::min(s.i, s.j);
            linked[lo] = 1;
            adj[s.i].push_back(f | (0 << 28) | ((s.i == lo ? 1 : 2) << 29));
            adj[s.j].push_back(f | (1 << 28) | ((s.j == lo ? 1 : 2) << 29));
        } else {
            adj[s.i].push_back(f | (0 << 28) | (3 << 29));
            adj[s.j].push_back(f | (1 << 28) | (3 << 29));
            loop_ij.push_back(s.i); loop_ij.push_back(s.j); loop_f.push_back(f);
        }
    }
    for (int k = 0; k + 1 < K; k++) if (!linked[k]) { h->err = "posegraph: key frames " + std::to_string(k) + " and " + std::to_string(k + 1) + " have no odometry edge"; return VILF_ERR_UNSUPPORTED; }
    const int L = (int)loop_f.size(), NL = 6 * L, NC = 1 + NL, nF = n_edges + 1;
    if (NL > vilf_lw_chol_max_n()) {       // checked before anything is uploaded or enqueued: the loop-closure block (6 L x 6 L) goes through vilf_lw_chol_solve
        h->err = "posegraph: " + std::to_string(L) + " loop edges; the dense loop-closure block supports " + std::to_string(vilf_lw_chol_max_n() / 6);
        return VILF_ERR_UNSUPPORTED;
    }
    std::vector<int> adj_off(K + 1, 0), adj_item;
    for (int k = 0; k < K; k++) { adj_off[k + 1] = adj_off[k] + (int)adj[k].size(); adj_item.insert(adj_item.end(), adj[k].begin(), adj[k].end()); }
    loop_ij.insert(loop_ij.end(), loop_f.begin(), loop_f.end());               // [2 L] endpoints, then [L] factor ids
    const size_t sK = (size_t)K;
    if (!c->x.ensure(sK * 56) || !c->prior.ensure(7 * 8 + 6 * 8) || !c->edges.ensure(ed.size() * sizeof(PgEdgeDev)) || !c->rec.ensure((size_t)nF * PG_REC * 8) ||
        !c->adj_off.ensure((sK + 1) * 4) || !c->adj_item.ensure(std::max<size_t>(adj_item.size(), 1) * 4) || !c->loop_ij.ensure(std::max<size_t>(loop_ij.size(), 1) * 4) ||
        !c->D.ensure(sK * 288) || !c->E.ensure(sK * 288) || !c->C.ensure(sK * 288) || !c->F.ensure(sK * 288) || !c->g.ensure(sK * 48) || !c->Y.ensure(sK * 6 * NC * 8) ||
        !c->Cm.ensure(std::max<size_t>((size_t)(NL + 1) * NL, 1) * 8) || !c->rhs.ensure(std::max(NL, 1) * 8) || !c->scal.ensure(64) || !c->info.ensure(64)) { h->err = "hipMalloc failed (pose graph)"; return VILF_ERR_DEVICE; }
    double pr[13];
    std::memcpy(pr, poses_qt, 56); std::memcpy(pr + 7, prior_sigma, 48);
    HIPCHECK(h, hipMemcpyAsync(c->x.p, poses_qt, sK * 56, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(c->prior.p, pr, sizeof(pr), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(c->edges.p, ed.data(), ed.size() * sizeof(PgEdgeDev), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(c->adj_off.p, adj_off.data(), (sK + 1) * 4, hipMemcpyHostToDevice, h->stream));
    if (!adj_item.empty()) HIPCHECK(h, hipMemcpyAsync(c->adj_item.p, adj_item.data(), adj_item.size() * 4, hipMemcpyHostToDevice, h->stream));
    if (!loop_ij.empty()) HIPCHECK(h, hipMemcpyAsync(c->loop_ij.p, loop_ij.data(), loop_ij.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemsetAsync(c->info.p, 0, 64, h->stream));
    double *x = c->x.as<double>(), *rec = c->rec.as<double>(), *scal = c->scal.as<double>(), *Y = c->Y.as<double>();
    const int *d_lij = c->loop_ij.as<int>(), *d_lf = d_lij + 2 * L;
    int it = 0;
    double cost = 0;
    for (;; it++) {
        hipLaunchKernelGGL(pg_linearize, dim3((nF + 63) / 64), dim3(64), 0, h->stream, nF, c->edges.as<PgEdgeDev>(), x, c->prior.as<double>(), c->prior.as<double>() + 7, rec);
        hipLaunchKernelGGL(pg_cost, dim3(1), dim3(1), 0, h->stream, nF, rec, scal);
        if (it >= max_iterations) {
            double sc[2];
            HIPCHECK(h, hipMemcpyAsync(sc, scal, 16, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(h, hipStreamSynchronize(h->stream));
            cost = sc[1];
            break;
        }
        hipLaunchKernelGGL(pg_assemble, dim3((K + 63) / 64), dim3(64), 0, h->stream, K, c->adj_off.as<int>(), c->adj_item.as<int>(), rec, c->D.as<double>(), c->E.as<double>(), c->g.as<double>());
        hipLaunchKernelGGL(pg_chain_factor, dim3(1), dim3(64), 0, h->stream, K, c->D.as<double>(), c->E.as<double>(), c->C.as<double>(), c->F.as<double>(), c->info.as<int>());
        HIPCHECK(h, hipMemsetAsync(Y, 0, sK * 6 * NC * 8, h->stream));
        hipLaunchKernelGGL(pg_build_rhs, dim3((std::max(6 * K, 36 * L) + 255) / 256), dim3(256), 0, h->stream, K, L, NC, d_lij, d_lf, rec, c->g.as<double>(), Y);
        hipLaunchKernelGGL(pg_chain_solve, dim3((NC + 63) / 64), dim3(64), 0, h->stream, K, NC, c->C.as<double>(), c->F.as<double>(), Y);
        if (L > 0) {
            const size_t ne = (size_t)NL * (NL + 1);
            // the capacitance matrix (symmetric entry by entry) with its right-hand side as row NL, solved by the blocked Cholesky of vilf_lw.hip (no vendor solver)
            hipLaunchKernelGGL(pg_capacitance, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, h->stream, L, NC, d_lij, d_lf, rec, Y, c->Cm.as<double>(), c->Cm.as<double>() + (size_t)NL * NL);
            const int rcs = vilf_lw_chol_solve(h, NL, c->Cm.as<double>(), c->rhs.as<double>(), c->info.as<int>() + 4);
            if (rcs != VILF_OK) return rcs;
        }
        hipLaunchKernelGGL(pg_update, dim3((K + 63) / 64), dim3(64), 0, h->stream, K, NC, NL, Y, c->rhs.as<double>(), x, scal);
        HIPCHECK(h, hipGetLastError());
        double sc[2]; int info[8];
        HIPCHECK(h, hipMemcpyAsync(sc, scal, 16, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipMemcpyAsync(info, c->info.p, 32, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        cost = sc[1];
        if (info[0] || info[4]) { h->err = "posegraph: normal equations not positive definite"; return VILF_ERR_UNSUPPORTED; }
        if (sc[0] < tol) { it++; break; }
    }
    HIPCHECK(h, hipMemcpyAsync(poses_qt, x, sK * 56, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (iterations_out) *iterations_out = it;
    if (final_cost) *final_cost = cost;
    return VILF_OK;
}

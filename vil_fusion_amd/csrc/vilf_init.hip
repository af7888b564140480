// vilf_init.hip — visual-inertial alignment on the device (second half).
//   vilf_visual_imu_alignment ≙ VisualIMUAlignment (vins_estimator/initial/initial_aligment.cpp:199-207):
//     k_preintegrate (vilf_host.hip)  every interval at its linearisation biases            integration_base.h:30-158
//     k_align_gyro                    solveGyroscopeBias: 3x3 normal equations + LDLT         :3-31
//     k_preintegrate                  repropagate(0, Bgs[0]) for every interval               :32-36
//     k_align_linear                  LinearAlignment + 4 RefineGravity sweeps                :55-197
// k_align_linear is ONE workgroup: the systems are (3 n + 4)^2 with n ≈ 11 .. 40 frames — start-up latency work, not throughput work —
// and every stage is laid out so that each number is produced by the same sequence of additions as the reference's loops (per-interval
// blocks summed in ascending interval order), which keeps the result independent of the thread count. The dense solves restate Eigen's
// LDLT (left-looking, symmetric pivoting on the largest stored |diagonal|, first maximum wins): one thread per row for the column update,
// wave-shuffle argmax for the pivot, the working copy of A in LDS while it fits (n <= 40 frames), in global memory beyond that.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

__global__ void k_preintegrate(int n, vilf_imu_noise nz, const double *acc0, const double *gyr0, const double *ba, const double *bg, const int *n_samples, int max_samples,
                               const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out);

namespace {
using namespace vd;

constexpr int AL_THREADS = 256;
constexpr int AL_LDS_FRAMES = 40;                    // 3 n + 4 = 124 -> 124 x 125 doubles = 121 KB of the 160 KB LDS

struct AlignArgs {
    int n;                                           // frames
    const double *frame_R, *frame_T;                 // [n][9], [n][3]
    const vilf_imu_preint *pre;                      // [n-1]
    const double *bgs0;                              // [3]
    double TIC[3], Gnorm;
    double *gy;                                      // [n-1][12] gyro-bias partial sums
    double *ba2, *bg2;                               // [n-1][3] biases of the re-integration
    double *T;                                       // [n-1][66]  tmp_A (6 x W) + tmp_b (6)
    double *RA;                                      // [n-1][110] r_A (W x W) + r_b (W)
    double *A, *b;                                   // accumulated system (3 n + 4)^2, (3 n + 4)
    double *Wg;                                      // global working copy of A (used when it does not fit the LDS)
    double *out;                                     // [0] ok, [1] n_x, [2..4] g, [5..7] delta_bg, [8 ..] x
};

// ---- solveGyroscopeBias ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_align_gyro(AlignArgs a) {
    const int m = a.n - 1;
    for (int k = threadIdx.x; k < m; k += 64) {
        double RiTRj[9];
        m3_mulT(a.frame_R + 9 * k, a.frame_R + 9 * (k + 1), RiTRj);
        const Q q_ij = q_fromR(RiTRj);
        const vilf_imu_preint &p = a.pre[k];
        double J[9];
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) J[3 * r + c] = p.jacobian[(3 + r) * 15 + 12 + c];
        const Q e = q_mul(q_inv(q_load(p.delta_q)), q_ij);
        const double ev[3] = {2.0 * e.x, 2.0 * e.y, 2.0 * e.z};
        double *o = a.gy + 12 * k;
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) { double s = 0; for (int t = 0; t < 3; t++) s += J[3 * t + r] * J[3 * t + c]; o[3 * r + c] = s; }
            double s = 0; for (int t = 0; t < 3; t++) s += J[3 * t + r] * ev[t]; o[9 + r] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double A[9] = {0}, b[3] = {0}, x[3];
        for (int k = 0; k < m; k++) { for (int i = 0; i < 9; i++) A[i] += a.gy[12 * k + i]; for (int i = 0; i < 3; i++) b[i] += a.gy[12 * k + 9 + i]; }
        // 3x3 pivoted LDLT, same recurrence as the big solver below
        int tr[3];
        double tmp[3];
        for (int k = 0; k < 3; k++) {
            int p = k; double big = fabs(A[4 * k]);
            for (int i = k + 1; i < 3; i++) if (fabs(A[4 * i]) > big) { big = fabs(A[4 * i]); p = i; }
            tr[k] = p;
            if (p != k) {
                for (int j = 0; j < k; j++) { double t = A[3 * k + j]; A[3 * k + j] = A[3 * p + j]; A[3 * p + j] = t; }
                for (int i = p + 1; i < 3; i++) { double t = A[3 * i + k]; A[3 * i + k] = A[3 * i + p]; A[3 * i + p] = t; }
                { double t = A[4 * k]; A[4 * k] = A[4 * p]; A[4 * p] = t; }
                for (int i = k + 1; i < p; i++) { double t = A[3 * i + k]; A[3 * i + k] = A[3 * p + i]; A[3 * p + i] = t; }
            }
            for (int j = 0; j < k; j++) tmp[j] = A[4 * j] * A[3 * k + j];
            for (int i = k; i < 3; i++) { double t = 0; for (int j = 0; j < k; j++) t += A[3 * i + j] * tmp[j]; A[3 * i + k] -= t; }
            const double d = A[4 * k];
            if (k == 0 && !(fabs(d) > 0.0)) { tr[0] = 0; tr[1] = 1; tr[2] = 2; break; }
            if (fabs(d) > 0.0) for (int i = k + 1; i < 3; i++) A[3 * i + k] /= d;
        }
        double y[3] = {b[0], b[1], b[2]};
        for (int k = 0; k < 3; k++) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
        for (int i = 0; i < 3; i++) { double s = y[i]; for (int j = 0; j < i; j++) s -= A[3 * i + j] * y[j]; y[i] = s; }
        for (int i = 0; i < 3; i++) y[i] = fabs(A[4 * i]) > 5.562684646268003e-309 ? y[i] / A[4 * i] : 0.0;
        for (int i = 2; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < 3; j++) s -= A[3 * j + i] * y[j]; y[i] = s; }
        for (int k = 2; k >= 0; k--) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
        for (int i = 0; i < 3; i++) { x[i] = y[i]; a.out[5 + i] = x[i]; }
        for (int k = 0; k < m; k++) for (int i = 0; i < 3; i++) { a.ba2[3 * k + i] = 0.0; a.bg2[3 * k + i] = a.bgs0[i] + x[i]; }
    }
}

// ---- pivoted LDLT solve by one workgroup ----------------------------------------------------------------------------------------
// M: working matrix (ns x ld, lower triangle used), y: right-hand side in / solution out, tr / tmp: LDS scratch [ns]
__device__ void wg_ldlt_solve(int ns, double *M, int ld, double *y, int *tr, double *tmp, int *s_p) {
    const int tid = threadIdx.x;
    for (int k = 0; k < ns; k++) {
        if (tid < 64) {                                                         // pivot: first maximum of |diag| over [k, ns)
            double best = -1.0; int bi = ns;
            for (int i = k + tid; i < ns; i += 64) { const double v = fabs(M[i * ld + i]); if (v > best) { best = v; bi = i; } }
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (tid == 0) { if (bi >= ns) bi = k; *s_p = bi; tr[k] = bi; }
        }
        __syncthreads();
        const int p = *s_p;
        if (p != k) {                                                           // symmetric swap k <-> p inside the lower triangle
            for (int e = tid; e < ns; e += AL_THREADS) {
                double *u, *v;
                if (e < k) { u = &M[k * ld + e]; v = &M[p * ld + e]; }
                else if (e == k) { u = &M[k * ld + k]; v = &M[p * ld + p]; }
                else if (e < p) { u = &M[e * ld + k]; v = &M[p * ld + e]; }
                else if (e == p) continue;
                else { u = &M[e * ld + k]; v = &M[e * ld + p]; }
                const double t = *u; *u = *v; *v = t;
            }
            __syncthreads();
        }
        for (int j = tid; j < k; j += AL_THREADS) tmp[j] = M[j * ld + j] * M[k * ld + j];
        __syncthreads();
        for (int i = k + tid; i < ns; i += AL_THREADS) {                        // column k of L (and the pivot) : one thread per row
            double t = 0.0;
            for (int j = 0; j < k; j++) t += M[i * ld + j] * tmp[j];
            M[i * ld + k] -= t;
        }
        __syncthreads();
        const double d = M[k * ld + k];
        if (k == 0 && !(fabs(d) > 0.0)) { for (int j = tid; j < ns; j += AL_THREADS) tr[j] = j; __syncthreads(); break; }
        if (fabs(d) > 0.0) for (int i = k + 1 + tid; i < ns; i += AL_THREADS) M[i * ld + k] /= d;
        __syncthreads();
    }
    if (tid == 0) {
        for (int k = 0; k < ns; k++) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    }
    __syncthreads();
    for (int i = 0; i < ns; i++) {                                               // L z = P b, column-oriented: same operation order as the row loop
        const double yi = y[i];
        for (int j = i + 1 + tid; j < ns; j += AL_THREADS) y[j] -= M[j * ld + i] * yi;
        __syncthreads();
    }
    for (int i = tid; i < ns; i += AL_THREADS) { const double d = M[i * ld + i]; y[i] = fabs(d) > 5.562684646268003e-309 ? y[i] / d : 0.0; }
    __syncthreads();
    if (tid == 0) {                                                              // L^T x = z, the reference's order: s = y[i] - sum_{j > i ascending} L(j, i) x[j]
        for (int i = ns - 1; i >= 0; i--) { double s = y[i]; for (int j = i + 1; j < ns; j++) s -= M[j * ld + i] * y[j]; y[i] = s; }
        for (int k = ns - 1; k >= 0; k--) { const double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
    }
    __syncthreads();
}

// per-interval block rows of LinearAlignment (T = 4: [.., g, s]) or RefineGravity (T = 3: [.., w1, w2, s]) -> a.T[i] = tmp_A (6 x W), tmp_b
__device__ void build_blocks(const AlignArgs &a, int T, const double *g0, const double *lx, const double *ly) {
    const int m = a.n - 1, W = 6 + T;
    for (int i = threadIdx.x; i < m; i += AL_THREADS) {
        const double *Ri = a.frame_R + 9 * i, *Rj = a.frame_R + 9 * (i + 1), *Ti = a.frame_T + 3 * i, *Tj = a.frame_T + 3 * (i + 1);
        const vilf_imu_preint &p = a.pre[i];
        const double t = p.sum_dt;
        double RR[9], h[9], h2[9], RiT[9];
        m3_mulT(Ri, Rj, RR);
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { RiT[3 * r + c] = Ri[3 * c + r]; h2[3 * r + c] = RiT[3 * r + c] * t; h[3 * r + c] = ((RiT[3 * r + c] * t) * t) * 0.5; }
        const double dT[3] = {Tj[0] - Ti[0], Tj[1] - Ti[1], Tj[2] - Ti[2]};
        double cs[3], rt[3];
        m3_vec(RiT, dT, cs);
        m3_vec(RR, a.TIC, rt);
        double *tA = a.T + 66 * i, *tb = tA + 60;
        for (int e = 0; e < 66; e++) tA[e] = 0.0;
        for (int r = 0; r < 3; r++) {
            tA[r * W + r] = -t;
            tA[(3 + r) * W + r] = -1.0;
            for (int c = 0; c < 3; c++) tA[(3 + r) * W + 3 + c] = RR[3 * r + c];
            tA[r * W + W - 1] = cs[r] / 100.0;
            double bp = (p.delta_p[r] + rt[r]) - a.TIC[r], bv = p.delta_v[r];
            if (T == 4) {
                for (int c = 0; c < 3; c++) { tA[r * W + 6 + c] = h[3 * r + c]; tA[(3 + r) * W + 6 + c] = h2[3 * r + c]; }
            } else {
                tA[r * W + 6] = h[3 * r] * lx[0] + h[3 * r + 1] * lx[1] + h[3 * r + 2] * lx[2];
                tA[r * W + 7] = h[3 * r] * ly[0] + h[3 * r + 1] * ly[1] + h[3 * r + 2] * ly[2];
                tA[(3 + r) * W + 6] = h2[3 * r] * lx[0] + h2[3 * r + 1] * lx[1] + h2[3 * r + 2] * lx[2];
                tA[(3 + r) * W + 7] = h2[3 * r] * ly[0] + h2[3 * r + 1] * ly[1] + h2[3 * r + 2] * ly[2];
                bp -= h[3 * r] * g0[0] + h[3 * r + 1] * g0[1] + h[3 * r + 2] * g0[2];
                bv -= h2[3 * r] * g0[0] + h2[3 * r + 1] * g0[1] + h2[3 * r + 2] * g0[2];
            }
            tb[r] = bp; tb[3 + r] = bv;
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < m * (W * W + W); e += AL_THREADS) {           // r_A = tmp_A^T tmp_A, r_b = tmp_A^T tmp_b
        const int i = e / (W * W + W), q = e - i * (W * W + W);
        const double *tA = a.T + 66 * i, *tb = tA + 60;
        double s = 0.0;
        if (q < W * W) { const int r = q / W, c = q - r * W; for (int k = 0; k < 6; k++) s += tA[k * W + r] * tA[k * W + c]; }
        else { const int r = q - W * W; for (int k = 0; k < 6; k++) s += tA[k * W + r] * tb[k]; }
        a.RA[110 * i + q] = s;
    }
    __syncthreads();
}

// A(r, c) += sum over intervals (ascending) of their block entry; then x1000. One thread per entry of A and of b.
__device__ void accumulate(const AlignArgs &a, int T, int ns, bool keep) {
    const int m = a.n - 1, W = 6 + T, nv = 3 * a.n;
    auto range = [&](int r, int &lo, int &hi) { if (r < nv) { lo = max(r / 3 - 1, 0); hi = min(r / 3, m - 1); } else { lo = 0; hi = m - 1; } };
    auto local = [&](int r, int i) { return r < nv ? r - 3 * i : 6 + (r - nv); };
    for (int e = threadIdx.x; e < ns * ns + ns; e += AL_THREADS) {
        const bool isb = e >= ns * ns;
        const int r = isb ? e - ns * ns : e / ns, c = isb ? r : e - r * ns;
        int lo, hi, lo2, hi2;
        range(r, lo, hi); range(c, lo2, hi2);
        lo = max(lo, lo2); hi = min(hi, hi2);
        double *dst = isb ? &a.b[r] : &a.A[r * ns + c];
        double s = keep ? *dst : 0.0;
        for (int i = lo; i <= hi; i++) {
            const int lr = local(r, i), lc = local(c, i);
            if (lr < 0 || lr >= W || lc < 0 || lc >= W) continue;
            s += isb ? a.RA[110 * i + W * W + lr] : a.RA[110 * i + lr * W + lc];
        }
        *dst = s * 1000.0;
    }
    __syncthreads();
}

template <bool IN_LDS>
__global__ void __launch_bounds__(AL_THREADS) k_align_linear(AlignArgs a) {
    extern __shared__ double lds[];
    const int n = a.n, tid = threadIdx.x;
    const int ns_max = 3 * n + 4;
    double *y = lds;                                   // [ns_max]
    double *tmp = y + ns_max;                          // [ns_max]
    int *tr = reinterpret_cast<int *>(tmp + ns_max);   // [ns_max] (+ pivot slot)
    int *s_p = tr + ns_max;
    double *sv = reinterpret_cast<double *>(tr + ((ns_max + 3) & ~1));   // g0[3] lx[3] ly[3] flag
    double *M = IN_LDS ? sv + 12 : a.Wg;

    // ---- LinearAlignment
    int ns = 3 * n + 4, ld = IN_LDS ? (ns | 1) : ns + 1;
    build_blocks(a, 4, nullptr, nullptr, nullptr);
    accumulate(a, 4, ns, false);
    for (int e = tid; e < ns * ns; e += AL_THREADS) { const int r = e / ns, c = e - r * ns; M[r * ld + c] = a.A[e]; }
    for (int e = tid; e < ns; e += AL_THREADS) y[e] = a.b[e];
    __syncthreads();
    wg_ldlt_solve(ns, M, ld, y, tr, tmp, s_p);
    for (int e = tid; e < ns; e += AL_THREADS) a.out[8 + e] = y[e];
    if (tid == 0) {
        const double gx = y[ns - 4], gy = y[ns - 3], gz = y[ns - 2], s = y[ns - 1] / 100.0;
        const double gn = sqrt(gx * gx + gy * gy + gz * gz);
        a.out[2] = gx; a.out[3] = gy; a.out[4] = gz; a.out[1] = ns;
        const bool bad = fabs(gn - a.Gnorm) > 1.0 || s < 0;
        sv[9] = bad ? 1.0 : 0.0;
        if (bad) a.out[0] = 0.0;
        sv[0] = gx / gn * a.Gnorm; sv[1] = gy / gn * a.Gnorm; sv[2] = gz / gn * a.Gnorm;
    }
    __syncthreads();
    if (sv[9] != 0.0) return;

    // ---- RefineGravity: the system is NOT cleared between the 4 sweeps (A, b live outside the loop, :63-66): it accumulates, x1000 per sweep
    ns = 3 * n + 3; ld = IN_LDS ? (ns | 1) : ns + 1;
    for (int sweep = 0; sweep < 4; sweep++) {
        if (tid == 0) {                                                          // TangentBasis (:40-53)
            const double gn = sqrt(sv[0] * sv[0] + sv[1] * sv[1] + sv[2] * sv[2]);
            const double ax = sv[0] / gn, ay = sv[1] / gn, az = sv[2] / gn;
            double tx = 0, ty = 0, tz = 1;
            if (ax == 0.0 && ay == 0.0 && az == 1.0) { tx = 1; tz = 0; }
            const double dt_ = ax * tx + ay * ty + az * tz;
            const double ux = tx - ax * dt_, uy = ty - ay * dt_, uz = tz - az * dt_;
            const double un = sqrt(ux * ux + uy * uy + uz * uz);
            const double bx = ux / un, by = uy / un, bz = uz / un;
            sv[3] = bx; sv[4] = by; sv[5] = bz;
            sv[6] = ay * bz - az * by; sv[7] = az * bx - ax * bz; sv[8] = ax * by - ay * bx;
        }
        __syncthreads();
        build_blocks(a, 3, sv, sv + 3, sv + 6);
        accumulate(a, 3, ns, sweep > 0);
        for (int e = tid; e < ns * ns; e += AL_THREADS) { const int r = e / ns, c = e - r * ns; M[r * ld + c] = a.A[e]; }
        for (int e = tid; e < ns; e += AL_THREADS) y[e] = a.b[e];
        __syncthreads();
        wg_ldlt_solve(ns, M, ld, y, tr, tmp, s_p);
        if (tid == 0) {
            const double w1 = y[ns - 3], w2 = y[ns - 2];
            const double gx = sv[0] + sv[3] * w1 + sv[6] * w2, gy = sv[1] + sv[4] * w1 + sv[7] * w2, gz = sv[2] + sv[5] * w1 + sv[8] * w2;
            const double gn = sqrt(gx * gx + gy * gy + gz * gz);
            sv[0] = gx / gn * a.Gnorm; sv[1] = gy / gn * a.Gnorm; sv[2] = gz / gn * a.Gnorm;
        }
        __syncthreads();
    }
    for (int e = tid; e < ns - 1; e += AL_THREADS) a.out[8 + e] = y[e];
    if (tid == 0) {
        const double s = y[ns - 1] / 100.0;
        a.out[8 + ns - 1] = s;
        a.out[1] = ns;
        a.out[2] = sv[0]; a.out[3] = sv[1]; a.out[4] = sv[2];
        a.out[0] = s < 0.0 ? 0.0 : 1.0;
    }
}

}  // namespace

extern "C" int vilf_visual_imu_alignment(vilf_handle *h, int n, const double *frame_R, const double *frame_T, const vilf_imu_noise *nz,
                                         const double *acc_0, const double *gyr_0, const double *lin_ba, const double *lin_bg, const int *n_samples,
                                         int max_samples, const double *dt, const double *acc, const double *gyr, const double bgs0[3],
                                         double delta_bg[3], double g[3], double *x, int *n_x, vilf_imu_preint *pre_out, int *ok) {
    if (!h || n < 2 || !frame_R || !frame_T || !nz || !acc_0 || !gyr_0 || !lin_ba || !lin_bg || !n_samples || max_samples < 0 ||
        (max_samples && (!dt || !acc || !gyr)) || !bgs0 || !delta_bg || !g || !x || !n_x || !ok) return VILF_ERR_INVALID_ARGUMENT;
    const int m = n - 1;
    for (int i = 0; i < m; i++) if (n_samples[i] < 0 || n_samples[i] > max_samples) { h->err = "n_samples out of range"; return VILF_ERR_INVALID_ARGUMENT; }
    if (n > 1000) { h->err = "too many frames for the alignment"; return VILF_ERR_INVALID_ARGUMENT; }
    HIPCHECK(h, hipSetDevice(h->device));
    const size_t sm = (size_t)m, sx = sm * std::max(max_samples, 1), ns = 3 * (size_t)n + 4;
    size_t off = 0;
    auto take = [&](size_t doubles) { size_t o = off; off += (doubles + 1) & ~(size_t)1; return o; };
    const size_t o_R = take(9 * (size_t)n), o_T = take(3 * (size_t)n), o_a0 = take(3 * sm), o_g0 = take(3 * sm), o_ba = take(3 * sm), o_bg = take(3 * sm), o_bgs = take(4),
                 o_dt = take(sx), o_acc = take(3 * sx), o_gyr = take(3 * sx), o_ns = take(sm / 2 + 1), o_ba2 = take(3 * sm), o_bg2 = take(3 * sm),
                 o_pre = take(sm * (sizeof(vilf_imu_preint) / 8)), o_gy = take(12 * sm), o_Tb = take(66 * sm), o_RA = take(110 * sm), o_A = take(ns * ns), o_b = take(ns),
                 o_W = take(ns * (ns + 1)), o_out = take(8 + ns);
    if (!h->d[D_HOOK].ensure(off * 8 + 64)) { h->err = "hipMalloc failed (alignment)"; return VILF_ERR_DEVICE; }
    double *d = h->d[D_HOOK].as<double>();
    auto up = [&](size_t o, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(d + o, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess; };
    HIPCHECK(h, up(o_R, frame_R, 72 * (size_t)n)); HIPCHECK(h, up(o_T, frame_T, 24 * (size_t)n));
    HIPCHECK(h, up(o_a0, acc_0, 24 * sm)); HIPCHECK(h, up(o_g0, gyr_0, 24 * sm)); HIPCHECK(h, up(o_ba, lin_ba, 24 * sm)); HIPCHECK(h, up(o_bg, lin_bg, 24 * sm));
    HIPCHECK(h, up(o_bgs, bgs0, 24));
    HIPCHECK(h, up(o_dt, dt, 8 * sm * max_samples)); HIPCHECK(h, up(o_acc, acc, 24 * sm * max_samples)); HIPCHECK(h, up(o_gyr, gyr, 24 * sm * max_samples));
    HIPCHECK(h, up(o_ns, n_samples, 4 * sm));
    vilf_imu_preint *d_pre = reinterpret_cast<vilf_imu_preint *>(d + o_pre);
    const int *d_ns = reinterpret_cast<const int *>(d + o_ns);
    AlignArgs a;
    a.n = n; a.frame_R = d + o_R; a.frame_T = d + o_T; a.pre = d_pre; a.bgs0 = d + o_bgs;
    for (int i = 0; i < 3; i++) a.TIC[i] = h->opts.TIC[i];
    a.Gnorm = std::sqrt(h->opts.G[0] * h->opts.G[0] + h->opts.G[1] * h->opts.G[1] + h->opts.G[2] * h->opts.G[2]);
    a.gy = d + o_gy; a.ba2 = d + o_ba2; a.bg2 = d + o_bg2; a.T = d + o_Tb; a.RA = d + o_RA; a.A = d + o_A; a.b = d + o_b; a.Wg = d + o_W; a.out = d + o_out;
    hipLaunchKernelGGL(k_preintegrate, dim3((m + 63) / 64), dim3(64), 0, h->stream, m, *nz, d + o_a0, d + o_g0, d + o_ba, d + o_bg, d_ns, max_samples, d + o_dt, d + o_acc, d + o_gyr, d_pre);
    hipLaunchKernelGGL(k_align_gyro, dim3(1), dim3(64), 0, h->stream, a);
    hipLaunchKernelGGL(k_preintegrate, dim3((m + 63) / 64), dim3(64), 0, h->stream, m, *nz, d + o_a0, d + o_g0, d + o_ba2, d + o_bg2, d_ns, max_samples, d + o_dt, d + o_acc, d + o_gyr, d_pre);
    const size_t small = (3 * ns) * 8 + 32 + 12 * 8;
    if (n <= AL_LDS_FRAMES) {
        const size_t lds = small + ns * (ns | 1) * 8;
        static bool attr = false;
        if (!attr) { HIPCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(k_align_linear<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); attr = true; }
        hipLaunchKernelGGL(k_align_linear<true>, dim3(1), dim3(AL_THREADS), lds, h->stream, a);
    } else {
        hipLaunchKernelGGL(k_align_linear<false>, dim3(1), dim3(AL_THREADS), small, h->stream, a);
    }
    HIPCHECK(h, hipGetLastError());
    std::vector<double> out(8 + ns);
    HIPCHECK(h, hipMemcpyAsync(out.data(), d + o_out, out.size() * 8, hipMemcpyDeviceToHost, h->stream));
    if (pre_out) HIPCHECK(h, hipMemcpyAsync(pre_out, d_pre, sm * sizeof(vilf_imu_preint), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    *ok = out[0] != 0.0; *n_x = (int)out[1];
    for (int i = 0; i < 3; i++) { g[i] = out[2 + i]; delta_bg[i] = out[5 + i]; }
    for (int i = 0; i < *n_x; i++) x[i] = out[8 + i];
    return VILF_OK;
}

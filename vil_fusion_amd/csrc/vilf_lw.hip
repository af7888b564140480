// vilf_lw.hip — Estimator::optimization()'s solve for window sizes other than the reference's WINDOW_SIZE = 10 (BASELINE configs[4]:
// the synthetic 51-frame / ~50 k-factor stress window; reduced camera / IMU system P = 15 * 51 = 765).
// The 11-frame kernels keep a window's whole reduced system in one workgroup's LDS; a 765 x 765 system does not fit, so this path
// spreads ONE window over the device instead:
//   lw_visual / lw_imu / lw_lidar   one lane per factor: residual + Jacobians (the same device functions as the 11-frame kernels), Cauchy
//                                   corrector, J^T J / J^T r scattered with hardware fp64 atomics into the reduced blocks
//                                   Hpp (P x P), W (one row per feature over the POSE columns only: a visual factor touches two poses, the extrinsic, td and
//                                   one inverse depth — never a SpeedBias block, estimator.cpp:750-794 — so W is F x PC with PC = 6 NF [+ 6 + 1], not F x P), h_f, g_f, g_p
//   lw_scale                        Jacobi scaling
//   lw_syrk_mfma                    the Schur reduce Wn^T Wn (Wn = W / sqrt(h_f + mu d_f^2), formed on the way into LDS) as a hand-written fp64 MFMA SYRK over the
//                                   compact columns, features in start-frame order and K-chunks skipped by a tile whose columns the chunk's frame span does not reach
//                                   (a feature seen in k frames fills a 6k x 6k block, SURVEY 8(d): 2 sum (6 k_f)^2 flop, not 2 F P^2); K-split partials in their own
//                                   buffers; the right-hand side's W^T g_f / den rides along in the diagonal tiles
//   lw_schur_prep                   S = Hpp' + mu D^2 - sum of the partials (fixed order: no atomics), rhs row
//   lw_rowdot                       matrix-vector products (Hpp v, W v per feature row)
//   lw_chol_panel / lw_chol_step    own blocked Cholesky of the reduced system, one launch per 64-column block: the panel workgroups (diagonal block + the slab
//                                   below as 16 x 4 fp64 MFMA tiles in registers) beside workgroups that apply the previous column's trailing update;
//   lw_chol_back                    back substitution with the triangle staged through LDS. No rocSOLVER / rocBLAS on this path.
//   lw_tr_*                         the trust-region loop (Ceres 2.0 TrustRegionMinimizer + traditional dogleg + Jacobi scaling, the same restatement as
//                                   k_solve / k_linearize's step) on the device: all iterations enqueued at once, one read-back per solve. The host loop
//                                   further down is the path of a wall-clock limit and the fallback of a failed factorisation.
// The same path runs the solves the batched LDS kernels do not cover at ANY window size: estimate_extrinsic (Ex_Pose a variable: six more
// columns after the frame blocks) and estimate_td (ProjectionTdFactor, one more column) — estimator.cpp:701-717,765-777. For an 11-frame
// window it then also applies the slot-0 marginalization prior resident on the device (lw_prior) and writes the solved state back into the
// batch buffers, so that vilf_window_marginalize() continues from it. Other window sizes have no prior (no device marginalization).
// No sum on the plain path (Ex_Pose / td constant) depends on an order the hardware picks: fixed reduction trees, host-assigned slots summed in slot order, one writer per
// entry — a window's result is bit-reproducible, alone or in any group. The estimate_extrinsic / estimate_td variant (lw_visual_ext) and the prior (lw_prior) still add
// their pose blocks with hardware atomics: reproducible to rounding (~1e-12 relative) there.
// Groups: every kernel of this file takes an array of window descriptors (LwWin, one per window, resident on the device) and finds its window in blockIdx.z —
// ONE chain of launches solves G independent windows side by side (vilf_window_solve_group; the estimate_td / estimate_extrinsic slots of vilf_batch_solve).
// A window's chain is ~32 small dependent launches per iteration on a handful of workgroups, so a single window leaves most of the chip idle and G handles on G
// streams do not help either (the runtime multiplexes its streams onto four hardware queues: 3.1 k iterations/s at eight streams, worse with more queues);
// in one launch the G windows' workgroups simply fill more CUs. A single window is a group of one. All device memory of a group is one arena (DBuf): the
// small inputs of all windows go up in one copy, states and minimizer scalars come back in one.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>
#include <thread>
#include <atomic>
#include <mutex>
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

extern "C" __global__ void k_imu_prep(int n, const double *cov, double *work, double *imu_rec);

struct LwVis { double pi[3], pj[3]; int f, i, j, cst; };     // one ProjectionFactor: feature, start frame, observing frame
struct LwTd { double vi[2], vj[2], tdi, tdj, rowi_c, rowj_c; };     // ProjectionTdFactor constants (projection_td_factor.cpp:6-21)
// The trust-region loop on the device (see lw_tr_* below): the scalars of the minimizer. skip_* = what the launches of the current iteration may leave out
// (a finished solve, a rejected step that reuses the linear solve, an invalid step that needs no evaluation), in the order of the LW_SK_* selectors.
struct LwCtl {
    double radius, mu, alpha, step_norm, x_cost, cand_cost, x_norm, gmax, model_change, initial_cost, gg;
    int iteration, reuse, done, termination, num_successful, num_linear_solves, consecutive_invalid, fallback;
    int skip_solve, skip_quad, skip_eval, skip_jac, scaling_ready, pad_;
};
enum { LW_SK_NONE = -1, LW_SK_SOLVE = 0, LW_SK_QUAD = 1, LW_SK_EVAL = 2, LW_SK_JAC = 3 };
// One window of a group: sizes, options and device pointers (into the group's arena; the prior's into the 11-frame batch buffers of the window's slot).
// State x (and cand): pose[NF][7] | sb[NF][9] | feat[F] | ex[7] | td. Tangent / N-vectors: [15 per frame: pose 6, speed-bias 9 | ex 6 | td | F].
struct LwWin {
    int NF, F, P, N, nvis, nimu, cEx, cTd, xo, est_ex, est_td, use_lidar, pn, pnb, max_it, pad_;
    int PC, WS, nchunk, nks;                       // W: F rows of WS doubles, PC = 6 NF (+ 6 Ex)(+ 1 td) of them used: compact column c <-> column lw_fullcol(c) of the reduced system
    // pose-pose blocks of the visual factors without atomics: every run of equal frame pairs inside a chunk of lw_visual owns a SLOT of PS (12 x 12 + 12 sums, stride 160);
    // lw_assemble adds the slots of a pair / of a frame's pairs in slot order. cslot: first slot of a chunk; prt: per pair {i, j, first slot, end slot}; froff / frlist:
    // CSR per frame of (slot << 1 | 0: the frame is the slot's pair's i, 1: its j), pairs ascending, a pair's slots ascending
    double *PS; const int *cslot, *prt, *froff, *frlist; int npairs, pad3_;
    // structure: Hpp(i, j) can be non-zero only for frames |f_i - f_j| <= bandf (feature track lengths; IMU / LiDAR factors join neighbours; NF when a prior or Ex_Pose /
    // td columns couple everything), row f of W only in the compact columns fspan[2 f] .. fspan[2 f + 1] (its frames) and the Ex_Pose / td columns at the end
    int bandf, ncostv; const int *fspan;
    double *costP;                                 // cost partials, one per producing workgroup: [0, ncostv) lw_visual chunks, [ncostv, ncostv + NF) frames of lw_imu_lidar,
                                                   // [ncostv + NF] lw_prior — summed in this order by tr_cost_sum (no atomics: the cost is bit-reproducible)
    const int *fvis, *fidx;                        // CSR over the features: the factors of feature f are vis[fidx[fvis[f] .. fvis[f + 1])] (vis itself is pair-sorted)
    const int *kspan;                              // the device's feature order is by start frame (the host permutes on the way in and out); per K-chunk of SY_KB
                                                   // features: first and last compact column any of them touches
    double *rsd;                                   // 1 / sqrt(den_f) of the current linear solve
    double *SC;                                    // [SY_KS][tiles][64 x 64] K-split partials of Wn^T Wn, + [SY_KS][SY_NT x 64] of Wn^T (g_f / sqrt(den))
    double sqrt_info, cauchy_b, tr_over_row;
    LwCtl *ctl;
    double *x, *cand;
    const LwVis *vis; const LwTd *tdr; const double *imu, *lid; const unsigned char *fconst;
    double *Hpp, *W, *hf, *gp, *gf, *S, *rhs, *tmpP, *tmpF, *vec, *yf, *scal, *den;     // scal[0] = cost, [1..4] q_il, [5..7] t_il, [8..10] G
    int *info;
    double *g, *diagH, *scale, *diagonal, *gradient, *gn, *step;                                    // N each: the minimizer's vectors
    const double *pJ, *pr0, *pH0, *px0; const int *phdr, *pcol; double *pdx;                       // marginalization prior of an 11-frame window (pn = 0: none)
};
__device__ __forceinline__ bool lw_skip(const LwWin &w, int sk) { return sk >= 0 && (&w.ctl->skip_solve)[sk] != 0; }
// compact column of W (pose columns frame-major, then Ex_Pose, then td) -> column of the reduced system, and back (-1: a SpeedBias column)
__device__ __forceinline__ int lw_fullcol(const LwWin &w, int c) { const int np = 6 * w.NF; return c < np ? 15 * (c / 6) + c % 6 : (w.est_ex && c < np + 6 ? w.cEx + (c - np) : w.cTd); }
__device__ __forceinline__ int lw_compcol(const LwWin &w, int i) {
    const int np = 15 * w.NF;
    if (i < np) { const int r = i % 15; return r < 6 ? 6 * (i / 15) + r : -1; }
    if (w.est_ex && i >= w.cEx && i < w.cEx + 6) return 6 * w.NF + (i - w.cEx);
    return (w.est_td && i == w.cTd) ? 6 * w.NF + (w.est_ex ? 6 : 0) : -1;
}
__device__ __forceinline__ bool lw_inband(const LwWin &w, int i, int j) {
    const int np = 15 * w.NF;
    if (i >= np || j >= np) return true;
    const int d = i / 15 - j / 15;
    return d <= w.bandf && -d <= w.bandf;
}
#define SY_KB 32
#define SY_KS 16                                   // most K splits of the Schur reduce (partials in their own buffers, summed in fixed order by lw_schur_prep); a group uses
                                                   // LwWin::nks of them: 4 when its windows' tiles fill the chip, up to 16 for a single window (60 workgroups of ~12 chunks otherwise)

struct LwCtx {
    DBuf arena, desc;                  // the group's device memory; its LwWin array
    char *stage = nullptr;             // host image of the arena's input region (pinned: the copy of a group's factors runs at the link's rate)
    size_t stage_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    double ms[4] = {0, 0, 0, 0};       // vilf_set_profiling: factor scatter, Schur SYRK, Cholesky (potrf + potrs), other device work
    long launches[4] = {0, 0, 0, 0};
    void release() {
        arena.release(); desc.release();
        if (stage) { hipHostFree(stage); stage = nullptr; stage_cap = 0; }
        for (hipEvent_t &e : ev) if (e) { hipEventDestroy(e); e = nullptr; }
    }
};
extern "C" int vilf_get_profile_large_window(vilf_handle *h, double ms_out[4], long launches_out[4]) {
    if (!h || !ms_out || !launches_out) return VILF_ERR_INVALID_ARGUMENT;
    for (int i = 0; i < 4; i++) { ms_out[i] = h->lw ? h->lw->ms[i] : 0.0; launches_out[i] = h->lw ? h->lw->launches[i] : 0; }
    return VILF_OK;
}
void vilf_lw_release(vilf_handle *h) { if (h->lw) { h->lw->release(); delete h->lw; h->lw = nullptr; } }

namespace {
using namespace vd;


__device__ __forceinline__ void add(double *p, double v) { unsafeAtomicAdd(p, v); }

// x layout: pose[NF][7] | sb[NF][9] | feat[F]
// The factors arrive sorted by frame pair (i, j). A workgroup takes LW_CH consecutive factors: every lane evaluates one and parks its
// corrected 2 x 12 Jacobian and residual in LDS; then, per run of equal pairs inside the chunk (a handful), 156 lanes each own one entry of
// the pair's 12 x 12 block / 12-vector, sum it over the run from LDS and store it to the run's own slot of PS (host-assigned: cslot); lw_assemble
// then adds a pair's slots / the slots of a frame's pairs in slot order into Hpp and g_p — no atomics (until round 4: one atomic per entry and run, all
// workgroups of a start frame colliding on its diagonal block: 0.3 of an iteration's 2.5 ms at 32 stress windows). The per-feature terms: lw_feature_rows.
#define LW_CH 128
typedef double lw_double4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(LW_CH) void lw_visual(const LwWin *ws, int which, int jac, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int n = w.nvis, NF = w.NF;
    if ((int)blockIdx.x * LW_CH >= n) return;         // the grid is sized for the group's largest window
    const double *x = which ? w.cand : w.x, *ex = x + w.xo;
    const double sqrt_info = w.sqrt_info, cauchy_b = w.cauchy_b;
    double *PS = w.PS;
    const LwVis *vis = w.vis;
    __shared__ double s_J[LW_CH][26];                 // 24 Jacobian entries (row 0: 12, row 1: 12), r0, r1
    __shared__ double s_red[4][64];
    __shared__ unsigned long long s_bnd[2];
    __shared__ int s_pair[LW_CH + 1];
    const int tid = threadIdx.x, t = blockIdx.x * LW_CH + tid;
    double c = 0;
    int pr = -1;
    if (t < n) {
        const LwVis v = vis[t];
        pr = v.i * NF + v.j;
        const double *pi = x + 7 * v.i, *pj = x + 7 * v.j;
        double Ri[9], Rj[9], ric[9];
        q_toR(q_load(pi + 3), Ri); q_toR(q_load(pj + 3), Rj); q_toR(q_load(ex + 3), ric);
        double r[2], Ji[12], Jj[12], Jf[2];
        const double inv_dep = x[16 * NF + v.f];
        if (jac) projection_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf);
        else projection_eval<false>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf);
        double rho0, sw;
        cauchy(r[0] * r[0] + r[1] * r[1], cauchy_b, rho0, sw);
        c = 0.5 * rho0;
        if (jac) {
            double J0[12], J1[12];
#pragma unroll
            for (int k = 0; k < 6; k++) { J0[k] = sw * Ji[k]; J1[k] = sw * Ji[6 + k]; J0[6 + k] = sw * Jj[k]; J1[6 + k] = sw * Jj[6 + k]; }
            const double r0 = sw * r[0], r1 = sw * r[1];
#pragma unroll
            for (int k = 0; k < 12; k++) { s_J[tid][k] = J0[k]; s_J[tid][12 + k] = J1[k]; }
            s_J[tid][24] = r0; s_J[tid][25] = r1;           // the feature's row of W, h_f, g_f: lw_feature_rows (feature-major, no atomics)
        }
    }
    if (jac) {
        // X = the run's rows [J0 | r0], [J1 | r1] (two per factor, 13 columns): X^T X on the matrix cores. v_mfma_f64_16x16x4_f64 takes four rows per instruction with ONE
        // operand register serving as A and as B (lane l holds X[4 s + l / 16][l % 16]); the two waves take alternate K steps of a run, wave 1 hands its accumulators to
        // wave 0 through LDS, wave 0 stores the run's slot. (Until round 4: a lane per entry summing over the run from LDS — 256 dependent LDS round trips per lane, 32 us
        // of a workgroup's life.) The run boundaries come from two ballots instead of a serial walk over s_pair.
        s_pair[tid] = pr;
        if (tid == 0) s_pair[LW_CH] = -2;
        __syncthreads();
        const int cnt = min(LW_CH, n - blockIdx.x * LW_CH), slot0 = w.cslot[blockIdx.x];
        const int lane = tid & 63, wave = tid >> 6, c16 = lane & 15, g4 = lane >> 4;
        const unsigned long long bm = __ballot(tid < cnt && s_pair[tid] != s_pair[tid + 1]);      // bit t: a run ends at factor t of this wave's half
        if (lane == 0) s_bnd[wave] = bm;
        __syncthreads();
        const unsigned long long m0 = s_bnd[0], m1 = s_bnd[1];
        int run = 0;
        for (int b0 = 0; b0 < cnt;) {                 // runs of equal pairs (uniform loop)
            int b1;
            { const unsigned long long lo = b0 < 64 ? (m0 >> b0) : 0ull; if (lo) b1 = b0 + __ffsll((long long)lo); else { const int s0 = b0 < 64 ? 0 : b0 - 64; b1 = 64 + s0 + __ffsll((long long)(m1 >> s0)); } }
            lw_double4 acc = lw_double4{0, 0, 0, 0};
            const int nst = (b1 - b0 + 1) >> 1;        // K steps of four rows = two factors
            for (int st = wave; st < nst; st += 2) {
                const int q = b0 + 2 * st + (g4 >> 1), hrow = g4 & 1;
                double xv = 0.0;
                if (q < b1 && c16 < 13) xv = s_J[q][c16 < 12 ? 12 * hrow + c16 : 24 + hrow];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xv, xv, acc, 0, 0, 0);
            }
            if (wave == 1) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) s_red[q4][lane] = acc[q4];
            }
            __syncthreads();
            if (wave == 0) {
                double *slot = PS + (size_t)(slot0 + run) * 160;       // this run's own slot: plain stores, lw_assemble adds the slots up in a fixed order
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) {
                    const int row = g4 + 4 * q4;                        // entry (row, c16) of X^T X
                    const double v = acc[q4] + s_red[q4][lane];
                    if (row < 12 && c16 < 12) slot[12 * row + c16] = v;
                    else if (row < 12 && c16 == 12) slot[144 + row] = v;
                }
            }
            __syncthreads();
            run++;
            b0 = b1;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __syncthreads();                                  // s_red is free again
    if ((tid & 63) == 0) s_red[0][tid >> 6] = c;
    __syncthreads();
    if (tid == 0) w.costP[blockIdx.x] = s_red[0][0] + s_red[0][1];       // this workgroup's own slot: summed in slot order by tr_cost_sum
}

// The same chunked scatter with Ex_Pose and / or td as variables (estimate_extrinsic / estimate_td): 19 Jacobian columns per factor row
// [pose_i 6 | pose_j 6 | Ex 6 | td 1]; a column whose block is constant maps to -1 and is skipped. x = pose | sb | feat | ex[7] | td.
__device__ __forceinline__ int lw_col(int a, int ci, int cj, int cEx, int cTd) { return a < 6 ? ci + a : (a < 12 ? cj + a - 6 : (a < 18 ? (cEx < 0 ? -1 : cEx + a - 12) : cTd)); }
__global__ __launch_bounds__(LW_CH) void lw_visual_ext(const LwWin *ws, int which, int jac, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int n = w.nvis, NF = w.NF, F = w.F, P = w.P, cEx = w.cEx, cTd = w.cTd;
    if ((int)blockIdx.x * LW_CH >= n) return;
    const double *x = which ? w.cand : w.x;
    const double sqrt_info = w.sqrt_info, cauchy_b = w.cauchy_b, tr_over_row = w.tr_over_row;
    double *Hpp = w.Hpp, *gp = w.gp;
    const LwVis *vis = w.vis; const LwTd *tdr = w.tdr;
    __shared__ double s_J[LW_CH][41];                 // row 0: 19, row 1: 19, r0, r1 (+ 1 pad)
    __shared__ int s_pair[LW_CH + 1];
    const int tid = threadIdx.x, t = blockIdx.x * LW_CH + tid;
    const double *ex = x + 16 * (size_t)NF + F;
    double c = 0;
    int pr = -1;
    if (t < n) {
        const LwVis v = vis[t];
        pr = v.i * NF + v.j;
        const double *pi = x + 7 * v.i, *pj = x + 7 * v.j;
        double Ri[9], Rj[9], ric[9];
        q_toR(q_load(pi + 3), Ri); q_toR(q_load(pj + 3), Rj); q_toR(q_load(ex + 3), ric);
        double r[2], Ji[12], Jj[12], Jf[2], Jex[12], Jtd[2] = {0.0, 0.0};
        const double inv_dep = x[16 * NF + v.f];
        if (cTd >= 0) {
            const LwTd q = tdr[t];
            if (jac) projection_td_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, q.vi, q.vj, ex[7], q.tdi, q.tdj, q.rowi_c, q.rowj_c, tr_over_row, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex, Jtd);
            else projection_td_eval<false>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, q.vi, q.vj, ex[7], q.tdi, q.tdj, q.rowi_c, q.rowj_c, tr_over_row, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex, Jtd);
        } else {
            if (jac) projection_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex);
            else projection_eval<false>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex);
        }
        double rho0, sw;
        cauchy(r[0] * r[0] + r[1] * r[1], cauchy_b, rho0, sw);
        c = 0.5 * rho0;
        if (jac) {
            double J0[19], J1[19];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                J0[k] = sw * Ji[k]; J1[k] = sw * Ji[6 + k]; J0[6 + k] = sw * Jj[k]; J1[6 + k] = sw * Jj[6 + k];
                J0[12 + k] = cEx < 0 ? 0.0 : sw * Jex[k]; J1[12 + k] = cEx < 0 ? 0.0 : sw * Jex[6 + k];
            }
            J0[18] = cTd < 0 ? 0.0 : sw * Jtd[0]; J1[18] = cTd < 0 ? 0.0 : sw * Jtd[1];
            const double r0 = sw * r[0], r1 = sw * r[1];
#pragma unroll
            for (int k = 0; k < 19; k++) { s_J[tid][k] = J0[k]; s_J[tid][19 + k] = J1[k]; }
            s_J[tid][38] = r0; s_J[tid][39] = r1;           // the feature's row of W, h_f, g_f: lw_feature_rows
        }
    }
    if (jac) {
        // X^T X of a frame pair's factors on the matrix cores (round 5): X = the run's rows [J (19 columns) | r], two per factor, 20 -> 32 columns = the tiles (0,0),
        // (1,0), (1,1) exactly as k_marg_prepare forms its pair products; a wave takes every other run of the chunk and adds its tiles to Hpp / g_p with atomics (as
        // before: the Ex_Pose and td blocks are shared by all pairs). Until then a thread per ENTRY summed over the run's factors from LDS: 361 serial sums per run,
        // twelve runs per chunk — 93 us per linearisation of 64 windows against 8 us for the evaluation alone.
        __shared__ int s_run[LW_CH + 1], s_nrun;
        __shared__ double s_zero;
        s_pair[tid] = pr;
        if (tid == 0) { s_pair[LW_CH] = -2; s_zero = 0.0; }
        __syncthreads();
        const int cnt = min(LW_CH, n - blockIdx.x * LW_CH), lane = tid & 63, wave = tid >> 6;
        {
            const bool head = tid < cnt && (tid == 0 || s_pair[tid] != s_pair[tid - 1]);
            const unsigned long long hb = __ballot(head);
            if (lane == 0 && wave == 0) s_run[LW_CH] = __popcll(hb);           // heads of wave 0 (a scratch slot: rewritten below)
            __syncthreads();
            const int before = (wave ? s_run[LW_CH] : 0) + __popcll(hb & ((1ULL << lane) - 1ULL));
            __syncthreads();
            if (head) s_run[before] = tid;
            if (tid == LW_CH - 1) s_nrun = before + (head ? 1 : 0);
            __syncthreads();
        }
        const int nrun = s_nrun, l16 = lane & 15, l4 = lane >> 4, sub = l4 & 1;
        const int comp0 = 19 * sub + l16, comp1 = l16 < 3 ? 19 * sub + 16 + l16 : (l16 == 3 ? 38 + sub : -1);
        for (int ri = wave; ri < nrun; ri += LW_CH / 64) {
            const int b0 = s_run[ri], b1 = ri + 1 < nrun ? s_run[ri + 1] : cnt, len = b1 - b0, pp = s_pair[b0];
            const int fi = pp / NF, fj = pp - fi * NF, ci = 15 * fi, cj = 15 * fj;
            lw_double4 T00 = {0, 0, 0, 0}, T10 = {0, 0, 0, 0}, T11 = {0, 0, 0, 0};
            for (int q0 = 0; q0 < len; q0 += 8) {                   // four k-steps (eight factors) per trip: the eight LDS reads first, then the twelve MFMAs
                double a0[4], a1[4];
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {                    // (rows past the run and the columns 20 .. 31 read a zero: the select sits on the address)
                    const int q = q0 + 2 * k4 + (l4 >> 1);
                    a0[k4] = *((q < len) ? &s_J[b0 + q][comp0] : &s_zero);
                    a1[k4] = *((q < len && comp1 >= 0) ? &s_J[b0 + q][comp1] : &s_zero);
                }
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {
                    T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k4], a0[k4], T00, 0, 0, 0);
                    T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k4], a0[k4], T10, 0, 0, 0);
                    T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k4], a1[k4], T11, 0, 0, 0);
                }
            }
            // tile entry (u, v) = sum over the run's rows of X[.][u] X[.][v]: element q of a tile: u = l4 + 4 q (+ 16), v = l16 (+ 16); column 19 = r
            const int cb0 = lw_col(l16, ci, cj, cEx, cTd);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int ca = lw_col(l4 + 4 * q, ci, cj, cEx, cTd);
                if (ca >= 0 && cb0 >= 0) add(Hpp + (size_t)ca * P + cb0, T00[q]);
            }
            {
                const int u1 = 16 + l4;                              // rows 16 .. 19 of the product live in element 0 of the tiles (1,0) and (1,1)
                if (u1 < 19) {
                    const int ca = lw_col(u1, ci, cj, cEx, cTd);
                    if (ca >= 0 && cb0 >= 0) { add(Hpp + (size_t)ca * P + cb0, T10[0]); add(Hpp + (size_t)cb0 * P + ca, T10[0]); }
                    if (l16 < 3) { const int cb1 = lw_col(16 + l16, ci, cj, cEx, cTd); if (ca >= 0 && cb1 >= 0) add(Hpp + (size_t)ca * P + cb1, T11[0]); }
                } else {                                             // u1 = 19: the right-hand side row: J^T r
                    if (cb0 >= 0) add(gp + cb0, T10[0]);
                    if (l16 < 3) { const int cb1 = lw_col(16 + l16, ci, cj, cEx, cTd); if (cb1 >= 0) add(gp + cb1, T11[0]); }
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    __syncthreads();
    double *s_cw = &s_J[0][0];                        // the Jacobian rows are done with
    if ((tid & 63) == 0) s_cw[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) w.costP[blockIdx.x] = s_cw[0] + s_cw[1];
}
// Hpp's pose-pose blocks and g_p's pose parts of the visual factors from the slots of lw_visual, every entry by ONE thread in slot order (bit-reproducible, plain
// stores into the cleared arrays; the IMU / LiDAR / prior kernels add on top afterwards). Workgroup b < NF: frame b's diagonal block = the i-i parts of the
// pairs (b, *) and the j-j parts of the pairs (*, b), and g_p[15 b ..]; workgroup NF + p: pair p's off-diagonal block and its transpose.
__global__ __launch_bounds__(256) void lw_assemble(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;
    const int NF = w.NF, P = w.P, tid = threadIdx.x, b = blockIdx.x;
    const double *PS = w.PS; const int *prt = w.prt;
    __shared__ double s_part[6][42];
    __shared__ int s_sl[256];                          // a round of the frame's list (slot << 1 | kind), in list order
    if (b < NF) {
        // a frame's list is a few dozen to a few hundred slots (slot << 1 | kind: 0 = the frame is the pair's i, 1 = its j; built by the host in pair order): six lanes
        // per entry take every sixth slot of a round of 256, the six partial sums are added in lane order — the same order every run. The list comes through LDS
        // with one coalesced read and a lane's loads from PS are issued sixteen at a time: until round 4 thread 0 composed the list from the pair table (two
        // dependent loads per pair) and every slot was its own round trip — 34 us of a single stress window's iteration.
        const int u0 = w.froff[b], u1 = w.froff[b + 1];
        const int e = tid % 42, part = tid / 42, a = e < 36 ? e / 6 : e - 36, c = e < 36 ? e % 6 : 0;
        double acc = 0;
        for (int base = u0; base < u1; base += 256) {
            const int n = min(256, u1 - base);
            __syncthreads();
            if (tid < n) s_sl[tid] = w.frlist[base + tid];
            __syncthreads();
            if (tid < 252)
                for (int k0 = part; k0 < n; k0 += 6 * 16) {
                    double v[16];
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        const int sk2 = s_sl[min(k0 + 6 * q, n - 1)], kind = sk2 & 1, o = e < 36 ? (6 * kind + a) * 12 + 6 * kind + c : 144 + 6 * kind + a;
                        v[q] = PS[(size_t)(sk2 >> 1) * 160 + o];
                    }
#pragma unroll
                    for (int q = 0; q < 16; q++) if (k0 + 6 * q < n) acc += v[q];
                }
        }
        if (tid < 252) s_part[part][e] = acc;
        __syncthreads();
        if (tid < 42) {
            const int a = tid < 36 ? tid / 6 : tid - 36, c = tid < 36 ? tid % 6 : 0;
            const double acc = ((s_part[0][tid] + s_part[1][tid]) + (s_part[2][tid] + s_part[3][tid])) + (s_part[4][tid] + s_part[5][tid]);
            if (tid < 36) w.Hpp[(size_t)(15 * b + a) * P + 15 * b + c] = acc; else w.gp[15 * b + a] = acc;
        }
    } else {
        const int p = b - NF;
        if (p >= w.npairs || tid >= 36) return;
        const int a = tid / 6, c = tid % 6, i = prt[4 * p], j = prt[4 * p + 1];
        double acc = 0;
        for (int sl = prt[4 * p + 2]; sl < prt[4 * p + 3]; sl++) acc += PS[(size_t)sl * 160 + a * 12 + 6 + c];
        w.Hpp[(size_t)(15 * i + a) * P + 15 * j + c] = acc;
        w.Hpp[(size_t)(15 * j + c) * P + 15 * i + a] = acc;
    }
}
// The per-feature terms of a linearisation — the feature's row of W (pose-feature blocks of J^T J over the compact columns), h_f, g_f — FEATURE-major and without
// atomics: one wave per feature, a lane per factor of it (fidx lists them; blocks of 64 for longer tracks). A factor's pose_j block lands in columns of its own
// (frame j observes the feature once), the pose_i block (every factor shares the start frame), the Ex_Pose / td blocks, h_f and g_f are wave sums in a fixed tree.
// The row is composed in LDS (zeros elsewhere) and written out whole: nothing to clear beforehand, one coalesced row store, bit-reproducible. Until round 4 these
// were 14 hardware atomics per factor from the pair-major kernel, 64 different rows per wave instruction: 0.6 of the 2.5 ms of an iteration of 32 stress windows.
// GRP = lanes per feature: 64 (a wave per feature, blocks of 64 factors for longer tracks) or 16 — four features per wave, for windows of up to 17 frames (a track
// has at most NF - 1 factors): an 11-frame window's features have ~7 factors, a wave per feature ran with a tenth of its lanes (round 5; the estimate_td batches). The
// group sums are the first four stages of the wave sum's tree (DPP inside a row of 16 lanes): the same bits as the 64-lane sum of a feature whose other rows are zero.
template <int GRP>
__device__ __forceinline__ double lw_group_sum(double v) {
    if (GRP == 64) return vilf_wave_sum64(v);
    v += vilf_dpp_f64<0xB1>(v); v += vilf_dpp_f64<0x4E>(v); v += vilf_dpp_f64<0x141>(v); v += vilf_dpp_f64<0x140>(v);
    return v;
}
template <bool EXT, int GRP>
__global__ __launch_bounds__(256, 2) void lw_feature_rows(const LwWin *ws, int which, int sk) {      // (two workgroups per CU: the estimate_td form would take 258 registers and run one)
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;
    extern __shared__ double s_rows[];
    constexpr int FPW = 64 / GRP;                      // features per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane / GRP, l = lane % GRP, f0 = (blockIdx.x * 4 + wave) * FPW, f = f0 + g, NF = w.NF, WS = w.WS, PC = w.PC;
    if (f0 >= w.F) return;                            // whole waves leave: no workgroup barrier below
    double *rows = s_rows + (size_t)wave * FPW * WS, *row = rows + (size_t)g * WS;
    for (int c = lane; c < FPW * WS; c += 64) rows[c] = 0.0;
    __builtin_amdgcn_wave_barrier();
    double hacc = 0, gacc = 0;
    const bool fok = f < w.F && !w.fconst[f];
    {
        const double *x = which ? w.cand : w.x, *ex = x + w.xo;
        const double sqrt_info = w.sqrt_info, cauchy_b = w.cauchy_b;
        const int n0 = fok ? w.fvis[f] : 0, n1 = fok ? w.fvis[f + 1] : 0;
        const double inv_dep = x[16 * NF + min(f, w.F - 1)];
        double ric[9];
        q_toR(q_load(ex + 3), ric);
        double iacc[6] = {0, 0, 0, 0, 0, 0}, eacc[7] = {0, 0, 0, 0, 0, 0, 0};
        int fi = 0;
        const int nblk = (GRP == 64) ? (n1 - n0 + 63) / 64 : 1;      // (GRP 16: the host chose it because no track of the group is longer)
        for (int bk = 0; bk < nblk; bk++) {
            const int t = n0 + bk * GRP + l;
            const bool act = t < n1;
            double wi_[6] = {0, 0, 0, 0, 0, 0}, we_[7] = {0, 0, 0, 0, 0, 0, 0}, hh = 0, gg = 0;
            if (act) {
                const int k = w.fidx[t];
                const LwVis v = w.vis[k];
                fi = v.i;
                const double *pi = x + 7 * v.i, *pj = x + 7 * v.j;
                double Ri[9], Rj[9];
                q_toR(q_load(pi + 3), Ri); q_toR(q_load(pj + 3), Rj);
                double r[2], Ji[12], Jj[12], Jf[2], Jex[12], Jtd[2] = {0.0, 0.0};
                if (EXT) {
                    if (w.cTd >= 0) { const LwTd q = w.tdr[k]; projection_td_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, q.vi, q.vj, ex[7], q.tdi, q.tdj, q.rowi_c, q.rowj_c, w.tr_over_row, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex, Jtd); }
                    else projection_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf, Jex);
                } else projection_eval<true>(pi, Ri, pj, Rj, ric, ex, v.pi, v.pj, inv_dep, sqrt_info, r, Ji, Jj, Jf);
                double rho0, sw;
                cauchy(r[0] * r[0] + r[1] * r[1], cauchy_b, rho0, sw);
                const double f0_ = sw * sw * Jf[0], f1_ = sw * sw * Jf[1];          // (sw J)^T (sw Jf): both factors of the product carry the corrector's sqrt(rho')
                hh = f0_ * Jf[0] + f1_ * Jf[1];
                gg = f0_ * r[0] + f1_ * r[1];
#pragma unroll
                for (int a = 0; a < 6; a++) { wi_[a] = Ji[a] * f0_ + Ji[6 + a] * f1_; row[6 * v.j + a] = Jj[a] * f0_ + Jj[6 + a] * f1_; }
                if (EXT) {
                    if (w.cEx >= 0) {
#pragma unroll
                        for (int a = 0; a < 6; a++) we_[a] = Jex[a] * f0_ + Jex[6 + a] * f1_;
                    }
                    if (w.cTd >= 0) we_[6] = Jtd[0] * f0_ + Jtd[1] * f1_;
                }
            }
            hacc += lw_group_sum<GRP>(hh); gacc += lw_group_sum<GRP>(gg);
#pragma unroll
            for (int a = 0; a < 6; a++) iacc[a] += lw_group_sum<GRP>(wi_[a]);
            if (EXT) {
#pragma unroll
                for (int a = 0; a < 7; a++) eacc[a] += lw_group_sum<GRP>(we_[a]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (l == 0 && n1 > n0) {                      // the start frame: every factor of the feature has it, and the group's first lane is active in every block
#pragma unroll
            for (int a = 0; a < 6; a++) row[6 * fi + a] = iacc[a];
            if (EXT) {
                if (w.cEx >= 0) {
#pragma unroll
                    for (int a = 0; a < 6; a++) row[6 * NF + a] = eacc[a];
                }
                if (w.cTd >= 0) row[PC - 1] = eacc[6];
            }
        }
    }
    if (l == 0 && f < w.F) { w.hf[f] = hacc; w.gf[f] = gacc; }
    __builtin_amdgcn_wave_barrier();
    double *dst = w.W + (size_t)f0 * WS;
    const int cnt = min(FPW, w.F - f0) * WS;
    for (int c = lane; c < cnt; c += 64) dst[c] = rows[c];
}
// MarginalizationFactor (marginalization_factor.cpp:333-381) of an 11-frame window: r = r0 + J0 dx with dx from lw_tr_* / the host (n <= 96 entries),
// J0^T J0 from k_prior_prep; pcol maps a prior column to its column of the reduced system (-1: block constant in this solve)
__device__ __forceinline__ void lw_prior_body(const LwWin &w, int jac) {
    if (!w.pn) return;
    const int n = w.pn, P = w.P;
    const double *J = w.pJ, *r0 = w.pr0, *H0 = w.pH0, *dx = w.pdx; const int *pcol = w.pcol;
    double *Hpp = w.Hpp, *gp = w.gp;
    __shared__ double s_r[VB_PRIOR_LD], s_dx[VB_PRIOR_LD];
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) s_dx[i] = dx[i];
    __syncthreads();
    for (int k = tid; k < n; k += 256) { double s = r0[k]; for (int i = 0; i < n; i++) s += J[(size_t)k * n + i] * s_dx[i]; s_r[k] = s; }
    __syncthreads();
    if (tid == 0) { double c = 0; for (int k = 0; k < n; k++) c += 0.5 * s_r[k] * s_r[k]; w.costP[w.ncostv + w.NF] = c; }
    if (!jac) return;
    for (int i = tid; i < n; i += 256) { if (pcol[i] < 0) continue; double s = 0; for (int k = 0; k < n; k++) s += J[(size_t)k * n + i] * s_r[k]; add(gp + pcol[i], s); }
    for (int e = tid; e < n * n; e += 256) { const int i = e / n, j = e - i * n; if (pcol[i] >= 0 && pcol[j] >= 0) add(Hpp + (size_t)pcol[i] * P + pcol[j], H0[(size_t)i * VB_PRIOR_LD + j]); }
}
__global__ __launch_bounds__(256) void lw_prior(const LwWin *ws, int jac, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;
    lw_prior_body(w, jac);
}
// IMUFactor between frames k, k + 1 (rec[287] = 0: skipped, sum_dt > 10 s) and the LiDAR between-factor of the same pair, FRAME-major: workgroup f owns the rows of
// frame f of Hpp and g_p and adds to them what the two factors next to the frame contribute — "prev" (pair f - 1, f: the frame is its j side) and "next" (pair f, f + 1:
// its i side) — prev before next, IMU before LiDAR, with plain read-modify-writes: every entry has ONE writer and a fixed order, so the linearisation is bit-reproducible
// (until round 4: a workgroup per pair and an atomic per entry; the diagonal blocks took atomics from two workgroups in either order). Each factor is evaluated by
// both of its frames' workgroups (lane 0 of a wave each, the four evaluations side by side); the products with sqrt_info (15 x 15 upper triangular times 15 x 31) and
// the J^T [J r] entries are one lane per entry. The pair's cost goes to the slot of the frame it is "next" of.
// ATOMIC: the adds into Hpp / g_p as atomics — the estimate_extrinsic / estimate_td solves run this body in ONE launch with the prior's (lw_imu_lidar_prior: their
// accumulation is not ordered anyway, lw_visual_ext and lw_prior add with atomics), which hides the prior's 30 us behind the IMU factors' 35 (round 5)
template <bool ATOMIC>
__device__ __forceinline__ void lw_acc(double *p, double v) { if (ATOMIC) add(p, v); else *p += v; }
template <bool ATOMIC>
__device__ __forceinline__ void lw_imu_lidar_body(const LwWin &w, int which, int jac, int f) {
    const int tid = threadIdx.x, NF = w.NF, P = w.P, use_lidar = w.use_lidar, nimu = w.nimu;
    if (f >= NF) return;
    const double *x = which ? w.cand : w.x, *imu_rec = w.imu, *lid = w.lid, *G = w.scal + 8, *qil = w.scal + 1, *til = w.scal + 5;
    double *Hpp = w.Hpp, *gp = w.gp;
    __shared__ double s_r[2][16], s_J[2][15 * 30], s_rw[2][16], s_lr[2][8], s_lJi[2][36], s_lJj[2][36], s_jo[2][480];
    // slot 0: prev (k = f - 1), slot 1: next (k = f)
    const bool has[2] = {f >= 1 && f - 1 < nimu && (jac != 0), f < nimu};           // the cost-only pass needs each pair once: as somebody's "next"
    bool imu_on[2];
#pragma unroll
    for (int sl = 0; sl < 2; sl++) { const int k = f - 1 + sl; imu_on[sl] = has[sl] && imu_rec[(size_t)k * IMU_REC + 287] != 0.0; }
    {
        const int sl = (tid >> 6) & 1, k = f - 1 + sl;
        if (has[sl]) {
            const double *pi = x + 7 * k, *pj = x + 7 * (k + 1), *sbi = x + 7 * NF + 9 * k, *sbj = sbi + 9;
            if ((tid == 0 || tid == 64) && imu_on[sl]) {
                const double *rec = imu_rec + (size_t)k * IMU_REC;
                if (jac) imu_raw_eval<true, 30, true>(pi, sbi, pj, sbj, rec, G, s_r[sl], s_J[sl]); else imu_raw_eval<false, 30, true>(pi, sbi, pj, sbj, rec, G, s_r[sl], s_J[sl]);
            }
            if ((tid == 128 || tid == 192) && use_lidar) {
                const double *lc = lid + 7 * (size_t)k;
                if (jac) lidar_between_eval<true>(pi, pj, q_load(qil), til, q_load(lc), lc + 4, s_lr[sl], s_lJi[sl], s_lJj[sl]); else lidar_between_eval<false>(pi, pj, q_load(qil), til, q_load(lc), lc + 4, s_lr[sl], s_lJi[sl], s_lJj[sl]);
            }
        }
    }
    __syncthreads();
    for (int e = tid; e < 2 * 465; e += 256) {               // weighted Jacobian (15 x 30) and residual (15) of both factors
        const int sl = e >= 465 ? 1 : 0, e1 = e - 465 * sl;
        if (!imu_on[sl] || (!jac && e1 >= 15)) continue;
        const double *S = imu_rec + (size_t)(f - 1 + sl) * IMU_REC + IMU_SQRT;   // upper-triangular sqrt_info (15 x 15, row-major)
        const int a = jac ? e1 / 31 : e1, col = jac ? e1 - 31 * a : 30;
        double sum = 0;
        for (int m = a; m < 15; m++) sum += S[15 * a + m] * (col < 30 ? s_J[sl][30 * m + col] : s_r[sl][m]);
        if (col < 30) s_jo[sl][30 * a + col] = sum; else { s_rw[sl][a] = sum; if (jac) s_jo[sl][450 + a] = sum; }
    }
    __syncthreads();
    if (jac)
        for (int e = tid; e < 690; e += 256) {               // row a of frame f against: its own block (b < 15), frame f + 1 (15 ..), frame f - 1 (30 ..), the gradient (e >= 675)
            const int a = e < 675 ? (e % 225) / 15 : e - 675, b = e < 675 ? e % 15 : 0, blk = e < 675 ? e / 225 : 3;
            double v0 = 0, v1 = 0;                           // prev's and next's contributions
            if (blk == 0 || blk == 3) {
                if (imu_on[0]) { double t = 0; for (int m = 0; m < 15; m++) t += s_jo[0][30 * m + 15 + a] * (blk == 0 ? s_jo[0][30 * m + 15 + b] : s_jo[0][450 + m]); v0 = t; }
                if (imu_on[1]) { double t = 0; for (int m = 0; m < 15; m++) t += s_jo[1][30 * m + a] * (blk == 0 ? s_jo[1][30 * m + b] : s_jo[1][450 + m]); v1 = t; }
                if (use_lidar && a < 6 && b < 6) {           // LiDAR between-factor: unweighted Jacobian, weighted residual (the reference's quirk)
                    if (has[0]) { double t = 0; for (int m = 0; m < 6; m++) t += s_lJj[0][6 * m + a] * (blk == 0 ? s_lJj[0][6 * m + b] : s_lr[0][m]); v0 += t; }
                    if (has[1]) { double t = 0; for (int m = 0; m < 6; m++) t += s_lJi[1][6 * m + a] * (blk == 0 ? s_lJi[1][6 * m + b] : s_lr[1][m]); v1 += t; }
                }
                if (!has[0] && !has[1]) continue;
                if (blk == 0) lw_acc<ATOMIC>(Hpp + (size_t)(15 * f + a) * P + 15 * f + b, v0 + v1); else lw_acc<ATOMIC>(gp + 15 * f + a, v0 + v1);
            } else if (blk == 1) {                           // (f, f + 1): next's i-j block
                if (!has[1]) continue;
                if (imu_on[1]) { double t = 0; for (int m = 0; m < 15; m++) t += s_jo[1][30 * m + a] * s_jo[1][30 * m + 15 + b]; v1 = t; }
                if (use_lidar && a < 6 && b < 6) { double t = 0; for (int m = 0; m < 6; m++) t += s_lJi[1][6 * m + a] * s_lJj[1][6 * m + b]; v1 += t; }
                lw_acc<ATOMIC>(Hpp + (size_t)(15 * f + a) * P + 15 * (f + 1) + b, v1);
            } else {                                         // (f, f - 1): prev's j-i block
                if (!has[0]) continue;
                if (imu_on[0]) { double t = 0; for (int m = 0; m < 15; m++) t += s_jo[0][30 * m + 15 + a] * s_jo[0][30 * m + b]; v0 = t; }
                if (use_lidar && a < 6 && b < 6) { double t = 0; for (int m = 0; m < 6; m++) t += s_lJj[0][6 * m + a] * s_lJi[0][6 * m + b]; v0 += t; }
                lw_acc<ATOMIC>(Hpp + (size_t)(15 * f + a) * P + 15 * (f - 1) + b, v0);
            }
        }
    if (tid == 0) {                                          // the pair (f, f + 1)'s cost
        double c = 0;
        if (imu_on[1]) for (int a = 0; a < 15; a++) c += 0.5 * s_rw[1][a] * s_rw[1][a];
        if (use_lidar && has[1]) for (int a = 0; a < 6; a++) c += 0.5 * s_lr[1][a] * s_lr[1][a];
        w.costP[w.ncostv + f] = c;
    }
}
__global__ __launch_bounds__(256) void lw_imu_lidar(const LwWin *ws, int which, int jac, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    lw_imu_lidar_body<false>(w, which, jac, (int)blockIdx.x);
}
// workgroups 0 .. maxNF - 1: the IMU / LiDAR factors of a frame (atomic adds); the last workgroup: the window's prior
__global__ __launch_bounds__(256) void lw_imu_lidar_prior(const LwWin *ws, int which, int jac, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;
    if (blockIdx.x + 1 == gridDim.x) lw_prior_body(w, jac);
    else lw_imu_lidar_body<true>(w, which, jac, (int)blockIdx.x);
}
// zero the accumulation targets of one linearisation that are summed into with atomics (Hpp, g_p) and the cost: four entries per thread. W, h_f and g_f are
// written whole by lw_feature_rows.
__global__ __launch_bounds__(256) void lw_clear(const LwWin *ws, int full, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int P = w.P, i = blockIdx.x, tid = threadIdx.x;
    if (i > P) return;
    if (i == P) { for (int c = tid; c < P; c += 256) w.gp[c] = 0.0; if (tid == 0) w.scal[0] = 0.0; return; }
    // row i of Hpp: outside the band nothing is ever added — zero since the solve's first (full) clear
    double *row = w.Hpp + (size_t)i * P;
    const int np = 15 * w.NF, fr = i / 15;
    const int c0 = (full || i >= np) ? 0 : 15 * max(0, fr - w.bandf), c1 = (full || i >= np) ? np : min(np, 15 * (fr + w.bandf + 1));
    for (int c = c0 + tid; c < c1; c += 256) row[c] = 0.0;
    for (int c = np + tid; c < P; c += 256) row[c] = 0.0;
}
// Jacobi scaling in place: Hpp(i, j) *= s_i s_j, W(f, c) *= s_f s_c, h_f *= s_f^2, g *= s. s = [P pose/speed-bias entries | F features] (src 0: the minimizer's
// scale vector, 1: vec — the host loop uploads it there)
__global__ __launch_bounds__(256) void lw_scale(const LwWin *ws, int src, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int P = w.P, F = w.F, b = blockIdx.x, tid = threadIdx.x;
    const double *s = src ? w.vec : w.scale;
    if (b < P) {                                       // row b of Hpp: its band (zeros elsewhere), and g_p[b]
        double *row = w.Hpp + (size_t)b * P;
        const double si = s[b];
        const int np = 15 * w.NF, fr = b / 15;
        const int c0 = b >= np ? 0 : 15 * max(0, fr - w.bandf), c1 = b >= np ? np : min(np, 15 * (fr + w.bandf + 1));
        for (int c = c0 + tid; c < c1; c += 256) row[c] *= si * s[c];
        for (int c = np + tid; c < P; c += 256) row[c] *= si * s[c];
        if (tid == 0) w.gp[b] *= si;
        return;
    }
    const int f = (b - P) * 4 + (tid >> 6), lane = tid & 63;       // a wave per row of W: its span and the Ex_Pose / td columns; h_f, g_f
    if (f >= F) return;
    const double sf = s[P + f];
    double *row = w.W + (size_t)f * w.WS;
    for (int c = w.fspan[2 * f] + lane; c <= w.fspan[2 * f + 1]; c += 64) row[c] *= sf * s[lw_fullcol(w, c)];
    for (int c = 6 * w.NF + lane; c < w.PC; c += 64) row[c] *= sf * s[lw_fullcol(w, c)];
    if (lane == 0) { w.hf[f] *= sf * sf; w.gf[f] *= sf; }
}
// ---- the Schur reduce Wn^T Wn as a hand-written fp64 MFMA SYRK over the compact (pose-only) columns ----------------------------------------
// W is F x WS row-major, one row per feature, PC columns used; Wn = W / sqrt(den_f) with den_f = h_f + lm_f^2 (1 for a constant feature: its row is zero) is formed
// on the way into LDS and never stored. One workgroup per 64 x 64 tile of the lower triangle (in compact columns) and per K split; four waves, each a 32 x 32
// sub-tile = 2 x 2 v_mfma_f64_16x16x4_f64 accumulators; the two K x 64 panels stream through LDS SY_KB feature rows at a time, the next chunk's loads in flight.
// Structure: K runs over the features in START-FRAME order (forder), and a chunk of SY_KB of them touches the compact columns kspan[2 q] .. kspan[2 q + 1] only
// (6 columns per frame of its features' spans) — a tile whose row or column range that interval misses skips the chunk. For the stress window (features seen in ~21
// of 51 frames) a chunk meets ~7 of the 15 tiles; the dense F x 765 product of rounds 1-3 issued 37 x the flops of SURVEY 8(d)'s 2 sum (6 k_f)^2.
// Output: every workgroup stores ITS partial tile (zeros when it met no chunk) to SC[split][tile]; diagonal-tile workgroups also accumulate the right-hand side's
// Wn^T (g_f / sqrt(den_f)) for their 64 columns (the panel is in LDS anyway) into SC's tail. lw_schur_prep sums the SY_KS partials in fixed order: no atomics,
// bit-reproducible.
#define SY_LD 65
#define SY_NT(PC) (((PC) + 63) / 64)
// den_f = h_f + lm_f^2 (1 for a constant feature: its row of W is zero), rsd_f = 1 / sqrt(den_f), tmpF_f = g_f / sqrt(den_f): once per linear solve
__global__ __launch_bounds__(256) void lw_den(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= w.F) return;
    const double lmf = w.vec[w.P + f], d = w.fconst[f] ? 1.0 : w.hf[f] + lmf * lmf, rs = 1.0 / sqrt(d);
    w.den[f] = d; w.rsd[f] = rs; w.tmpF[f] = w.gf[f] * rs;
}
#define SY_LIST 1024
__global__ __launch_bounds__(256) void lw_syrk_mfma(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int PC = w.PC, WS = w.WS, F = w.F, nchunk = w.nchunk;
    const double *W = w.W, *rsd = w.rsd, *gn = w.tmpF; const int *kspan = w.kspan;
    __shared__ double sA[SY_KB * SY_LD], sB[SY_KB * SY_LD], sG[SY_KB];
    __shared__ int s_list[SY_LIST], s_cnt[5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= (int)blockIdx.x) ti++;
    const int nt = SY_NT(PC);
    if (ti >= nt || F == 0) return;                   // the grid is sized for the group's largest window
    const int tj = blockIdx.x - ti * (ti + 1) / 2, i0 = 64 * ti, j0 = 64 * tj, ks = blockIdx.y, nks = w.nks;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const bool diag = ti == tj;
    lw_double4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = lw_double4{0, 0, 0, 0};
    double racc = 0;                                  // diagonal tiles, tid < 64: sum_f Wn(f, j0 + tid) g_f / sqrt(den_f)
    double ra[SY_KB / 4], rb[SY_KB / 4], rg = 0;
    auto gload = [&](int q) {                         // straight-line loads: an entry outside the matrix reads a valid address and is zeroed on the way to LDS
        const int k0 = q * SY_KB;
#pragma unroll
        for (int u = 0; u < SY_KB / 4; u++) {
            const int e = tid + 256 * u, kk = e >> 6, cc = e & 63, k = k0 + kk;
            const bool kin = k < F;
            const int f = kin ? k : 0;
            const double rs = rsd[f];
            const size_t rowoff = (size_t)f * WS;
            const double vb = W[rowoff + min(j0 + cc, PC - 1)];
            rb[u] = (kin && j0 + cc < PC) ? vb * rs : 0.0;
            if (!diag) { const double va = W[rowoff + min(i0 + cc, PC - 1)]; ra[u] = (kin && i0 + cc < PC) ? va * rs : 0.0; }
        }
        if (diag && tid < SY_KB) { const int k = k0 + tid; rg = k < F ? gn[k] : 0.0; }
    };
    // chunks of this K split: q = ks, ks + nks, ... (interleaved: neighbouring chunks have neighbouring spans, so every split meets every tile about equally often).
    // The ones whose span reaches both the tile's rows and its columns are listed in LDS first, in ascending order (ballot compaction: the summation order is fixed).
    const int ncand = (nchunk - ks + nks - 1) / nks;
    for (int base = 0; base < ncand; base += SY_LIST) {
        int nl = 0;
        for (int r0 = base; r0 < min(ncand, base + SY_LIST); r0 += 256) {
            const int cnd = r0 + tid, q = ks + nks * cnd;
            bool rel = false;
            if (cnd < ncand) { const int lo = kspan[2 * q], hi = kspan[2 * q + 1]; rel = lo < j0 + 64 && hi >= j0 && lo < i0 + 64 && hi >= i0; }
            const unsigned long long m = __ballot(rel);
            if (lane == 0) s_cnt[wave] = __popcll(m);
            __syncthreads();
            int off = nl;
            for (int v = 0; v < wave; v++) off += s_cnt[v];
            if (rel) s_list[off + __popcll(m & ((1ull << lane) - 1ull))] = q;
            nl += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
            __syncthreads();
        }
        if (nl) gload(s_list[0]);
        for (int it = 0; it < nl; it++) {
            if (!diag) {
#pragma unroll
                for (int u = 0; u < SY_KB / 4; u++) { const int e = tid + 256 * u; sA[(e >> 6) * SY_LD + (e & 63)] = ra[u]; }
            }
#pragma unroll
            for (int u = 0; u < SY_KB / 4; u++) { const int e = tid + 256 * u; sB[(e >> 6) * SY_LD + (e & 63)] = rb[u]; }
            if (diag && tid < SY_KB) sG[tid] = rg;
            __syncthreads();
            if (it + 1 < nl) gload(s_list[it + 1]);
            const double *pA = diag ? sB : sA;
#pragma unroll
            for (int s4 = 0; s4 < SY_KB / 4; s4++) {
                double av[2], bv[2];
                const int kr = (4 * s4 + (lane >> 4)) * SY_LD + (lane & 15);
#pragma unroll
                for (int a = 0; a < 2; a++) { av[a] = pA[kr + wi + 16 * a]; bv[a] = sB[kr + wj + 16 * a]; }
#pragma unroll
                for (int a = 0; a < 2; a++)
#pragma unroll
                    for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
            }
            if (diag && tid < 64) {
#pragma unroll 8
                for (int kk = 0; kk < SY_KB; kk++) racc += sB[kk * SY_LD + tid] * sG[kk];
            }
            __syncthreads();
        }
    }
    const int ntile = nt * (nt + 1) / 2;
    double *out = w.SC + ((size_t)ks * ntile + blockIdx.x) * 4096;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) out[(wi + 16 * a + (lane >> 4) + 4 * q4) * 64 + wj + 16 * b + (lane & 15)] = acc[a][b][q4];
    if (diag && tid < 64) w.SC[(size_t)nks * ntile * 4096 + (size_t)ks * nt * 64 + j0 + tid] = racc;
}
// S = Hpp + diag(lm_p^2) - sum over the K splits of Wn^T Wn (the partial tiles of lw_syrk_mfma, summed in split order: no atomics); row P of S = g_p - Wn^T (g_f /
// sqrt(den)) (the right-hand side rides through the factorisation as one more row); the factorisation's status word = 0.
__global__ __launch_bounds__(256) void lw_schur_prep(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int P = w.P, F = w.F, nt = SY_NT(w.PC), ntile = nt * (nt + 1) / 2, nks = w.nks, i = blockIdx.x, tid = threadIdx.x;
    if (i > P) return;
    const double *lm = w.vec, *SC = w.SC;
    double *row = w.S + (size_t)i * P;
    if (i == P) {                                      // the right-hand side's row
        for (int c = tid; c < P; c += 256) {
            double v = w.gp[c];
            const int cc = lw_compcol(w, c);
            if (F && cc >= 0) { double sub = 0; for (int k = 0; k < nks; k++) sub += SC[(size_t)nks * ntile * 4096 + (size_t)k * nt * 64 + cc]; v -= sub; }
            row[c] = v;
        }
        if (tid == 0) *w.info = 0;
        return;
    }
    const double *hrow = w.Hpp + (size_t)i * P;
    const int np = 15 * w.NF, fr = i / 15, ci = lw_compcol(w, i);
    const int c0 = i >= np ? 0 : 15 * max(0, fr - w.bandf), c1 = i >= np ? np : min(np, 15 * (fr + w.bandf + 1));
    for (int c = tid; c < P; c += 256) {
        if (c < np && (c < c0 || c >= c1)) { row[c] = 0.0; continue; }      // structurally zero: nothing to read
        double v = hrow[c] + (i == c ? lm[i] * lm[i] : 0.0);
        const int cj = lw_compcol(w, c);
        if (F && ci >= 0 && cj >= 0) {
            const int r = max(ci, cj), cm = min(ci, cj), tr = r >> 6, tc = cm >> 6;
            const double *src = SC + ((size_t)(tr * (tr + 1) / 2 + tc)) * 4096 + (r & 63) * 64 + (cm & 63);
            double sub = 0;
            for (int k = 0; k < nks; k++) sub += src[(size_t)k * ntile * 4096];
            v -= sub;
        }
        row[c] = v;
    }
}

// ---- dense Cholesky of the reduced system (P x P, fp64) with the right-hand side as row P: S = L L^T, L[P][0..P-1] = L^-1 rhs -----------------------
// Right-looking over 64-column blocks, ONE launch per block column and no vendor library:
//   lw_chol_panel   (block column 0) a workgroup = the 64 x 64 diagonal block + 64 rows of the panel below it, all as MFMA tiles in registers: four 16-column
//                   sub-panels, each a one-wave factorisation of the diagonal 16 x 16 tile (lane = row, v_readlane) that also yields the tile's inverse, the rows
//                   below by MFMA against that inverse, rank-16 updates by MFMA. Every workgroup factors the diagonal block itself — the rows below cannot start
//                   before it is known, so that costs no time and saves a launch per block column.
//   lw_chol_step    (block columns 1 ..) the same panel workgroups, which first take the previous column's update of THEIR block column into their registers
//                   (T -= L21 L21^T, operands through LDS), beside workgroups that apply the previous column's update to the 64 x 64 tiles further right (the two
//                   panels through LDS, MFMA), rhs row included. Two launches per block column (panel, then the whole trailing update) were 0.63 ms per
//                   765 x 765 factorisation; one launch with 256-row slabs and 4-column panels 0.55; 128-row slabs and 16-column sub-panels 0.39 (a step:
//                   2.3 us load, 6.1 us the slab's own update, 4 x (2.6 us tile factorisation + 0.7 us MFMA strip and update); lw_chol_back 0.10).
//   lw_chol_back    L^T y = z by one workgroup (z = row P of the factor).
#define CH_NB 64
#define CH_LD 65
#define CH_M 2                // tile rows per wave: a slab = 64 CH_M rows. fp64 MFMA is 64 cycles an instruction on this part and a slab's MFMAs all run on ONE CU:
#define CH_ROWS (64 * CH_M)   // with 256-row slabs (CH_M = 4) the update a slab takes first was 11 us of a 33 us step, mostly MFMA issue
#define CH_LDS (CH_ROWS * 17 + 2 * 64 * 17 + 2 * 16 * 17)     // the panel workgroup's LDS, doubles
#define CH_BELOW (CH_ROWS - 64)          // rows of the panel below the diagonal block per workgroup (256 rows with the block's own 64)
// One workgroup = a CH_ROWS x 64 slab: the 64 rows of the diagonal block + CH_BELOW rows below it, as 16 x 16 MFMA accumulator tiles in registers (wave w owns the
// tile rows w, w + 4, ...).
// PRE: the update of the previous block column (jp = j0 - 64) has not been applied to this block column yet — the slab takes it itself, in registers, before it
// factors (lw_chol_step: the rest of that update runs beside it in the same launch)
__device__ __forceinline__ double lw_readlane(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
// phase stamps of one panel workgroup (diagnostic build only: make DEFS=-DVILF_LW_STAMPS): 100 MHz wall clock of workgroup 0 at block column 128
#ifdef VILF_LW_STAMPS
__device__ long long lw_dbg_stamps[64];
#define LWSTAMP(i) do { if (wg == 0 && threadIdx.x == 0 && j0 == 128) lw_dbg_stamps[i] = wall_clock64(); } while (0)
#else
#define LWSTAMP(i) do { } while (0)
#endif
template <bool PRE>
__device__ __forceinline__ void lw_chol_panel_body(int P, double *S, int j0, int *info, int wg, double *s_pan, double *s_lp) {
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, g4 = lane >> 4;
    const int nb = min(CH_NB, P - j0), P1 = P + 1;                 // the last block is padded with an identity
    const int below0 = j0 + nb + CH_BELOW * wg;                    // first global row of this workgroup's rows below the block
    auto grow = [&](int lr) { return lr < CH_NB ? j0 + lr : below0 + (lr - CH_NB); };      // slab row -> row of S (rows 64.. of the slab are below the block)
    auto live = [&](int lr) { return lr < CH_NB ? lr < nb : below0 + (lr - CH_NB) < P1; };
    LWSTAMP(0);
    lw_double4 T[CH_M][4];                                            // [m][tile column]: tile row wave + 4 m
#pragma unroll
    for (int m = 0; m < CH_M; m++)
#pragma unroll
        for (int tcl = 0; tcl < 4; tcl++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                // straight-line: every entry loads (a dead one from the block's first entry) and is selected afterwards — 64 loads in flight instead of a branch each
                const int lr = 16 * (wave + 4 * m) + g4 + 4 * q, c = 16 * tcl + c16;
                const bool ld = lr < CH_NB ? (lr < nb && c <= lr) : (live(lr) && c < nb);
                const double v = S[ld ? (size_t)grow(lr) * P + j0 + c : (size_t)j0 * P + j0];
                T[m][tcl][q] = ld ? v : ((lr < CH_NB && lr == c && lr >= nb) ? 1.0 : 0.0);
            }
    LWSTAMP(1);
    if (PRE) {
        // T -= L[slab rows][jp .. jp + 63] L[block rows][jp .. jp + 63]^T. The two operands go through LDS in four K-quarters of 16, read from S along k (a wave
        // instruction covers four rows x 128 contiguous bytes; MFMA operands fetched straight from S touch sixteen rows per instruction and took 18 us instead of 4).
        // The staging area is the panel's s_pan / s_lp region and what follows it (the caller's buffer holds 16 x 257 + 16 x 65 doubles).
        const int jp = j0 - CH_NB;
        double *sA = s_pan, *sB = s_pan + 16 * (CH_ROWS + 1);
        // straight-line loads (a dead row reads a live one and is zeroed on the way to LDS: a branch per load kept them from being issued together), the next
        // quarter's loads are in flight while this quarter's MFMAs run
        const double *pa[CH_ROWS / 16], *pb[4];
        unsigned alive = 0;
#pragma unroll
        for (int u = 0; u < CH_ROWS / 16; u++) { const int e = tid + 256 * u, row = e >> 4; const bool lv = live(row); alive |= (lv ? 1u : 0u) << u; pa[u] = S + (size_t)(lv ? grow(row) : j0) * P + jp + (e & 15); }
#pragma unroll
        for (int u = 0; u < 4; u++) { const int e = tid + 256 * u, c = e >> 4; const bool lv = c < nb; alive |= (lv ? 1u : 0u) << (CH_ROWS / 16 + u); pb[u] = S + (size_t)(j0 + (lv ? c : 0)) * P + jp + (e & 15); }
        // every quarter's loads are issued here, in one batch: a round trip to S (written by the previous launch on other CUs) is ~2.8 us, and a quarter's
        // MFMAs hide nothing of it — four dependent round trips were 11 of a step's 33 us (80 values per lane: the register file of a one-wave-per-SIMD kernel holds them)
        double va[4][CH_ROWS / 16], vb[4][4];
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
#pragma unroll
            for (int u = 0; u < CH_ROWS / 16; u++) va[kq][u] = pa[u][16 * kq];
#pragma unroll
            for (int u = 0; u < 4; u++) vb[kq][u] = pb[u][16 * kq];
        }
#pragma unroll
        for (int kq = 0; kq < 4; kq++) {
            __syncthreads();
#pragma unroll
            for (int u = 0; u < CH_ROWS / 16; u++) { const int e = tid + 256 * u; sA[(e & 15) * (CH_ROWS + 1) + (e >> 4)] = ((alive >> u) & 1) ? -va[kq][u] : 0.0; }
#pragma unroll
            for (int u = 0; u < 4; u++) { const int e = tid + 256 * u; sB[(e & 15) * 65 + (e >> 4)] = ((alive >> (CH_ROWS / 16 + u)) & 1) ? vb[kq][u] : 0.0; }
            __syncthreads();
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                double av[CH_M], bv[4];
                const int k = 4 * s4 + g4;
#pragma unroll
                for (int t = 0; t < CH_M; t++) av[t] = sA[k * (CH_ROWS + 1) + 16 * (wave + 4 * t) + c16];
#pragma unroll
                for (int t = 0; t < 4; t++) bv[t] = sB[k * 65 + 16 * t + c16];
#pragma unroll
                for (int m = 0; m < CH_M; m++)
#pragma unroll
                    for (int tcl = 0; tcl < 4; tcl++)
                        if (m > 0 || tcl <= wave) T[m][tcl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[tcl], T[m][tcl], 0, 0, 0);
            }
        }
        __syncthreads();                                           // the staging area becomes s_pan / s_lp
    }
    // ---- the slab's 64 columns as four 16-column sub-panels, two barriers each (4-column panels with a thread per row were 32 barriers and 16 dependent 4 x 4
    // factorisations per block column: 28 of a step's 38 us):
    //   A  the wave that owns the diagonal 16 x 16 tile (tile row tc = wave tc) factors it ALONE, lane = row: the tile goes through LDS into row layout, the pivot
    //      row's entries travel by v_readlane, nothing waits for a barrier. Lanes 16 .. 31 run the same elimination on the rows of an identity: they end as L^-T.
    //      Meanwhile every wave parks its tiles of the sub-panel in LDS (rows it owns: no other wave touches them).
    //   B  X = T L^-T for every tile row of the sub-panel, four MFMAs per tile (the triangular solve as a product with the explicit 16 x 16 inverse), X -> S, and
    //      into LDS as the operand of
    //   C  the rank-16 update of the tiles right of the sub-panel (own rows as A operand, the block's rows — double-buffered copy — as B operand).
    LWSTAMP(2);
    double *s_own = s_pan, *s_blk = s_pan + CH_ROWS * 17, *s_inv = s_blk + 2 * 64 * 17, *s_dg = s_inv + 16 * 17;      
#pragma unroll
    for (int tc = 0; tc < 4; tc++) {
#pragma unroll
        for (int m = 0; m < CH_M; m++)
            if (m > 0 || wave >= tc) {
#pragma unroll
                for (int q = 0; q < 4; q++) s_own[(16 * (wave + 4 * m) + g4 + 4 * q) * 17 + c16] = T[m][tc][q];
            }
        if (wave == tc) {
#pragma unroll
            for (int q = 0; q < 4; q++) s_dg[(g4 + 4 * q) * 17 + c16] = T[0][tc][q];
            __builtin_amdgcn_wave_barrier();
            // lane = row: lanes 0 .. 15 the tile's rows, lanes 16 .. 31 the rows of an identity (they end as L^-T); column k: scale by 1 / sqrt(pivot), then
            // [j] -= [k] l_jk with l_jk by v_readlane. (Tried: the same elimination with v_mov_b64_dpp row_newbcast instead of v_readlane, square-root free with a
            // shorter pivot chain: 2.6 us per tile like this one, but 180 more registers — one workgroup per CU instead of two, which costs a group of 32 windows 2x.)
            double bb[16];
            const int r = lane & 15;
#pragma unroll
            for (int j = 0; j < 16; j++) { const double dv = s_dg[r * 17 + j]; bb[j] = lane < 16 ? (j <= r ? dv : 0.0) : ((lane < 32 && j == r) ? 1.0 : 0.0); }
            bool bad = false;
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const double piv = lw_readlane(bb[k], k);
                bad = bad || !(piv > 0.0);
                bb[k] *= rsqrt_h3(piv);
                double sv[16];
#pragma unroll
                for (int j = k + 1; j < 16; j++) sv[j] = lw_readlane(bb[k], j);
#pragma unroll
                for (int j = k + 1; j < 16; j++) bb[j] = fma(-bb[k], sv[j], bb[j]);
            }
            if (lane >= 16 && lane < 32) {
#pragma unroll
                for (int n = 0; n < 16; n++) s_inv[r * 17 + n] = bb[n];
            }
            if (bad && wg == 0 && lane == 0) *info = 1;
        }
        LWSTAMP(3 + 4 * tc);
        __syncthreads();
        LWSTAMP(4 + 4 * tc);
        double bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) bv[s4] = s_inv[(4 * s4 + g4) * 17 + c16];
#pragma unroll
        for (int m = 0; m < CH_M; m++)
            if (m > 0 || wave >= tc) {
                lw_double4 X = lw_double4{0, 0, 0, 0};
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) X = __builtin_amdgcn_mfma_f64_16x16x4f64(s_own[(16 * (wave + 4 * m) + c16) * 17 + 4 * s4 + g4], bv[s4], X, 0, 0, 0);
                const bool dtile = m == 0 && wave == tc;               // the diagonal tile: X = L, its upper part is rounding noise
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int lr = 16 * (wave + 4 * m) + g4 + 4 * q, col = 16 * tc + c16;
                    const double xv = (dtile && c16 > g4 + 4 * q) ? 0.0 : X[q];
                    T[m][tc][q] = xv;
                    s_own[lr * 17 + c16] = xv;
                    if (m == 0) s_blk[(tc & 1) * 64 * 17 + lr * 17 + c16] = xv;
                    if (live(lr) && (lr >= CH_NB || wg == 0) && col < nb && (!dtile || c16 <= g4 + 4 * q)) S[(size_t)grow(lr) * P + j0 + col] = xv;      // the block's own rows are written by workgroup 0 only
                }
            }
        LWSTAMP(5 + 4 * tc);
        if (tc < 3) {
            __syncthreads();
            LWSTAMP(6 + 4 * tc);
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                double av[CH_M], bw[4];
#pragma unroll
                for (int m = 0; m < CH_M; m++) av[m] = -s_own[(16 * (wave + 4 * m) + c16) * 17 + 4 * s4 + g4];
#pragma unroll
                for (int tcl = 0; tcl < 4; tcl++) bw[tcl] = tcl > tc ? s_blk[(tc & 1) * 64 * 17 + (16 * tcl + c16) * 17 + 4 * s4 + g4] : 0.0;
#pragma unroll
                for (int m = 0; m < CH_M; m++)
#pragma unroll
                    for (int tcl = 0; tcl < 4; tcl++)
                        if (tcl > tc && (m > 0 || tcl <= wave)) T[m][tcl] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bw[tcl], T[m][tcl], 0, 0, 0);
            }
        }
    }
}
// block column 0. npanel = the slabs of 256 rows the block column needs (the grid is sized for the group's largest window)
__device__ __forceinline__ int lw_chol_npanel(int P, int j0) { const int nb = min(CH_NB, P - j0), below = P + 1 - (j0 + nb); return max(1, (below + CH_BELOW - 1) / CH_BELOW); }
__global__ __launch_bounds__(256) void lw_chol_panel(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    if ((int)blockIdx.x >= lw_chol_npanel(w.P, 0)) return;
    __shared__ double s_pan[CH_LDS];
    lw_chol_panel_body<false>(w.P, w.S, 0, w.info, (int)blockIdx.x, s_pan, nullptr);
}
__global__ __launch_bounds__(256) void lw_chol_panel_raw(int P, double *S, int *info) {
    __shared__ double s_pan[CH_LDS];
    lw_chol_panel_body<false>(P, S, 0, info, (int)blockIdx.x, s_pan, nullptr);
}
// A22 -= L21 L21^T, lower 64 x 64 tiles of the rows / columns j1 .. P (row P = rhs). Tile u of the lower triangle, shifted by `shift` tile rows / columns
// (shift 1 = lw_chol_step: the tiles right of the next block column; that column itself is the panel workgroups')
__device__ __forceinline__ void lw_chol_update_body(int P, double *S, int j0, int nb, int u, int shift, double *sA, double *sB) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, P1 = P + 1, j1 = j0 + nb;
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= u) ti++;
    const int tj = u - ti * (ti + 1) / 2 + shift;
    ti += shift;
    const int r0 = j1 + 64 * ti, c0 = j1 + 64 * tj;
    // panels: element (row i, k) of L21 -> s[k][i] (k-major for the MFMA operand reads); the global read is coalesced along k
    for (int e = tid; e < 64 * CH_NB; e += 256) {
        const int i = e >> 6, k = e & 63;
        sA[k * CH_LD + i] = (r0 + i < P1 && k < nb) ? S[(size_t)(r0 + i) * P + j0 + k] : 0.0;
        sB[k * CH_LD + i] = (c0 + i < P1 && k < nb) ? S[(size_t)(c0 + i) * P + j0 + k] : 0.0;
    }
    __syncthreads();
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    lw_double4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = lw_double4{0, 0, 0, 0};
#pragma unroll
    for (int s4 = 0; s4 < CH_NB / 4; s4++) {
        double av[2], bv[2];
        const int kr = (4 * s4 + (lane >> 4)) * CH_LD + (lane & 15);
#pragma unroll
        for (int a = 0; a < 2; a++) { av[a] = sA[kr + wi + 16 * a]; bv[a] = sB[kr + wj + 16 * a]; }
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int row = r0 + wi + 16 * a + (lane >> 4) + 4 * q, col = c0 + wj + 16 * b + (lane & 15);
                if (row < P1 && col < P && col <= row) S[(size_t)row * P + col] -= acc[a][b][q];
            }
}
// One launch per block column j0 >= 64: the first npanel workgroups factor block column j0 (taking the update of column j0 - 64 into their registers first),
// the others apply the update of column j0 - 64 to the tiles right of block column j0. The two sets write disjoint parts of S and both only read column j0 - 64:
// the panel no longer waits for a whole trailing update, and a block column costs one launch instead of two.
__device__ __forceinline__ void lw_chol_step_body(int P, double *S, int j0, int *info, double *s_buf) {
    const int npanel = lw_chol_npanel(P, j0), nt = (P + 1 - j0 + 63) / 64;     // tiles of the rows / columns j0 .. P; column 0 of them is the panel's
    if ((int)blockIdx.x >= npanel + nt * (nt - 1) / 2) return;
    if ((int)blockIdx.x < npanel) lw_chol_panel_body<true>(P, S, j0, info, (int)blockIdx.x, s_buf, nullptr);
    else lw_chol_update_body(P, S, j0 - CH_NB, CH_NB, (int)blockIdx.x - npanel, 1, s_buf, s_buf + CH_NB * CH_LD);
}
__global__ __launch_bounds__(256) void lw_chol_step(const LwWin *ws, int j0, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk) || j0 >= w.P) return;          // device trust-region loop: not needed; a smaller window of the group: already factored
    __shared__ double s_buf[2 * CH_NB * CH_LD];
    lw_chol_step_body(w.P, w.S, j0, w.info, s_buf);
}
__global__ __launch_bounds__(256) void lw_chol_step_raw(int P, double *S, int j0, int *info) {
    __shared__ double s_buf[2 * CH_NB * CH_LD];
    lw_chol_step_body(P, S, j0, info, s_buf);
}
// L^T y = z, z = row P of the factor; one workgroup, columns right to left in 64-blocks. Per block: wave 0 solves the block's triangle — lane = row, the 64 steps
// fully unrolled so that the pivot row's value travels by v_readlane (a shuffle per step through LDS and a division per step were most of the 215 us this kernel
// took), reciprocal diagonal computed once per block — while the other waves already stage the NEXT block's triangle in the second LDS buffer; then every earlier
// entry subtracts its part, thread per entry, the block's 64 rows of L read along a row (coalesced, eight loads in flight).
__device__ __forceinline__ void lw_chol_back_body(int P, const double *S, double *y, double *s_y, double *s_blk, double *s_tri) {
    const int tid = threadIdx.x;
    for (int i = tid; i < P; i += 1024) s_y[i] = S[(size_t)P * P + i];
    const int nblk = (P + CH_NB - 1) / CH_NB;
    // element e of a block's triangle; rows / columns past the matrix end: identity (the unrolled solve walks all 64)
    auto tri_at = [&](int bk, int e) { const int j0 = CH_NB * bk, nb = min(CH_NB, P - j0), r = e >> 6, c = e & 63; return (r < nb && c < nb) ? S[(size_t)(j0 + r) * P + j0 + c] : (r == c ? 1.0 : 0.0); };
    for (int e = tid; e < CH_NB * CH_NB; e += 1024) s_tri[(e >> 6) * CH_LD + (e & 63)] = tri_at(nblk - 1, e);
    if (tid < CH_NB) s_tri[tid * CH_LD + CH_NB] = 0.0;       // the padding column: the zero a lane at or above the pivot reads in the solve below (never rewritten)
    __syncthreads();
    constexpr int NST = 1024 - 64, NPER = (CH_NB * CH_NB + NST - 1) / NST;
    for (int bk = nblk - 1; bk >= 0; bk--) {
        const int j0 = CH_NB * bk, nb = min(CH_NB, P - j0);
        double nx[NPER];
        if (tid < 64) {                            // y_blk = L_bb^-T z_blk
            double z = (tid < nb) ? s_y[j0 + tid] : 0.0;
            const double dinv = 1.0 / s_tri[tid * CH_LD + tid];
#pragma unroll 1
            for (int jc = CH_NB - 16; jc >= 0; jc -= 16) {          // 16 steps at a time: their 16 LDS reads are hoisted in front of the chain, not all 64
#pragma unroll
                for (int k = 15; k >= 0; k--) {
                    const int j = jc + k;
                    const double yj = lw_readlane(z, j) * lw_readlane(dinv, j);
                    // the read is unconditional — lanes at or above the pivot read the row's zero padding, z - 0 yj = z: written as a conditional read it is compiled
                    // into a branch around the ds_read with its own s_waitcnt, 64 dependent LDS round trips per block instead of 16 reads in flight in front of the chain
                    const double t = s_tri[j * CH_LD + ((tid < j) ? tid : CH_NB)];
                    z = (tid == j) ? yj : z - t * yj;
                }
            }
            if (tid < nb) s_y[j0 + tid] = z;
            s_blk[tid] = (tid < nb) ? z : 0.0;
        } else if (bk > 0) {                       // meanwhile: the next block's triangle on its way (registers; into LDS once this block's solve is done)
#pragma unroll
            for (int u = 0; u < NPER; u++) { const int e = tid - 64 + u * NST; nx[u] = e < CH_NB * CH_NB ? tri_at(bk - 1, e) : 0.0; }
        }
        __syncthreads();
        if (tid >= 64 && bk > 0) {
#pragma unroll
            for (int u = 0; u < NPER; u++) { const int e = tid - 64 + u * NST; if (e < CH_NB * CH_NB) s_tri[(e >> 6) * CH_LD + (e & 63)] = nx[u]; }
        }
        for (int i = tid; i < j0; i += 1024) {     // z_i -= sum_r L[j0 + r][i] y[j0 + r]
            const double *col = S + (size_t)j0 * P + i;
            double sum = 0;
#pragma unroll 1
            for (int h = 0; h < CH_NB; h += 16) {  // 16 loads of the lane in flight at a time (rows past the block: clamped, their y is zero)
                double lv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) lv[r] = col[(size_t)min(h + r, nb - 1) * P];
#pragma unroll
                for (int r = 0; r < 16; r++) sum += lv[r] * s_blk[h + r];
            }
            s_y[i] -= sum;
        }
        __syncthreads();
    }
    for (int i = tid; i < P; i += 1024) y[i] = s_y[i];
}
__global__ __launch_bounds__(1024) void lw_chol_back(const LwWin *ws, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    extern __shared__ double s_y[];                   // P entries (of the group's largest window)
    __shared__ double s_blk[CH_NB], s_tri[CH_NB * CH_LD];
    lw_chol_back_body(w.P, w.S, w.rhs, s_y, s_blk, s_tri);
}
__global__ __launch_bounds__(1024) void lw_chol_back_raw(int P, const double *S, double *y) {
    extern __shared__ double s_y[];
    __shared__ double s_blk[CH_NB], s_tri[CH_NB * CH_LD];
    lw_chol_back_body(P, S, y, s_y, s_blk, s_tri);
}
// row-wise dots, one 64-lane wave per row. mode 0 (x^T H x pieces for the vector in vec): rows 0 .. P - 1: Hpp vec -> tmpP, rows P .. P + F - 1:
// W_f . vec -> tmpF; mode 1 (back substitution of the features): y_f = (g_f - W_f . rhs) / den_f. W's rows run over the compact (pose) columns.
__global__ __launch_bounds__(256) void lw_rowdot(const LwWin *ws, int mode, int sk) {
    const LwWin &w = ws[blockIdx.z];
    if (lw_skip(w, sk)) return;                       // device trust-region loop: this part of the iteration is not needed
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= (mode == 0 ? w.P + w.F : w.F)) return;
    const bool hp = mode == 0 && row < w.P;
    const int r = hp ? row : (mode == 0 ? row - w.P : row);
    const double *x = mode == 0 ? w.vec : w.rhs;
    double s = 0;
    if (hp) {                                          // the band's columns (and the Ex_Pose / td columns behind the frames)
        const double *a = w.Hpp + (size_t)r * w.P;
        const int np = 15 * w.NF, fr = r / 15, c0 = r >= np ? 0 : 15 * max(0, fr - w.bandf), c1 = r >= np ? np : min(np, 15 * (fr + w.bandf + 1));
        for (int c = c0 + lane; c < c1; c += 64) s += a[c] * x[c];
        for (int c = np + lane; c < w.P; c += 64) s += a[c] * x[c];
    } else {                                           // a row of W: its span and the Ex_Pose / td columns
        const double *a = w.W + (size_t)r * w.WS;
        for (int c = w.fspan[2 * r] + lane; c <= w.fspan[2 * r + 1]; c += 64) s += a[c] * x[lw_fullcol(w, c)];
        for (int c = 6 * w.NF + lane; c < w.PC; c += 64) s += a[c] * x[lw_fullcol(w, c)];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) {
        if (mode == 0) (hp ? w.tmpP : w.tmpF)[r] = s;
        else w.yf[r] = (w.gf[r] - s) / w.den[r];            // y_f = (g_f - W_f . y_p) / den_f
    }
}

// ---- host-side manifold helpers (PoseLocalParameterization, utility.h) ------------------------------------------------------------
struct HQ { double x, y, z, w; };
inline HQ hq_mul(const HQ &a, const HQ &b) { return HQ{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z}; }
inline void h_pose_plus(const double *x, const double *d, double *o) {      // pose_local_parameterization.cpp:3-17
    for (int k = 0; k < 3; k++) o[k] = x[k] + d[k];
    HQ q{x[3], x[4], x[5], x[6]}, dq{d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0};
    HQ r = hq_mul(q, dq);
    const double n = std::sqrt(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
    o[3] = r.x / n; o[4] = r.y / n; o[5] = r.z / n; o[6] = r.w / n;
}
inline void h_q2R(const double *q_xyzw, double *R) {
    const double n = std::sqrt(q_xyzw[0] * q_xyzw[0] + q_xyzw[1] * q_xyzw[1] + q_xyzw[2] * q_xyzw[2] + q_xyzw[3] * q_xyzw[3]);
    const double x = q_xyzw[0] / n, y = q_xyzw[1] / n, z = q_xyzw[2] / n, w = q_xyzw[3] / n;
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w); R[2] = 2 * (x * z + y * w);
    R[3] = 2 * (x * y + z * w); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
    R[6] = 2 * (x * z - y * w); R[7] = 2 * (y * z + x * w); R[8] = 1 - 2 * (x * x + y * y);
}
inline void h_R2ypr(const double *R, double *ypr) {
    const double y = std::atan2(R[3], R[0]);
    const double p = std::atan2(-R[6], R[0] * std::cos(y) + R[3] * std::sin(y));
    const double r = std::atan2(R[2] * std::sin(y) - R[5] * std::cos(y), -R[1] * std::sin(y) + R[4] * std::cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}
inline void h_ypr2R(const double *ypr, double *R) {
    const double y = ypr[0] / 180.0 * M_PI, p = ypr[1] / 180.0 * M_PI, r = ypr[2] / 180.0 * M_PI;
    const double Rz[9] = {std::cos(y), -std::sin(y), 0, std::sin(y), std::cos(y), 0, 0, 0, 1}, Ry[9] = {std::cos(p), 0, std::sin(p), 0, 1, 0, -std::sin(p), 0, std::cos(p)},
                 Rx[9] = {1, 0, 0, 0, std::cos(r), -std::sin(r), 0, std::sin(r), std::cos(r)};
    double T[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += Rz[3 * i + k] * Ry[3 * k + j]; T[3 * i + j] = s; }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += T[3 * i + k] * Rx[3 * k + j]; R[3 * i + j] = s; }
}
}  // namespace


// ------------------------------------------------------------------------------------------------------------------
// The trust-region loop on the device (trust_region_minimizer.cc + dogleg_strategy.cc, traditional dogleg, Jacobi scaling — the logic of the host
// loop further down, restated once more): the scalars of the minimizer live in LwCtl, the N-vectors in device buffers, and five single-workgroup
// kernels carry the decisions between the factor / Schur / Cholesky launches. The host enqueues max_num_iterations iterations without reading
// anything back; every launch of an iteration looks at the skip flags (a finished solve, a rejected step that reuses the linear solve, an invalid
// step that needs no evaluation) and returns at once when its part is not needed. What the device loop does not do: raise mu and factor again after a
// failed Cholesky (the number of launches is not known when they are enqueued) — it sets `fallback` and the host loop redoes the solve.
namespace {
using namespace vd;
#define TR_T 1024
__device__ __forceinline__ double tr_sum(double v, double *s_red) {       // all TR_T threads; result to every thread
    v = vilf_wave_sum64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int k = 0; k < TR_T / 64; k++) t += s_red[k];
    return t;
}
// the cost of the evaluation that just ran: its producers' partials in slot order (strided partial sums, then tr_sum's fixed tree): the same bits every run
__device__ __forceinline__ double tr_cost_sum(const LwWin &a, double *s_red) {
    const int n = a.ncostv + a.NF + (a.pn ? 1 : 0);
    double v = 0;
    for (int i = threadIdx.x; i < n; i += TR_T) v += a.costP[i];
    return tr_sum(v, s_red);
}
__device__ __forceinline__ double tr_max(double v, double *s_red) {
    v = vilf_wave_max64(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0;
#pragma unroll
    for (int k = 0; k < TR_T / 64; k++) t = fmax(t, s_red[k]);
    return t;
}
__device__ __forceinline__ void tr_finish(LwCtl *c, int term) { c->termination = term; c->done = 1; c->skip_solve = c->skip_quad = c->skip_eval = c->skip_jac = 1; }
// o = Plus(xx, d): tangent d[N] = [15 per frame: pose 6, speed-bias 9 | ex 6 | td | F]
__device__ void tr_plus(const LwWin &a, const double *xx, const double *d, double sgn, double *o) {
    const int tid = threadIdx.x, NF = a.NF;
    for (int i = tid; i < NF; i += TR_T) {
        double dd[6];
        for (int k = 0; k < 6; k++) dd[k] = sgn * d[15 * i + k];
        pose_plus(xx + 7 * i, dd, o + 7 * i);
        for (int k = 0; k < 9; k++) o[7 * NF + 9 * i + k] = xx[7 * NF + 9 * i + k] + sgn * d[15 * i + 6 + k];
    }
    for (int f = tid; f < a.F; f += TR_T) o[16 * NF + f] = xx[16 * NF + f] + (a.fconst[f] ? 0.0 : sgn * d[a.P + f]);
    if (tid == 0) {
        if (a.est_ex) { double dd[6]; for (int k = 0; k < 6; k++) dd[k] = sgn * d[a.cEx + k]; pose_plus(xx + a.xo, dd, o + a.xo); }
        else for (int k = 0; k < 7; k++) o[a.xo + k] = xx[a.xo + k];
        o[a.xo + 7] = a.est_td ? xx[a.xo + 7] + sgn * d[a.cTd] : xx[a.xo + 7];
    }
}
// sum over the entries Ceres counts in ||x||: every frame block, the free features, Ex_Pose / td when they are estimated. fn(i) = the term of entry i
template <class Fn>
__device__ __forceinline__ double tr_state_sum(const LwWin &a, Fn fn, double *s_red) {
    double s = 0;
    for (int i = threadIdx.x; i < 16 * a.NF; i += TR_T) s += fn(i);
    for (int f = threadIdx.x; f < a.F; f += TR_T) if (!a.fconst[f]) s += fn(16 * a.NF + f);
    if (threadIdx.x == 0) {
        if (a.est_ex) for (int k = 0; k < 7; k++) s += fn(a.xo + k);
        if (a.est_td) s += fn(a.xo + 7);
    }
    return tr_sum(s, s_red);
}
// marginalization_factor.cpp:345-363: dx of the prior's blocks at the state xx
__device__ void tr_prior_dx(const LwWin &a, const double *xx) {
    const int bk = threadIdx.x;
    if (bk >= a.pnb) return;
    const int id = a.phdr[3 + bk], idx = a.phdr[51 + bk], NF = a.NF;
    const double *x0 = a.px0 + 9 * bk;
    auto pose_dx = [&](const double *xb, double *d) {
        for (int k = 0; k < 3; k++) d[k] = xb[k] - x0[k];
        const Q dq = q_mul(q_inv(q_load(x0 + 3)), q_load(xb + 3));
        const double sg = dq.w >= 0 ? 2.0 : -2.0;
        d[3] = sg * dq.x; d[4] = sg * dq.y; d[5] = sg * dq.z;
    };
    if (id < NF) pose_dx(xx + 7 * id, a.pdx + idx);
    else if (id < 2 * NF) for (int k = 0; k < 9; k++) a.pdx[idx + k] = xx[7 * NF + 9 * (id - NF) + k] - x0[k];
    else if (id == 2 * NF) pose_dx(xx + a.xo, a.pdx + idx);
    else if (id == 2 * NF + 1) a.pdx[idx] = xx[a.xo + 7] - x0[0];
}
__global__ __launch_bounds__(TR_T) void lw_tr_init(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    for (int i = threadIdx.x; i < VB_PRIOR_LD; i += TR_T) a.pdx[i] = 0.0;
    __syncthreads();
    if (a.pn) tr_prior_dx(a, a.x);
    const double *x = a.x;
    const double xs = tr_state_sum(a, [&](int i) { return x[i] * x[i]; }, s_red);
    if (threadIdx.x == 0) {
        c->radius = 1e4; c->mu = 1e-8; c->alpha = 0; c->step_norm = 0; c->x_cost = 0; c->cand_cost = 0; c->x_norm = sqrt(xs); c->gmax = 0; c->model_change = 0; c->initial_cost = 0; c->gg = 0;
        c->iteration = 0; c->reuse = 0; c->done = 0; c->termination = VILF_TERM_NO_CONVERGENCE; c->num_successful = 0; c->num_linear_solves = 0; c->consecutive_invalid = 0; c->fallback = 0;
        c->skip_solve = 1; c->skip_quad = 1; c->skip_eval = 1; c->skip_jac = 0; c->scaling_ready = 0; c->pad_ = 0;
    }
}
// after a linearisation at x (eval_grad_jac of the host loop): cost, diagonal / gradient pieces, the Jacobi scaling (fixed by the first linearisation),
// gradient_max_norm = || x - Plus(x, -g) ||_inf with the unscaled gradient; lw_scale follows with a.scale
__global__ __launch_bounds__(TR_T) void lw_tr_post(const LwWin *ws, int first) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    if (c->skip_jac) return;
    const int tid = threadIdx.x, P = a.P, N = a.N;
    for (int i = tid; i < N; i += TR_T) {
        double d, gg;
        if (i < P) { d = a.Hpp[(size_t)i * (P + 1)]; gg = a.gp[i]; }
        else { const int f = i - P; const bool cst = a.fconst[f]; d = cst ? 0.0 : a.hf[f]; gg = cst ? 0.0 : a.gf[f]; }
        a.diagH[i] = d; a.g[i] = gg;
    }
    __syncthreads();
    if (!c->scaling_ready) for (int i = tid; i < N; i += TR_T) a.scale[i] = 1.0 / (1.0 + sqrt(a.diagH[i]));
    tr_plus(a, a.x, a.g, -1.0, a.cand);                 // cand is free between a decision and the next step
    __syncthreads();
    double m = 0;
    for (int i = tid; i < a.xo + 8; i += TR_T) m = fmax(m, fabs(a.x[i] - a.cand[i]));
    m = tr_max(m, s_red);
    for (int i = tid; i < N; i += TR_T) { const double sc = a.scale[i]; a.g[i] *= sc; a.diagH[i] *= sc * sc; }
    const double cost_now = tr_cost_sum(a, s_red);
    if (tid == 0) { a.scal[0] = cost_now; c->gmax = m; c->x_cost = cost_now; if (first) c->initial_cost = cost_now; c->scaling_ready = 1; }
}
// the host loop's evaluations: the cost partials summed into scal[0]
__global__ __launch_bounds__(TR_T) void lw_cost_sum(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    const double c = tr_cost_sum(a, s_red);
    if (threadIdx.x == 0) a.scal[0] = c;
}
// top of an iteration (the tests of trust_region_minimizer.cc before a step) and, unless the last linear solve is reused, the vectors the dogleg needs
__global__ __launch_bounds__(TR_T) void lw_tr_begin(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    if (c->done) return;
    const int tid = threadIdx.x, N = a.N;
    const bool stop_it = c->iteration >= a.max_it, stop_g = c->gmax <= 1e-10, stop_r = c->radius <= 1e-32;
    const int reuse = c->reuse;
    __syncthreads();
    if (stop_it || stop_g || stop_r) { if (tid == 0) tr_finish(c, stop_it ? VILF_TERM_NO_CONVERGENCE : stop_g ? VILF_TERM_CONVERGENCE_GRADIENT : VILF_TERM_FAILURE); return; }
    double gg = 0;
    if (!reuse) for (int i = tid; i < N; i += TR_T) {
        const double dg = sqrt(fmin(fmax(a.diagH[i], 1e-6), 1e32)), gr = a.g[i] / dg;
        a.diagonal[i] = dg; a.gradient[i] = gr; a.vec[i] = gr / dg;
        gg += gr * gr;
    }
    gg = tr_sum(gg, s_red);
    if (tid == 0) {
        c->iteration++;
        c->skip_solve = reuse; c->skip_quad = 0; c->skip_eval = 0; c->skip_jac = 1;
        if (!reuse) { c->gg = gg; c->reuse = 1; }
    }
}
// alpha = |gradient|^2 / |J gradient|^2 (Cauchy point) from the products lw_rowdot left in tmpP / tmpF; the LM diagonal of this solve goes to vec
__global__ __launch_bounds__(TR_T) void lw_tr_alpha(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    if (c->skip_solve) return;
    const int tid = threadIdx.x, P = a.P, N = a.N;
    double q = 0;
    for (int i = tid; i < N; i += TR_T) {
        const double v = a.vec[i];
        if (i < P) q += v * a.tmpP[i];
        else if (!a.fconst[i - P]) q += v * (2.0 * a.tmpF[i - P] + a.diagH[i] * v);
    }
    q = tr_sum(q, s_red);
    const double smu = sqrt(c->mu);
    double bad = 0;
    for (int i = tid; i < N; i += TR_T) {
        const double l = a.diagonal[i] * smu;
        a.vec[i] = l;
        if (i >= P && !a.fconst[i - P] && !(a.diagH[i] + l * l > 0.0)) bad = 1;
    }
    bad = tr_max(bad, s_red);
    if (tid == 0) {
        c->alpha = c->gg / q; c->num_linear_solves++;
        if (bad > 0) { tr_finish(c, VILF_TERM_FAILURE); c->fallback = 1; }
    }
}
// the Gauss-Newton step from the linear solve (or the one kept from the last solve) and the traditional dogleg step for the current radius; step -> vec
__global__ __launch_bounds__(TR_T) void lw_tr_step(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    if (c->done) return;
    const int tid = threadIdx.x, P = a.P, N = a.N;
    if (!c->skip_solve) {
        double bad = (*a.info != 0) ? 1.0 : 0.0;
        for (int i = tid; i < N; i += TR_T) {
            const double y = i < P ? a.rhs[i] : (a.fconst[i - P] ? 0.0 : a.yf[i - P]);
            if (!isfinite(y)) bad = 1;
            a.gn[i] = y * -a.diagonal[i];
        }
        bad = tr_max(bad, s_red);
        if (bad > 0) { if (tid == 0) { tr_finish(c, VILF_TERM_FAILURE); c->fallback = 1; } return; }     // a failed factorisation: mu is raised by the host loop
    }
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = tid; i < N; i += TR_T) { const double gr = a.gradient[i], n = a.gn[i]; s0 += gr * gr; s1 += n * n; s2 += gr * n; }
    s0 = tr_sum(s0, s_red); s1 = tr_sum(s1, s_red); s2 = tr_sum(s2, s_red);
    const double gradient_norm = sqrt(s0), gn_norm = sqrt(s1), radius = c->radius, alpha = c->alpha;
    double nrm;
    if (gn_norm <= radius) { for (int i = tid; i < N; i += TR_T) a.step[i] = a.gn[i] / a.diagonal[i]; nrm = gn_norm; }
    else if (gradient_norm * alpha >= radius) { for (int i = tid; i < N; i += TR_T) a.step[i] = -(radius / gradient_norm) * a.gradient[i] / a.diagonal[i]; nrm = radius; }
    else {
        const double b_dot_a = -alpha * s2, a_sq = pow(alpha * gradient_norm, 2.0), bma = a_sq - 2 * b_dot_a + pow(gn_norm, 2.0);
        const double cc = b_dot_a - a_sq, dd = sqrt(cc * cc + bma * (pow(radius, 2.0) - a_sq));
        const double beta = (cc <= 0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
        double nn = 0;
        for (int i = tid; i < N; i += TR_T) { const double st = (-alpha * (1.0 - beta)) * a.gradient[i] + beta * a.gn[i]; a.step[i] = st; nn += st * st; }
        nn = tr_sum(nn, s_red);
        nrm = sqrt(nn);
        for (int i = tid; i < N; i += TR_T) a.step[i] /= a.diagonal[i];
    }
    __syncthreads();
    for (int i = tid; i < N; i += TR_T) a.vec[i] = a.step[i];
    if (tid == 0) c->step_norm = nrm;
}
// model_cost_change from step^T H step (lw_rowdot products), the candidate Plus(x, step * scale), its prior dx, the parameter-tolerance test
__global__ __launch_bounds__(TR_T) void lw_tr_model(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    LwCtl *c = a.ctl;
    if (c->done) return;
    const int tid = threadIdx.x, P = a.P, N = a.N;
    if (tid == 0) a.scal[0] = 0.0;                              // the cost at the candidate accumulates here (the evaluation that follows)
    double q = 0, gs = 0;
    for (int i = tid; i < N; i += TR_T) {
        const double v = a.step[i];
        gs += a.g[i] * v;
        if (i < P) q += v * a.tmpP[i];
        else if (!a.fconst[i - P]) q += v * (2.0 * a.tmpF[i - P] + a.diagH[i] * v);
    }
    q = tr_sum(q, s_red); gs = tr_sum(gs, s_red);
    const double model = -(gs + 0.5 * q);
    if (!(model > 0.0)) {                                       // step_is_invalid: mu up, solve again next iteration
        if (tid == 0) {
            c->consecutive_invalid++; c->mu *= 10.0; c->reuse = 0; c->skip_eval = 1; c->skip_jac = 1;
            if (c->consecutive_invalid >= 5) tr_finish(c, VILF_TERM_FAILURE);
        }
        return;
    }
    // delta = step * scale in place of vec (no launch reads vec before the next lw_tr_begin / lw_tr_step rewrites it)
    for (int i = tid; i < N; i += TR_T) a.vec[i] = a.step[i] * a.scale[i];
    __syncthreads();
    tr_plus(a, a.x, a.vec, 1.0, a.cand);
    __syncthreads();
    if (a.pn) tr_prior_dx(a, a.cand);
    const double *x = a.x, *cd = a.cand;
    const double sn = tr_state_sum(a, [&](int i) { const double d = x[i] - cd[i]; return d * d; }, s_red);
    if (tid == 0) {
        c->consecutive_invalid = 0; c->model_change = model;
        if (sqrt(sn) <= 1e-8 * (c->x_norm + 1e-8)) tr_finish(c, VILF_TERM_CONVERGENCE_PARAMETER);
    }
}
// the cost at the candidate is in scal[0]: function tolerance, relative decrease, accept / reject, radius and mu updates
__global__ __launch_bounds__(TR_T) void lw_tr_decide(const LwWin *ws) {
    const LwWin &a = ws[blockIdx.z];
    __shared__ double s_red[TR_T / 64];
    __shared__ int s_acc;
    LwCtl *c = a.ctl;
    if (c->done || c->skip_eval) return;
    const int tid = threadIdx.x;
    const double cost_now = tr_cost_sum(a, s_red);              // (its barriers also mean: every wave has read the flags before thread 0 may change them)
    __syncthreads();
    if (tid == 0) {
        const double cand_cost = cost_now, cost_change = c->x_cost - cand_cost;
        a.scal[0] = cost_now;
        int acc = 0;
        c->cand_cost = cand_cost;
        if (fabs(cost_change) <= 1e-6 * c->x_cost) tr_finish(c, VILF_TERM_CONVERGENCE_FUNCTION);
        else {
            const double rd = cost_change / c->model_change;
            if (rd > 1e-3) {
                acc = 1;
                if (c->iteration < a.max_it) c->skip_jac = 0; else c->x_cost = cand_cost;      // the budget is spent: nothing would use the linearisation at the accepted point
                c->num_successful++;
                if (rd < 0.25) c->radius *= 0.5;
                if (rd > 0.75) c->radius = fmax(c->radius, 3.0 * c->step_norm);
                c->mu = fmax(1e-8, 2.0 * c->mu / 10.0);
                c->reuse = 0;
            } else { c->radius *= 0.5; c->reuse = 1; }
        }
        s_acc = acc;
    }
    __syncthreads();
    if (!s_acc) return;
    for (int i = tid; i < a.xo + 8; i += TR_T) a.x[i] = a.cand[i];
    __syncthreads();
    const double *x = a.x;
    const double xs = tr_state_sum(a, [&](int i) { return x[i] * x[i]; }, s_red);
    if (tid == 0) c->x_norm = sqrt(xs);
}
}  // namespace

// The blocked Cholesky of this file as a service to the other translation units (the pose graph's loop-closure block): S is (n + 1) x n, row-major, rows 0 .. n - 1 the
// symmetric positive-definite matrix (lower triangle read, overwritten by L), row n the right-hand side; y (n) receives the solution. *info (device, zeroed by the caller)
// is set to 1 when a pivot is not positive. Everything is enqueued on the handle's stream.
// lw_chol_back keeps the solution vector in LDS (n doubles of dynamic LDS beside 33.8 KB of static staging): 12288 x 8 + 33.8 KB = 130 KB of the CU's 160 KB
#ifdef VILF_LW_STAMPS
extern "C" int vilf_debug_lw_stamps(long long *out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(lw_dbg_stamps), sizeof(long long) * 64) == hipSuccess ? 0 : -1; }
#endif
int vilf_lw_chol_max_n() { return 12288; }
static int lw_chol_back_attr(vilf_handle *h) {     // above 64 KB in all, a launch needs the attribute; set once per process (handles on several host threads may get here together), for the largest supported n
    static std::once_flag once;
    static hipError_t e1 = hipSuccess, e2 = hipSuccess;
    std::call_once(once, []() {
        e1 = hipFuncSetAttribute(reinterpret_cast<const void *>(lw_chol_back_raw), hipFuncAttributeMaxDynamicSharedMemorySize, vilf_lw_chol_max_n() * 8);
        e2 = hipFuncSetAttribute(reinterpret_cast<const void *>(lw_chol_back), hipFuncAttributeMaxDynamicSharedMemorySize, vilf_lw_chol_max_n() * 8);
    });
    HIPCHECK(h, e1);
    HIPCHECK(h, e2);
    return VILF_OK;
}
int vilf_lw_chol_solve(vilf_handle *h, int n, double *S, double *y, int *info) {
    if (n < 1 || n > vilf_lw_chol_max_n()) { h->err = "vilf_lw_chol_solve: dimension outside the supported range (1 .. 12288)"; return VILF_ERR_UNSUPPORTED; }
    const int rca = lw_chol_back_attr(h);
    if (rca != VILF_OK) return rca;
    for (int j0 = 0; j0 < n; j0 += CH_NB) {
        const int nb = std::min(CH_NB, n - j0), below = n + 1 - (j0 + nb), npanel = std::max(1, (below + CH_BELOW - 1) / CH_BELOW);
        if (j0 == 0) hipLaunchKernelGGL(lw_chol_panel_raw, dim3(npanel), dim3(256), 0, h->stream, n, S, info);
        else {
            const int nt = (n + 1 - j0 + 63) / 64;
            hipLaunchKernelGGL(lw_chol_step_raw, dim3(npanel + nt * (nt - 1) / 2), dim3(256), 0, h->stream, n, S, j0, info);
        }
    }
    hipLaunchKernelGGL(lw_chol_back_raw, dim3(1), dim3(1024), (size_t)n * 8, h->stream, n, S, y);
    HIPCHECK(h, hipGetLastError());
    return VILF_OK;
}

namespace {
// ---- one window of a group on the host -------------------------------------------------------------------------------------------
struct LwHostWin {
    const vilf_window_in *in = nullptr;
    vilf_window_out *out = nullptr;
    bool resident = false;             // also a slot of the 11-frame batch: its prior applies, the solved state goes back into the batch buffers
    size_t slot = 0;
    int NF = 0, F = 0, P = 0, N = 0, nvis = 0, nimu = 0, cEx = -1, cTd = -1, PC = 0, WS = 0, nchunk = 0, npairs_cap = 0, nslots_cap = 0, npairs = 0;
    size_t xo = 0;
    bool use_lidar = false;
    int pn = 0, pnb = 0, phdr[VB_PRIOR_HDR];
    double px0[24 * 9];
    std::vector<double> x;             // the state: pose | sb | feat | ex[7] | td — feat in the DEVICE's feature order (by start frame) until the results are handed out
    std::vector<int> fdev;             // host feature index -> device feature index
    std::vector<unsigned char> fc;     // feature_const in device order
    LwWin dw;                          // the descriptor (device pointers)
    size_t o_vis = 0, o_tdr = 0, o_fconst = 0, o_scal = 0, o_pcol = 0, o_lid = 0, o_kspan = 0, o_fvis = 0, o_fidx = 0, o_cslot = 0, o_prt = 0, o_froff = 0, o_frlist = 0, o_fspan = 0;       // offsets of the inputs in the staging image (imu / cov / x: group-wide runs)
    // results of the device loop
    LwCtl hc;
};
// launch shapes for a set of windows (the whole group, or one window for the host loop): every grid is sized for the largest window, the others' surplus
// workgroups return at once
struct LwDims {
    int G = 0, maxP = 0, maxF = 0, maxNvis = 0, maxNimu = 0, maxPC = 0, maxWS = 0, maxNF = 0, maxNpairs = 0, nks = 4;
    bool any_prior = false;
    void take(const LwHostWin &w) {
        G++; maxP = std::max(maxP, w.P); maxF = std::max(maxF, w.F); maxNvis = std::max(maxNvis, w.nvis); maxNimu = std::max(maxNimu, w.nimu);
        nks = w.dw.nks;
        maxPC = std::max(maxPC, w.PC); maxWS = std::max(maxWS, w.WS); maxNF = std::max(maxNF, w.NF); maxNpairs = std::max(maxNpairs, w.npairs_cap);
        any_prior = any_prior || w.pn != 0;
    }
};
struct LwEnq {
    vilf_handle *h; LwCtx *c; const LwWin *ws; LwDims d; bool ext, prof;
    bool cleared_full = false;         // the first linearisation of a solve clears all of Hpp, the later ones its band
    void tic() { if (prof) hipEventRecord(c->ev[0], h->stream); }
    void toc(int grp) { if (prof) { hipEventRecord(c->ev[1], h->stream); hipEventSynchronize(c->ev[1]); float t = 0; hipEventElapsedTime(&t, c->ev[0], c->ev[1]); c->ms[grp] += t; c->launches[grp] += 1; } }
    dim3 grid(size_t gx, unsigned gy = 1) const { return dim3((unsigned)std::max<size_t>(gx, 1), gy, (unsigned)d.G); }
    // one evaluation at the device state x (which = 0) or cand (1), the prior's dx already in pdx. Cost only (jac = 0): the caller has zeroed scal[0]
    void evaluate(int which, int jac, int sk) {
        const size_t sP = d.maxP;
        if (jac) {      // one launch clears the cost and the five accumulation targets
            hipLaunchKernelGGL(lw_clear, grid(sP + 1), dim3(256), 0, h->stream, ws, cleared_full ? 0 : 1, sk);
            cleared_full = true;
            tic();
        }
        if (d.maxNvis && jac) {                       // the rows of W, h_f, g_f: feature-major
            // windows of up to 17 frames (tracks of up to 16 factors): four features per wave
            const bool g16 = d.maxNF <= 17 && !std::getenv("VILF_LW_FEATURE_WAVES");      // (test hook: a wave per feature for every window size — the two forms give the same bits)
            const int fpb = g16 ? 16 : 4;               // features per workgroup
            const size_t lds = (size_t)fpb * d.maxWS * 8;
            if (ext) { if (g16) hipLaunchKernelGGL((lw_feature_rows<true, 16>), grid((d.maxF + fpb - 1) / fpb), dim3(256), lds, h->stream, ws, which, sk);
                       else hipLaunchKernelGGL((lw_feature_rows<true, 64>), grid((d.maxF + fpb - 1) / fpb), dim3(256), lds, h->stream, ws, which, sk); }
            else { if (g16) hipLaunchKernelGGL((lw_feature_rows<false, 16>), grid((d.maxF + fpb - 1) / fpb), dim3(256), lds, h->stream, ws, which, sk);
                   else hipLaunchKernelGGL((lw_feature_rows<false, 64>), grid((d.maxF + fpb - 1) / fpb), dim3(256), lds, h->stream, ws, which, sk); }
        }
        if (d.maxNvis) {
            if (ext) hipLaunchKernelGGL(lw_visual_ext, grid((d.maxNvis + LW_CH - 1) / LW_CH), dim3(LW_CH), 0, h->stream, ws, which, jac, sk);
            else {
                hipLaunchKernelGGL(lw_visual, grid((d.maxNvis + LW_CH - 1) / LW_CH), dim3(LW_CH), 0, h->stream, ws, which, jac, sk);
                if (jac) hipLaunchKernelGGL(lw_assemble, grid(d.maxNF + d.maxNpairs), dim3(256), 0, h->stream, ws, sk);
            }
        }
        if (ext && d.any_prior) hipLaunchKernelGGL(lw_imu_lidar_prior, grid(d.maxNF + 1), dim3(256), 0, h->stream, ws, which, jac, sk);
        else {
            if (d.any_prior) hipLaunchKernelGGL(lw_prior, grid(1), dim3(256), 0, h->stream, ws, jac, sk);
            hipLaunchKernelGGL(lw_imu_lidar, grid(d.maxNF), dim3(256), 0, h->stream, ws, which, jac, sk);
        }
        if (jac) toc(0);
    }
    // Hpp v_p -> tmpP, W_f . v_p -> tmpF for the vector in vec
    void quad(int sk) { hipLaunchKernelGGL(lw_rowdot, grid((d.maxP + d.maxF + 3) / 4), dim3(256), 0, h->stream, ws, 0, sk); }
    void scale(int src, int sk) {
        hipLaunchKernelGGL(lw_scale, grid((size_t)d.maxP + (d.maxF + 3) / 4), dim3(256), 0, h->stream, ws, src, sk);
    }
    // one linear solve (H' + lm^2) y = g' with lm in vec: y_p -> rhs, y_f -> yf, the Cholesky's status -> info
    void linear_solve(int sk) {
        const size_t sP = d.maxP;
        if (d.maxF) {
            hipLaunchKernelGGL(lw_den, grid((d.maxF + 255) / 256), dim3(256), 0, h->stream, ws, sk);
            // the Schur reduce over the compact columns: K-split partial tiles of Wn^T Wn (+ Wn^T g_f / sqrt(den) in the diagonal tiles), summed by lw_schur_prep
            tic();
            const int nt = (d.maxPC + 63) / 64;
            hipLaunchKernelGGL(lw_syrk_mfma, grid(nt * (nt + 1) / 2, d.nks), dim3(256), 0, h->stream, ws, sk);
            toc(1);
        }
        hipLaunchKernelGGL(lw_schur_prep, grid(sP + 1), dim3(256), 0, h->stream, ws, sk);
        tic();
        for (int j0 = 0; j0 < d.maxP; j0 += CH_NB) {                     // blocked Cholesky, one launch per 64-column block (panel of this column + the rest of the previous column's update)
            const int nb = std::min(CH_NB, d.maxP - j0), below = d.maxP + 1 - (j0 + nb), npanel = std::max(1, (below + CH_BELOW - 1) / CH_BELOW);
            if (j0 == 0) hipLaunchKernelGGL(lw_chol_panel, grid(npanel), dim3(256), 0, h->stream, ws, sk);
            else {
                const int nt = (d.maxP + 1 - j0 + 63) / 64;
                hipLaunchKernelGGL(lw_chol_step, grid(npanel + nt * (nt - 1) / 2), dim3(256), 0, h->stream, ws, j0, sk);
            }
        }
        hipLaunchKernelGGL(lw_chol_back, grid(1), dim3(1024), (size_t)d.maxP * 8, h->stream, ws, sk);
        toc(2);
        if (d.maxF) {
            hipLaunchKernelGGL(lw_rowdot, grid((d.maxF + 3) / 4), dim3(256), 0, h->stream, ws, 1, sk);       // y_f = (g_f - W_f . y_p) / den_f
        }
    }
};

// ---- the trust-region loop on the host (trust_region_minimizer.cc with the traditional dogleg strategy, dogleg_strategy.cc): the path of a wall-clock limit
// (Ceres tests the clock at the top of every iteration) and the fallback of a failed factorisation in the device loop. One window; `dws` = its descriptor on the device.
int lw_host_loop(vilf_handle *h, LwCtx *c, LwHostWin &hw, const LwWin *dws, double tlim, const std::chrono::steady_clock::time_point t_start, bool ext) {
    (void)hw.in;
    const LwWin &dw = hw.dw;
    const int NF = hw.NF, F = hw.F, P = hw.P, N = hw.N, cEx = hw.cEx, cTd = hw.cTd, pn = hw.pn, pnb = hw.pnb;
    const size_t xo = hw.xo, sP = P, sN = N;
    const bool est_ex = cEx >= 0, est_td = cTd >= 0;
    LwEnq q{h, c, dws, LwDims(), ext, h->profiling != 0};
    q.d.take(hw);
    std::vector<double> &x = hw.x;
    std::vector<double> cand(x.size()), pdx(VB_PRIOR_LD, 0.0);
    auto prior_dx = [&](const std::vector<double> &xx) {                 // marginalization_factor.cpp:345-363
        auto pose_dx = [&](const double *xb, const double *x0, double *d) {
            for (int k = 0; k < 3; k++) d[k] = xb[k] - x0[k];
            const double n0 = x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5] + x0[6] * x0[6];
            const HQ qi{-x0[3] / n0, -x0[4] / n0, -x0[5] / n0, x0[6] / n0}, dq = hq_mul(qi, HQ{xb[3], xb[4], xb[5], xb[6]});
            const double sgn = dq.w >= 0 ? 2.0 : -2.0;
            d[3] = sgn * dq.x; d[4] = sgn * dq.y; d[5] = sgn * dq.z;
        };
        for (int bk = 0; bk < pnb; bk++) {
            const int id = hw.phdr[3 + bk], idx = hw.phdr[51 + bk];
            const double *x0 = &hw.px0[9 * bk];
            if (id < NF) pose_dx(&xx[7 * id], x0, &pdx[idx]);
            else if (id < 2 * NF) for (int k = 0; k < 9; k++) pdx[idx + k] = xx[7 * NF + 9 * (id - NF) + k] - x0[k];
            else if (id == 2 * NF) pose_dx(&xx[xo], x0, &pdx[idx]);
            else if (id == 2 * NF + 1) pdx[idx] = xx[xo + 7] - x0[0];
        }
    };
    auto xnorm = [&](const std::vector<double> &v) {
        double s = 0;
        for (int i = 0; i < 16 * NF; i++) s += v[i] * v[i];
        for (int f = 0; f < F; f++) if (!hw.fc[f]) s += v[16 * NF + f] * v[16 * NF + f];
        if (est_ex) for (int k = 0; k < 7; k++) s += v[xo + k] * v[xo + k];
        if (est_td) s += v[xo + 7] * v[xo + 7];
        return std::sqrt(s);
    };
    // tangent vector d[N] = [15 per frame: pose 6, speed-bias 9 | ex 6 | td | F] applied to the state
    auto plus = [&](const std::vector<double> &xx, const std::vector<double> &d, std::vector<double> &o) {
        o = xx;
        for (int i = 0; i < NF; i++) { h_pose_plus(&xx[7 * i], &d[15 * i], &o[7 * i]); for (int k = 0; k < 9; k++) o[7 * NF + 9 * i + k] = xx[7 * NF + 9 * i + k] + d[15 * i + 6 + k]; }
        for (int f = 0; f < F; f++) o[16 * NF + f] = xx[16 * NF + f] + (hw.fc[f] ? 0.0 : d[P + f]);
        if (est_ex) h_pose_plus(&xx[xo], &d[cEx], &o[xo]);
        if (est_td) o[xo + 7] = xx[xo + 7] + d[cTd];
    };
    auto evaluate = [&](const std::vector<double> &xx, bool jac, double &cost) -> int {
        HIPCHECK(h, hipMemcpyAsync(dw.x, xx.data(), xx.size() * 8, hipMemcpyHostToDevice, h->stream));
        if (pn) {
            prior_dx(xx);
            HIPCHECK(h, hipMemcpyAsync(dw.pdx, pdx.data(), VB_PRIOR_LD * 8, hipMemcpyHostToDevice, h->stream));
        }
        if (!jac) HIPCHECK(h, hipMemsetAsync(dw.scal, 0, 8, h->stream));
        q.evaluate(0, jac ? 1 : 0, LW_SK_NONE);
        hipLaunchKernelGGL(lw_cost_sum, dim3(1, 1, 1), dim3(TR_T), 0, h->stream, dws);
        HIPCHECK(h, hipGetLastError());
        HIPCHECK(h, hipMemcpyAsync(&cost, dw.scal, 8, hipMemcpyDeviceToHost, h->stream));
        if (!jac) HIPCHECK(h, hipStreamSynchronize(h->stream));         // with the Jacobians: fetch_diag_grad follows and waits once for both
        return VILF_OK;
    };
    // host copies of the (scaled) diagonal / gradient pieces
    const size_t sF = std::max(F, 1);
    std::vector<double> g(N), scale(N, 1.0), diagH(N), hfh(sF), diagonal(N), gradient(N), gn(N), step(N), delta(N), lm(N), y(N), tP(P), tF(sF), v(N);
    bool scaling_ready = false;
    double gradient_max_norm = 0, x_cost = 0;
    auto fetch_diag_grad = [&]() -> int {            // diag(Hpp) / h_f and g_p / g_f from the device
        HIPCHECK(h, hipMemcpy2DAsync(diagH.data(), 8, dw.Hpp, (sP + 1) * 8, 8, P, hipMemcpyDeviceToHost, h->stream));     // only the diagonal of Hpp: strided copy
        if (F) { HIPCHECK(h, hipMemcpyAsync(&diagH[P], dw.hf, (size_t)F * 8, hipMemcpyDeviceToHost, h->stream)); HIPCHECK(h, hipMemcpyAsync(&g[P], dw.gf, (size_t)F * 8, hipMemcpyDeviceToHost, h->stream)); }
        HIPCHECK(h, hipMemcpyAsync(&g[0], dw.gp, sP * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        for (int f = 0; f < F; f++) if (hw.fc[f]) { diagH[P + f] = 0.0; g[P + f] = 0.0; }
        return VILF_OK;
    };
    // x^T H x with the (scaled) blocks on the device: v_p^T Hpp v_p + 2 sum_f v_f (W_f . v_p) + sum_f h_f v_f^2
    auto quad = [&](const std::vector<double> &vv, double &qq) -> int {
        HIPCHECK(h, hipMemcpyAsync(dw.vec, vv.data(), sN * 8, hipMemcpyHostToDevice, h->stream));
        q.quad(LW_SK_NONE);
        HIPCHECK(h, hipMemcpyAsync(tP.data(), dw.tmpP, sP * 8, hipMemcpyDeviceToHost, h->stream));
        if (F) HIPCHECK(h, hipMemcpyAsync(tF.data(), dw.tmpF, (size_t)F * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        qq = 0;
        for (int i = 0; i < P; i++) qq += vv[i] * tP[i];
        for (int f = 0; f < F; f++) if (!hw.fc[f]) qq += vv[P + f] * (2.0 * tF[f] + hfh[f] * vv[P + f]);
        return VILF_OK;
    };
    auto eval_grad_jac = [&]() -> int {
        int rc = evaluate(x, true, x_cost);
        if (rc != VILF_OK) return rc;
        if ((rc = fetch_diag_grad()) != VILF_OK) return rc;           // unscaled
        if (!scaling_ready) { for (int i = 0; i < N; i++) scale[i] = 1.0 / (1.0 + std::sqrt(diagH[i])); scaling_ready = true; }
        // gradient_max_norm = || x - Plus(x, -g) ||_inf with the unscaled gradient
        std::vector<double> ng(N), proj;
        for (int i = 0; i < N; i++) ng[i] = -g[i];
        plus(x, ng, proj);
        gradient_max_norm = 0;
        for (size_t i = 0; i < x.size(); i++) gradient_max_norm = std::max(gradient_max_norm, std::fabs(x[i] - proj[i]));
        // Jacobi scaling of the blocks on the device and of the host copies
        HIPCHECK(h, hipMemcpyAsync(dw.vec, scale.data(), sN * 8, hipMemcpyHostToDevice, h->stream));
        q.scale(1, LW_SK_NONE);
        for (int i = 0; i < N; i++) { g[i] *= scale[i]; diagH[i] *= scale[i] * scale[i]; }
        for (int f = 0; f < F; f++) hfh[f] = diagH[P + f];
        return VILF_OK;
    };
    // ---- the linear solve: (H' + lm^2) y = g'
    auto linear_solve = [&](bool &ok) -> int {
        ok = false;
        for (int f = 0; f < F; f++) if (!hw.fc[f] && !(hfh[f] + lm[P + f] * lm[P + f] > 0.0)) return VILF_OK;
        HIPCHECK(h, hipMemcpyAsync(dw.vec, lm.data(), sN * 8, hipMemcpyHostToDevice, h->stream));
        q.linear_solve(LW_SK_NONE);
        HIPCHECK(h, hipGetLastError());
        int info = 0;
        HIPCHECK(h, hipMemcpyAsync(&info, dw.info, 4, hipMemcpyDeviceToHost, h->stream));      // read with the solution below: one wait per linear solve
        if (F) HIPCHECK(h, hipMemcpyAsync(&y[P], dw.yf, (size_t)F * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipMemcpyAsync(&y[0], dw.rhs, sP * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        if (info != 0) return VILF_OK;                                  // not positive definite: the caller raises mu
        for (int f = 0; f < F; f++) if (hw.fc[f]) y[P + f] = 0.0;
        for (double a : y) if (!std::isfinite(a)) return VILF_OK;
        ok = true;
        return VILF_OK;
    };
    const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32, min_relative_decrease = 1e-3, function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    double radius = 1e4, mu = 1e-8, alpha = 0, dogleg_step_norm = 0;
    bool reuse = false;
    int iteration = 0, consecutive_invalid = 0, num_successful = 0, num_linear_solves = 0, termination = VILF_TERM_NO_CONVERGENCE;
    auto vdotN = [&](const std::vector<double> &a, const std::vector<double> &b) { double s = 0; for (int i = 0; i < N; i++) s += a[i] * b[i]; return s; };
    auto traditional = [&]() {
        const double gradient_norm = std::sqrt(vdotN(gradient, gradient)), gn_norm = std::sqrt(vdotN(gn, gn));
        if (gn_norm <= radius) { for (int i = 0; i < N; i++) step[i] = gn[i] / diagonal[i]; dogleg_step_norm = gn_norm; return; }
        if (gradient_norm * alpha >= radius) { for (int i = 0; i < N; i++) step[i] = -(radius / gradient_norm) * gradient[i] / diagonal[i]; dogleg_step_norm = radius; return; }
        const double b_dot_a = -alpha * vdotN(gradient, gn), a_sq = std::pow(alpha * gradient_norm, 2.0), bma = a_sq - 2 * b_dot_a + std::pow(gn_norm, 2);
        const double cc = b_dot_a - a_sq, dd = std::sqrt(cc * cc + bma * (std::pow(radius, 2.0) - a_sq));
        const double beta = (cc <= 0) ? (dd - cc) / bma : (radius * radius - a_sq) / (dd + cc);
        double nn = 0;
        for (int i = 0; i < N; i++) { step[i] = (-alpha * (1.0 - beta)) * gradient[i] + beta * gn[i]; nn += step[i] * step[i]; }
        dogleg_step_norm = std::sqrt(nn);
        for (int i = 0; i < N; i++) step[i] /= diagonal[i];
    };
    const int max_it = h->opts.max_num_iterations;
    int rc = eval_grad_jac();
    if (rc != VILF_OK) return rc;
    const double initial_cost = x_cost;
    double x_norm = xnorm(x);
    while (true) {
        if (tlim > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= tlim) { termination = VILF_TERM_NO_CONVERGENCE; break; }
        if (iteration >= max_it) { termination = VILF_TERM_NO_CONVERGENCE; break; }
        if (gradient_max_norm <= gradient_tolerance) { termination = VILF_TERM_CONVERGENCE_GRADIENT; break; }
        if (radius <= 1e-32) { termination = VILF_TERM_FAILURE; break; }
        iteration++;
        bool valid = true;
        if (reuse) traditional();
        else {
            reuse = true;
            for (int i = 0; i < N; i++) diagonal[i] = std::sqrt(std::min(std::max(diagH[i], min_lm_diagonal), max_lm_diagonal));
            for (int i = 0; i < N; i++) gradient[i] = g[i] / diagonal[i];
            for (int i = 0; i < N; i++) v[i] = gradient[i] / diagonal[i];
            double Jg2;
            if ((rc = quad(v, Jg2)) != VILF_OK) return rc;
            alpha = vdotN(gradient, gradient) / Jg2;
            bool ok = false;
            while (mu < 1.0) {
                for (int i = 0; i < N; i++) lm[i] = diagonal[i] * std::sqrt(mu);
                num_linear_solves++;
                if ((rc = linear_solve(ok)) != VILF_OK) return rc;
                if (ok) break;
                mu *= 10.0;
            }
            if (!ok) valid = false;
            else { for (int i = 0; i < N; i++) gn[i] = y[i] * -diagonal[i]; traditional(); }
        }
        double model_cost_change = 0;
        if (valid) {
            double sHs;
            if ((rc = quad(step, sHs)) != VILF_OK) return rc;
            model_cost_change = -(vdotN(g, step) + 0.5 * sHs);
            if (model_cost_change <= 0.0) valid = false;
        }
        if (!valid) {
            consecutive_invalid++;
            mu *= 10.0; reuse = false;                                   // step_is_invalid
            if (consecutive_invalid >= 5) { termination = VILF_TERM_FAILURE; break; }
            continue;
        }
        consecutive_invalid = 0;
        for (int i = 0; i < N; i++) delta[i] = step[i] * scale[i];
        plus(x, delta, cand);
        double cand_cost;
        if ((rc = evaluate(cand, false, cand_cost)) != VILF_OK) return rc;
        double sn = 0;
        for (int i = 0; i < 16 * NF; i++) sn += (x[i] - cand[i]) * (x[i] - cand[i]);
        for (int f = 0; f < F; f++) if (!hw.fc[f]) sn += (x[16 * NF + f] - cand[16 * NF + f]) * (x[16 * NF + f] - cand[16 * NF + f]);
        if (est_ex) for (int k = 0; k < 7; k++) sn += (x[xo + k] - cand[xo + k]) * (x[xo + k] - cand[xo + k]);
        if (est_td) sn += (x[xo + 7] - cand[xo + 7]) * (x[xo + 7] - cand[xo + 7]);
        if (std::sqrt(sn) <= parameter_tolerance * (x_norm + parameter_tolerance)) { termination = VILF_TERM_CONVERGENCE_PARAMETER; break; }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= function_tolerance * x_cost) { termination = VILF_TERM_CONVERGENCE_FUNCTION; break; }
        const double rd = cost_change / model_cost_change;
        if (rd > min_relative_decrease) {
            x = cand; x_norm = xnorm(x);
            if (iteration < max_it) { if ((rc = eval_grad_jac()) != VILF_OK) return rc; }
            else x_cost = cand_cost;                                     // the budget is spent: nothing would use the linearisation at the accepted point
            num_successful++;
            if (rd < 0.25) radius *= 0.5;                                // step_accepted
            if (rd > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
            mu = std::max(1e-8, 2.0 * mu / 10.0);
            reuse = false;
        } else { radius *= 0.5; reuse = true; }                          // step_rejected
    }
    LwCtl &hc = hw.hc;
    hc.iteration = iteration; hc.num_successful = num_successful; hc.num_linear_solves = num_linear_solves; hc.termination = termination;
    hc.initial_cost = initial_cost; hc.x_cost = x_cost; hc.radius = radius; hc.fallback = 0;
    return VILF_OK;
}
inline size_t lw_al(size_t b) { return (b + 255) & ~(size_t)255; }
}  // namespace

// G independent windows in one chain of launches. slot1[g] != 0: window g is also resident as slot slot1[g] - 1 of the 11-frame batch (vilf_batch_upload ran):
// that slot's prior applies and the solved state is written back into the batch buffers (the marginalization reads them). slot1 = nullptr: none is.
int vilf_lw_group_solve(vilf_handle *h, int G, const vilf_window_in *const *ins, vilf_window_out *const *outs, const int *slot1) {
    const auto t_start = std::chrono::steady_clock::now();
    if (G < 1 || G > 65535) return VILF_ERR_INVALID_ARGUMENT;
    const bool est_ex = h->opts.estimate_extrinsic != 0, est_td = h->opts.estimate_td != 0, ext = est_ex || est_td;
    HIPCHECK(h, hipSetDevice(h->device));
    if (!h->lw) h->lw = new LwCtx();
    LwCtx *c = h->lw;
    const bool prof = h->profiling != 0;
    if (prof && !c->ev[0]) { hipEventCreate(&c->ev[0]); hipEventCreate(&c->ev[1]); }
    std::vector<LwHostWin> hws(G);
    const bool trace = std::getenv("VILF_LW_TRACE") != nullptr;
    double t_ph[6] = {0, 0, 0, 0, 0, 0};              // VILF_LW_TRACE: host phases (validate + priors, pack, upload + enqueue, wait, outputs)
    auto t_last = t_start;
    auto lap = [&](int i) { if (trace) { const auto n = std::chrono::steady_clock::now(); t_ph[i] += std::chrono::duration<double, std::milli>(n - t_last).count(); t_last = n; } };
    // ---- validation, sizes
    bool any_slot = false, contig = slot1 != nullptr;       // contig: the group is a run of consecutive batch slots (vilf_batch_solve): priors in, states back in bulk copies
    size_t tot_imu = 0, tot_x = 0;
    for (int g = 0; g < G; g++) {
        LwHostWin &w = hws[g];
        const vilf_window_in *in = ins[g];
        w.in = in; w.out = outs[g];
        w.resident = slot1 && slot1[g] != 0; w.slot = w.resident ? (size_t)(slot1[g] - 1) : 0;
        any_slot = any_slot || w.resident;
        if (!w.resident || w.slot != hws[0].slot + (size_t)g) contig = false;
        const int NF = in->n_frames, F = in->n_features;
        if (NF < 2 || F < 0 || !in->para_pose || !in->para_speed_bias || !in->imu || (F && (!in->para_feature || !in->feature_const || !in->feature_start_frame || !in->feature_obs_offset || !in->obs_point)))
            return VILF_ERR_INVALID_ARGUMENT;
        if (in->n_obs < 0 || (F && (in->feature_obs_offset[0] != 0 || in->feature_obs_offset[F] != in->n_obs)) || (!F && in->n_obs != 0)) { h->err = "feature_obs_offset must start at 0 and end at n_obs"; return VILF_ERR_INVALID_ARGUMENT; }
        for (int f = 0; f < F; f++) if (in->feature_obs_offset[f + 1] - in->feature_obs_offset[f] < 2) { h->err = "feature with fewer than two observations"; return VILF_ERR_INVALID_ARGUMENT; }
        if (est_td && F && (!in->obs_velocity || !in->obs_cur_td || !in->obs_row)) { h->err = "estimate_td needs obs_velocity / obs_cur_td / obs_row"; return VILF_ERR_INVALID_ARGUMENT; }
        if (w.resident && (NF != VB_NF || !h->resident || (int)w.slot >= h->B)) return VILF_ERR_INVALID_ARGUMENT;
        for (int f = 0; f < F; f++) {
            const int o0 = in->feature_obs_offset[f], o1 = in->feature_obs_offset[f + 1], s = in->feature_start_frame[f];
            if (s < 0 || s + (o1 - o0) > NF) { h->err = "feature track leaves the window"; return VILF_ERR_INVALID_ARGUMENT; }
        }
        w.NF = NF; w.F = F;
        w.cEx = est_ex ? 15 * NF : -1; w.cTd = est_td ? 15 * NF + (est_ex ? 6 : 0) : -1;      // columns of Ex_Pose / td after the frame blocks
        w.P = 15 * NF + (est_ex ? 6 : 0) + (est_td ? 1 : 0); w.N = w.P + F;
        w.PC = 6 * NF + (est_ex ? 6 : 0) + (est_td ? 1 : 0); w.WS = (w.PC + 7) / 8 * 8; w.nchunk = (F + SY_KB - 1) / SY_KB;      // W over the pose / Ex / td columns only, 64-byte rows
        if (w.P > vilf_lw_chol_max_n()) { h->err = "window too large for the general path (reduced system above 12288 columns)"; return VILF_ERR_UNSUPPORTED; }
        w.xo = 16 * (size_t)NF + F;
        w.nvis = std::max(0, in->n_obs - F); w.nimu = NF - 1;
        w.npairs_cap = (int)std::min<long long>(w.nvis, (long long)NF * (NF - 1) / 2); w.nslots_cap = (w.nvis + LW_CH - 1) / LW_CH + w.npairs_cap;
        w.use_lidar = h->opts.use_lidar_const && in->lidar;
        tot_imu += w.nimu; tot_x += w.xo + 8;
    }
    {
        const int rca = lw_chol_back_attr(h);
        if (rca != VILF_OK) return rca;
    }
    // ---- the resident windows: their priors (marginalization_factor.cpp:333-381: block tables to the host, J0 / r0 / J0^T J0 stay on the device) and their CURRENT state
    // in the batch buffers — like the batched kernels, a solve continues from the resident state (the uploaded one after an upload or a rewind, the solved one otherwise)
    std::vector<double> r_pose, r_sb, r_feat, r_ex, r_td;
    const size_t rF = any_slot ? (size_t)h->batch.Fmax : 0;
    if (any_slot) {
        std::vector<int> hdr_all; std::vector<double> x0_all;
        r_pose.resize((size_t)G * 77); r_sb.resize((size_t)G * 99); r_feat.resize((size_t)G * std::max<size_t>(rF, 1)); r_ex.resize((size_t)G * 7); r_td.resize(G);
        if (contig) {
            const size_t s0 = hws[0].slot;
            hdr_all.resize((size_t)G * VB_PRIOR_HDR); x0_all.resize((size_t)G * 24 * 9);
            // seven copies: through the pinned image of the previous group when there is one (it is idle here) — a copy into pageable memory is staged by the runtime
            // and waited for, ~25 us each; from pinned memory the seven are enqueued in ~25 us together (round 5)
            struct Seg { void *dst; const void *src; size_t bytes; };
            const Seg segs[7] = {{hdr_all.data(), h->d[D_PHDR].as<int>() + s0 * VB_PRIOR_HDR, hdr_all.size() * 4}, {x0_all.data(), h->d[D_PX0].as<double>() + s0 * 24 * 9, x0_all.size() * 8},
                                 {r_pose.data(), h->d[D_POSE].as<double>() + s0 * 77, r_pose.size() * 8}, {r_sb.data(), h->d[D_SB].as<double>() + s0 * 99, r_sb.size() * 8},
                                 {r_feat.data(), h->d[D_FEAT].as<double>() + s0 * rF, (size_t)G * rF * 8}, {r_ex.data(), h->d[D_EX].as<double>() + s0 * 7, r_ex.size() * 8},
                                 {r_td.data(), h->d[D_TD].as<double>() + s0, r_td.size() * 8}};
            size_t tot = 0;
            for (const Seg &sg : segs) tot += (sg.bytes + 63) & ~(size_t)63;
            if (c->stage && c->stage_cap >= tot) {
                size_t at = 0;
                for (const Seg &sg : segs) { if (sg.bytes) HIPCHECK(h, hipMemcpyAsync(c->stage + at, sg.src, sg.bytes, hipMemcpyDeviceToHost, h->stream)); at += (sg.bytes + 63) & ~(size_t)63; }
                HIPCHECK(h, hipStreamSynchronize(h->stream));
                at = 0;
                for (const Seg &sg : segs) { if (sg.bytes) std::memcpy(sg.dst, c->stage + at, sg.bytes); at += (sg.bytes + 63) & ~(size_t)63; }
            } else
                for (const Seg &sg : segs) if (sg.bytes) HIPCHECK(h, hipMemcpyAsync(sg.dst, sg.src, sg.bytes, hipMemcpyDeviceToHost, h->stream));
        } else
            for (int g = 0; g < G; g++) {
                LwHostWin &w = hws[g];
                if (!w.resident) continue;
                HIPCHECK(h, hipMemcpyAsync(w.phdr, h->d[D_PHDR].as<int>() + w.slot * VB_PRIOR_HDR, sizeof(w.phdr), hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(h, hipMemcpyAsync(w.px0, h->d[D_PX0].as<double>() + w.slot * 24 * 9, sizeof(w.px0), hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(h, hipMemcpyAsync(&r_pose[(size_t)g * 77], h->d[D_POSE].as<double>() + w.slot * 77, 77 * 8, hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(h, hipMemcpyAsync(&r_sb[(size_t)g * 99], h->d[D_SB].as<double>() + w.slot * 99, 99 * 8, hipMemcpyDeviceToHost, h->stream));
                if (rF) HIPCHECK(h, hipMemcpyAsync(&r_feat[(size_t)g * rF], h->d[D_FEAT].as<double>() + w.slot * rF, rF * 8, hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(h, hipMemcpyAsync(&r_ex[(size_t)g * 7], h->d[D_EX].as<double>() + w.slot * 7, 56, hipMemcpyDeviceToHost, h->stream));
                HIPCHECK(h, hipMemcpyAsync(&r_td[g], h->d[D_TD].as<double>() + w.slot, 8, hipMemcpyDeviceToHost, h->stream));
            }
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        for (int g = 0; g < G; g++) {
            LwHostWin &w = hws[g];
            if (!w.resident) continue;
            if (contig) { std::memcpy(w.phdr, &hdr_all[(size_t)g * VB_PRIOR_HDR], sizeof(w.phdr)); std::memcpy(w.px0, &x0_all[(size_t)g * 24 * 9], sizeof(w.px0)); }
            if (w.phdr[0]) { w.pn = w.phdr[1]; w.pnb = w.phdr[2]; }
        }
    }
    lap(0);
    // ---- arena layout. Inputs first (one host image, one copy): per window vis | tdr | lid | fconst | scal | pcol; then group-wide runs imu | cov | x (k_imu_prep
    // takes the IMU factors of all windows in one launch; the states come back in one copy); then the minimizer scalars (one copy back) and the work areas.
    size_t off = 0;
    for (LwHostWin &w : hws) {
        w.o_vis = off; off = lw_al(off + std::max(w.nvis, 1) * sizeof(LwVis));
        w.o_tdr = off; off = lw_al(off + (est_td ? std::max(w.nvis, 1) : 1) * sizeof(LwTd));
        w.o_lid = off; off = lw_al(off + (size_t)w.nimu * 7 * 8);
        w.o_fconst = off; off = lw_al(off + std::max(w.F, 1));
        w.o_scal = off; off = lw_al(off + 16 * 8);
        w.o_pcol = off; off = lw_al(off + VB_PRIOR_LD * sizeof(int));
        w.o_kspan = off; off = lw_al(off + (size_t)std::max(w.nchunk, 1) * 2 * sizeof(int));
        w.o_fvis = off; off = lw_al(off + ((size_t)w.F + 1) * sizeof(int));
        w.o_fspan = off; off = lw_al(off + (size_t)std::max(w.F, 1) * 2 * sizeof(int));
        w.o_fidx = off; off = lw_al(off + (size_t)std::max(w.nvis, 1) * sizeof(int));
        w.o_cslot = off; off = lw_al(off + ((size_t)(w.nvis + LW_CH - 1) / LW_CH + 1) * sizeof(int));
        w.o_prt = off; off = lw_al(off + (size_t)std::max(w.npairs_cap, 1) * 4 * sizeof(int));
        w.o_froff = off; off = lw_al(off + ((size_t)w.NF + 1) * sizeof(int));
        w.o_frlist = off; off = lw_al(off + (size_t)std::max(w.nslots_cap, 1) * 2 * sizeof(int));
    }
    const size_t o_imu = off; off = lw_al(off + tot_imu * IMU_REC * 8);
    const size_t o_cov = off; off = lw_al(off + tot_imu * 225 * 8);
    const size_t o_x = off; off = lw_al(off + tot_x * 8);
    const size_t n_input = off;
    const size_t o_ctl = off; off = lw_al(off + (size_t)G * sizeof(LwCtl));
    struct WorkOff { size_t costP, cand, Hpp, W, hf, gp, gf, S, SC, PS, rhs, tmpP, tmpF, vec, den, rsd, info, nvec, pdx; };
    std::vector<WorkOff> wo(G);
    for (int g = 0; g < G; g++) {
        const LwHostWin &w = hws[g];
        const size_t sP = w.P, sF = std::max(w.F, 1), sN = w.N;
        WorkOff &o = wo[g];
        auto take = [&](size_t bytes) { const size_t at = off; off = lw_al(off + bytes); return at; };
        o.cand = take((w.xo + 8) * 8); o.Hpp = take(sP * sP * 8); o.W = take(sF * (size_t)w.WS * 8); o.hf = take(sF * 8); o.gp = take(sP * 8); o.gf = take(sF * 8);
        o.S = take((sP + 1) * sP * 8);
        { const size_t nt = (w.PC + 63) / 64, ntile = nt * (nt + 1) / 2; o.SC = take((size_t)SY_KS * (ntile * 4096 + nt * 64) * 8); }
        o.PS = take((size_t)std::max(w.nslots_cap, 1) * 160 * 8);
        o.costP = take(((size_t)(w.nvis + LW_CH - 1) / LW_CH + w.NF + 2) * 8);
        o.rhs = take(sP * 8); o.tmpP = take(sP * 8); o.tmpF = take(sF * 8); o.vec = take(2 * sN * 8);
        o.den = take(sF * 8); o.rsd = take(sF * 8); o.info = take(64); o.nvec = take(7 * sN * 8); o.pdx = take(VB_PRIOR_LD * 8);
    }
    if (!c->arena.ensure(off) || !c->desc.ensure((size_t)G * sizeof(LwWin))) { h->err = "hipMalloc failed (general-path solve)"; return VILF_ERR_DEVICE; }
    if (c->stage_cap < n_input) {
        if (c->stage) hipHostFree(c->stage);
        c->stage = nullptr; c->stage_cap = 0;
        if (hipHostMalloc(reinterpret_cast<void **>(&c->stage), n_input + n_input / 8, hipHostMallocDefault) != hipSuccess) { h->err = "hipHostMalloc failed (general-path solve)"; return VILF_ERR_DEVICE; }
        c->stage_cap = n_input + n_input / 8;
    }
    char *st = c->stage, *dev = c->arena.as<char>();
    // ---- pack the factors
    // extrinsic-derived constants (lidar_factor.h:28-29): q_il = RIC RCL, t_il = RIC TCL + TIC
    double Ril[9], qil[4], til[3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += h->opts.RIC[3 * i + k] * h->opts.RCL[3 * k + j]; Ril[3 * i + j] = s; }
    for (int i = 0; i < 3; i++) { double s = h->opts.TIC[i]; for (int k = 0; k < 3; k++) s += h->opts.RIC[3 * i + k] * h->opts.TCL[k]; til[i] = s; }
    {   // rotation matrix -> quaternion (Eigen's branch on the trace)
        const double *m = Ril; const double tr = m[0] + m[4] + m[8];
        double q[4];
        if (tr > 0) { double t = std::sqrt(tr + 1.0); q[3] = 0.5 * t; t = 0.5 / t; q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t; }
        else { int i = 0; if (m[4] > m[0]) i = 1; if (m[8] > m[4 * i]) i = 2; const int j = (i + 1) % 3, k = (j + 1) % 3; double t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0); q[i] = 0.5 * t; t = 0.5 / t; q[3] = (m[3 * k + j] - m[3 * j + k]) * t; q[j] = (m[3 * j + i] + m[3 * i + j]) * t; q[k] = (m[3 * k + i] + m[3 * i + k]) * t; }
        for (int k = 0; k < 4; k++) qil[k] = q[k];
    }
    const double sqrt_info = h->opts.focal_length / 1.5, cauchy_b = h->opts.cauchy_a * h->opts.cauchy_a;      // rho(s) = b log(1 + s / b), b = a^2 (ceres CauchyLoss; same as the batched path)
    std::vector<LwWin> dws(G);
    // K splits of the Schur reduce: a constant — the partials are summed in split order, so a count that depended on the group would make a window's bits depend on
    // who it is solved next to. 8 keeps a single window's reduce at 120 workgroups and costs a full group little (measured: 4 / 8 / 16 splits within 4 % at 32 windows)
    int group_nks = 8;
    if (const char *e = std::getenv("VILF_LW_NKS")) group_nks = std::min(SY_KS, std::max(1, std::atoi(e)));
    std::vector<size_t> imu_at(G), x_at(G);
    { size_t a = 0, b = 0; for (int g = 0; g < G; g++) { imu_at[g] = a; x_at[g] = b; a += hws[g].nimu; b += hws[g].xo + 8; } }
    auto pack = [&](int g) {
        std::vector<int> pair_cnt, vis_obs;
        const size_t at_imu = imu_at[g], at_x = x_at[g];
        LwHostWin &w = hws[g];
        const vilf_window_in *in = w.in;
        const int NF = w.NF, F = w.F, nvis = w.nvis, nimu = w.nimu;
        // the device's feature order: by (start frame, track length) (counting sort, stable) — the Schur reduce runs K over the features in chunks and skips a chunk
        // for the tiles its features' frames cannot meet: neighbours in this order have similar spans
        w.fdev.assign(std::max(F, 1), 0); w.fc.assign(std::max(F, 1), 0);
        {
            const size_t nkey = (size_t)NF * (NF + 1) + 1;
            std::vector<int> cnt(nkey + 1, 0);
            auto key = [&](int f) { return (size_t)in->feature_start_frame[f] * (NF + 1) + std::min(NF, std::max(0, in->feature_obs_offset[f + 1] - in->feature_obs_offset[f])); };
            for (int f = 0; f < F; f++) cnt[key(f) + 1]++;
            for (size_t k = 1; k <= nkey; k++) cnt[k] += cnt[k - 1];
            for (int f = 0; f < F; f++) { const int dvi = cnt[key(f)]++; w.fdev[f] = dvi; w.fc[dvi] = in->feature_const[f] ? 1 : 0; }
        }
        // pair-sorted (lw_visual flushes one block per run of equal pairs): counting sort over the NF^2 pair keys, stable in feature order
        LwVis *vis = reinterpret_cast<LwVis *>(st + w.o_vis);
        vis_obs.assign(2 * (size_t)std::max(nvis, 1), 0);           // (first, this) observation index of every factor: the td constants follow the pair sort
        pair_cnt.assign((size_t)NF * NF + 1, 0);
        for (int f = 0; f < F; f++) {
            const int o0 = in->feature_obs_offset[f], o1 = in->feature_obs_offset[f + 1], s = in->feature_start_frame[f];
            for (int t = o0 + 1; t < o1; t++) pair_cnt[(size_t)s * NF + s + (t - o0) + 1]++;
        }
        for (size_t k = 1; k < pair_cnt.size(); k++) pair_cnt[k] += pair_cnt[k - 1];
        int *fvis = reinterpret_cast<int *>(st + w.o_fvis), *fidx = reinterpret_cast<int *>(st + w.o_fidx);     // the feature-major view of the pair-sorted records
        fvis[0] = 0;
        for (int f = 0; f < F; f++) fvis[w.fdev[f] + 1] = std::max(0, in->feature_obs_offset[f + 1] - in->feature_obs_offset[f] - 1);
        for (int f = 0; f < F; f++) fvis[f + 1] += fvis[f];
        for (int f = 0; f < F; f++) {
            const int o0 = in->feature_obs_offset[f], o1 = in->feature_obs_offset[f + 1], s = in->feature_start_frame[f];
            for (int t = o0 + 1; t < o1; t++) {
                const int k = pair_cnt[(size_t)s * NF + s + (t - o0)]++;
                fidx[fvis[w.fdev[f]] + (t - o0 - 1)] = k;
                LwVis &v = vis[k];
                for (int q = 0; q < 3; q++) { v.pi[q] = in->obs_point[3 * (size_t)o0 + q]; v.pj[q] = in->obs_point[3 * (size_t)t + q]; }
                v.f = w.fdev[f]; v.i = s; v.j = s + (t - o0); v.cst = in->feature_const[f] ? 1 : 0;
                vis_obs[2 * (size_t)k] = o0; vis_obs[2 * (size_t)k + 1] = t;
            }
        }
        {   // slots of the pair-major kernel: a run of equal pairs inside a chunk of LW_CH factors owns one; pair table and the per-frame lists of lw_assemble
            int *cslot = reinterpret_cast<int *>(st + w.o_cslot), *prt = reinterpret_cast<int *>(st + w.o_prt), *froff = reinterpret_cast<int *>(st + w.o_froff), *frlist = reinterpret_cast<int *>(st + w.o_frlist);
            int nslot = 0, np = 0, last_key = -1;
            for (int k = 0; k < nvis; k++) {
                const int key = vis[k].i * NF + vis[k].j;
                if (k % LW_CH == 0) cslot[k / LW_CH] = nslot;
                if (key != last_key) { if (np) prt[4 * (np - 1) + 3] = nslot; prt[4 * np] = vis[k].i; prt[4 * np + 1] = vis[k].j; prt[4 * np + 2] = nslot; np++; last_key = key; nslot++; }
                else if (k % LW_CH == 0) nslot++;
            }
            if (np) prt[4 * (np - 1) + 3] = nslot;
            w.npairs = np;
            std::vector<int> cnt((size_t)NF + 1, 0);
            for (int q = 0; q < np; q++) { const int ns = prt[4 * q + 3] - prt[4 * q + 2]; cnt[prt[4 * q] + 1] += ns; cnt[prt[4 * q + 1] + 1] += ns; }
            for (int k = 1; k <= NF; k++) cnt[k] += cnt[k - 1];
            for (int k = 0; k <= NF; k++) froff[k] = cnt[k];
            for (int q = 0; q < np; q++)
                for (int sl = prt[4 * q + 2]; sl < prt[4 * q + 3]; sl++) { frlist[cnt[prt[4 * q]]++] = sl << 1; frlist[cnt[prt[4 * q + 1]]++] = (sl << 1) | 1; }
        }
        if (est_td) {                                     // projection_td_factor.cpp:6-21
            LwTd *tdrec = reinterpret_cast<LwTd *>(st + w.o_tdr);
            for (int k = 0; k < nvis; k++) {
                const int oi = vis_obs[2 * (size_t)k], oj = vis_obs[2 * (size_t)k + 1];
                LwTd &q = tdrec[k];
                q.vi[0] = in->obs_velocity[2 * (size_t)oi]; q.vi[1] = in->obs_velocity[2 * (size_t)oi + 1]; q.vj[0] = in->obs_velocity[2 * (size_t)oj]; q.vj[1] = in->obs_velocity[2 * (size_t)oj + 1];
                q.tdi = in->obs_cur_td[oi]; q.tdj = in->obs_cur_td[oj]; q.rowi_c = in->obs_row[oi] - h->opts.ROW / 2; q.rowj_c = in->obs_row[oj] - h->opts.ROW / 2;
            }
        }
        double *imu = reinterpret_cast<double *>(st + o_imu) + at_imu * IMU_REC, *cov = reinterpret_cast<double *>(st + o_cov) + at_imu * 225, *lid = reinterpret_cast<double *>(st + w.o_lid);
        std::memset(imu, 0, (size_t)nimu * IMU_REC * 8); std::memset(lid, 0, (size_t)nimu * 7 * 8);
        for (int k = 0; k < nimu; k++) {
            const vilf_imu_preint &p = in->imu[k + 1];
            double *rec = imu + (size_t)k * IMU_REC;
            rec[0] = p.sum_dt;
            for (int i = 0; i < 3; i++) { rec[1 + i] = p.delta_p[i]; rec[8 + i] = p.delta_v[i]; rec[11 + i] = p.linearized_ba[i]; rec[14 + i] = p.linearized_bg[i]; }
            for (int i = 0; i < 4; i++) rec[4 + i] = p.delta_q[i];
            auto blk = [&](int o2, int r0, int c0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rec[o2 + 3 * i + j] = p.jacobian[(r0 + i) * 15 + c0 + j]; };
            blk(17, 0, 9); blk(26, 0, 12); blk(35, 3, 12); blk(44, 6, 9); blk(53, 6, 12);
            rec[287] = (p.sum_dt > 10.0) ? 0.0 : 1.0;
            std::memcpy(cov + (size_t)k * 225, p.covariance, 225 * 8);
            if (in->lidar) { const vilf_lidar_constraint &l = in->lidar[k + 1]; for (int i = 0; i < 4; i++) lid[7 * k + i] = l.q[i]; for (int i = 0; i < 3; i++) lid[7 * k + 4 + i] = l.t[i]; }
            else lid[7 * k + 3] = 1.0;
        }
        if (F) std::memcpy(st + w.o_fconst, w.fc.data(), F);
        int bandf = 1;
        {   // per chunk of SY_KB features (device order) the compact columns their frame spans cover; with Ex_Pose / td as variables every row also reaches the last columns
            int *kspan = reinterpret_cast<int *>(st + w.o_kspan), *fspan = reinterpret_cast<int *>(st + w.o_fspan);
            for (int q = 0; q < w.nchunk; q++) { kspan[2 * q] = w.PC; kspan[2 * q + 1] = -1; }
            for (int f = 0; f < F; f++) {
                const int s0 = in->feature_start_frame[f], nobs = in->feature_obs_offset[f + 1] - in->feature_obs_offset[f], q = w.fdev[f] / SY_KB;
                fspan[2 * w.fdev[f]] = 6 * s0; fspan[2 * w.fdev[f] + 1] = (nobs < 2 || in->feature_const[f]) ? 6 * s0 - 1 : 6 * (s0 + nobs - 1) + 5;      // empty span: the row is zero
                if (nobs >= 2) bandf = std::max(bandf, nobs - 1);         // constant-depth features too: their factors still fill pose-pose blocks
                if (in->feature_const[f]) continue;                       // a constant depth: no row in W
                if (nobs < 2) continue;
                kspan[2 * q] = std::min(kspan[2 * q], 6 * s0);
                kspan[2 * q + 1] = std::max(kspan[2 * q + 1], w.PC > 6 * NF ? w.PC - 1 : 6 * (s0 + nobs - 1) + 5);      // lo > hi: nothing in this chunk
            }
        }
        double *scal = reinterpret_cast<double *>(st + w.o_scal);
        std::memset(scal, 0, 16 * 8);
        for (int k = 0; k < 4; k++) scal[1 + k] = qil[k];
        for (int k = 0; k < 3; k++) { scal[5 + k] = til[k]; scal[8 + k] = h->opts.G[k]; }
        // the state: x = pose | sb | feat | ex[7] | td
        w.x.resize(w.xo + 8);
        if (w.resident) {                              // NF = 11: the slot's current state
            std::memcpy(&w.x[0], &r_pose[(size_t)g * 77], 77 * 8); std::memcpy(&w.x[77], &r_sb[(size_t)g * 99], 99 * 8);
            for (int f = 0; f < F; f++) w.x[16 * (size_t)NF + w.fdev[f]] = r_feat[(size_t)g * rF + f];
            std::memcpy(&w.x[w.xo], &r_ex[(size_t)g * 7], 56); w.x[w.xo + 7] = r_td[g];
        } else {
            std::memcpy(&w.x[0], in->para_pose, 7 * (size_t)NF * 8); std::memcpy(&w.x[7 * (size_t)NF], in->para_speed_bias, 9 * (size_t)NF * 8);
            for (int f = 0; f < F; f++) w.x[16 * (size_t)NF + w.fdev[f]] = in->para_feature[f];
            std::memcpy(&w.x[w.xo], in->para_ex_pose, 56); w.x[w.xo + 7] = in->para_td;
        }
        std::memcpy(reinterpret_cast<double *>(st + o_x) + at_x, w.x.data(), w.x.size() * 8);
        int *pcol = reinterpret_cast<int *>(st + w.o_pcol);
        for (int i = 0; i < VB_PRIOR_LD; i++) pcol[i] = -1;
        for (int bk = 0; bk < w.pnb; bk++) {
            const int id = w.phdr[3 + bk], idx = w.phdr[51 + bk];
            if (id < NF) for (int k = 0; k < 6; k++) pcol[idx + k] = 15 * id + k;
            else if (id < 2 * NF) for (int k = 0; k < 9; k++) pcol[idx + k] = 15 * (id - NF) + 6 + k;
            else if (id == 2 * NF && est_ex) for (int k = 0; k < 6; k++) pcol[idx + k] = w.cEx + k;
            else if (id == 2 * NF + 1 && est_td) pcol[idx] = w.cTd;                       // para_Td, kept by the marginalization (estimator.cpp:930-935,968-969)
        }
        // ---- the descriptor
        LwWin &d = dws[g];
        std::memset(&d, 0, sizeof(d));
        const WorkOff &o = wo[g];
        const size_t sN = w.N;
        d.NF = NF; d.F = F; d.P = w.P; d.N = w.N; d.nvis = nvis; d.nimu = nimu; d.cEx = w.cEx; d.cTd = w.cTd; d.xo = (int)w.xo; d.est_ex = est_ex ? 1 : 0; d.est_td = est_td ? 1 : 0;
        d.use_lidar = w.use_lidar ? 1 : 0; d.pn = w.pn; d.pnb = w.pnb; d.max_it = h->opts.max_num_iterations;
        d.PC = w.PC; d.WS = w.WS; d.nchunk = w.nchunk; d.nks = group_nks;
        d.kspan = reinterpret_cast<const int *>(dev + w.o_kspan); d.fspan = reinterpret_cast<const int *>(dev + w.o_fspan);
        d.bandf = (w.pn || est_ex || est_td) ? NF : std::min(NF, bandf);
        d.fvis = reinterpret_cast<const int *>(dev + w.o_fvis); d.fidx = reinterpret_cast<const int *>(dev + w.o_fidx);
        d.PS = reinterpret_cast<double *>(dev + o.PS); d.costP = reinterpret_cast<double *>(dev + o.costP); d.ncostv = (nvis + LW_CH - 1) / LW_CH; d.cslot = reinterpret_cast<const int *>(dev + w.o_cslot); d.prt = reinterpret_cast<const int *>(dev + w.o_prt);
        d.froff = reinterpret_cast<const int *>(dev + w.o_froff); d.frlist = reinterpret_cast<const int *>(dev + w.o_frlist); d.npairs = w.npairs;
        d.sqrt_info = sqrt_info; d.cauchy_b = cauchy_b; d.tr_over_row = h->opts.TR / h->opts.ROW;
        d.ctl = reinterpret_cast<LwCtl *>(dev + o_ctl) + g;
        d.x = reinterpret_cast<double *>(dev + o_x) + at_x; d.cand = reinterpret_cast<double *>(dev + o.cand);
        d.vis = reinterpret_cast<const LwVis *>(dev + w.o_vis); d.tdr = reinterpret_cast<const LwTd *>(dev + w.o_tdr);
        d.imu = reinterpret_cast<const double *>(dev + o_imu) + at_imu * IMU_REC; d.lid = reinterpret_cast<const double *>(dev + w.o_lid);
        d.fconst = reinterpret_cast<const unsigned char *>(dev + w.o_fconst);
        auto dp = [&](size_t at) { return reinterpret_cast<double *>(dev + at); };
        d.Hpp = dp(o.Hpp); d.W = dp(o.W); d.hf = dp(o.hf); d.gp = dp(o.gp); d.gf = dp(o.gf); d.S = dp(o.S); d.SC = dp(o.SC); d.rhs = dp(o.rhs); d.tmpP = dp(o.tmpP); d.tmpF = dp(o.tmpF);
        d.vec = dp(o.vec); d.yf = dp(o.vec) + sN; d.scal = dp(w.o_scal); d.den = dp(o.den); d.rsd = dp(o.rsd); d.info = reinterpret_cast<int *>(dev + o.info);
        double *nv = dp(o.nvec);
        d.g = nv; d.diagH = nv + sN; d.scale = nv + 2 * sN; d.diagonal = nv + 3 * sN; d.gradient = nv + 4 * sN; d.gn = nv + 5 * sN; d.step = nv + 6 * sN;
        d.pcol = reinterpret_cast<const int *>(dev + w.o_pcol); d.pdx = dp(o.pdx);
        if (w.resident) {
            d.pJ = h->d[D_PJ].as<double>() + w.slot * VB_PRIOR_LD * VB_PRIOR_LD; d.pr0 = h->d[D_PR].as<double>() + w.slot * VB_PRIOR_LD; d.pH0 = h->d[D_PH].as<double>() + w.slot * VB_PRIOR_LD * VB_PRIOR_LD;
            d.phdr = h->d[D_PHDR].as<int>() + w.slot * VB_PRIOR_HDR; d.px0 = h->d[D_PX0].as<double>() + w.slot * 24 * 9;
        }
        w.dw = d;
    };
    {   // a group's windows are packed by a few host threads (a stress window is ~3 MB of factor records)
        const int nthr = (int)std::min<size_t>({(size_t)G, (size_t)12, (size_t)std::max(1u, std::thread::hardware_concurrency())});
        size_t tot_vis = 0;
        for (const LwHostWin &w : hws) tot_vis += w.nvis;
        if (nthr <= 1 || tot_vis < 20000) for (int g = 0; g < G; g++) pack(g);
        else {
            std::atomic<int> next(0);
            std::vector<std::thread> th;
            for (int t = 0; t < nthr; t++) th.emplace_back([&]() { for (int g = next.fetch_add(1); g < G; g = next.fetch_add(1)) pack(g); });
            for (std::thread &t : th) t.join();
        }
    }
    lap(1);
    HIPCHECK(h, hipMemcpyAsync(dev, st, n_input, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(c->desc.p, dws.data(), (size_t)G * sizeof(LwWin), hipMemcpyHostToDevice, h->stream));
    const LwWin *dws_dev = c->desc.as<LwWin>();
    hipLaunchKernelGGL(k_imu_prep, dim3((unsigned)((tot_imu + 3) / 4)), dim3(64), 0, h->stream, (int)tot_imu, reinterpret_cast<const double *>(dev + o_cov), (double *)nullptr, reinterpret_cast<double *>(dev + o_imu));

    int rc = VILF_OK;
    const int max_it = h->opts.max_num_iterations;
    // ---- the loop on the device: every iteration's launches are enqueued at once, nothing is read back until the end. Not with a wall-clock limit
    // (Ceres tests the clock at the top of every iteration: the host loop does) and not for a window whose factorisation fails (it sets `fallback`).
    bool ran_device = false;
    const bool tlim_on = h->opts.max_solver_time > 0;
    if (!tlim_on && !std::getenv("VILF_LW_HOST_LOOP")) {
        LwEnq q{h, c, dws_dev, LwDims(), ext, prof};
        for (const LwHostWin &w : hws) q.d.take(w);
        const dim3 one(1, 1, (unsigned)G);
        hipLaunchKernelGGL(lw_tr_init, one, dim3(TR_T), 0, h->stream, dws_dev);
        auto enq_linearize = [&](int first) {                              // eval_grad_jac of the host loop
            q.evaluate(0, 1, LW_SK_JAC);
            hipLaunchKernelGGL(lw_tr_post, one, dim3(TR_T), 0, h->stream, dws_dev, first);
            q.scale(0, LW_SK_JAC);
        };
        enq_linearize(1);
        for (int it = 0; it < max_it; it++) {
            hipLaunchKernelGGL(lw_tr_begin, one, dim3(TR_T), 0, h->stream, dws_dev);
            q.quad(LW_SK_SOLVE);
            hipLaunchKernelGGL(lw_tr_alpha, one, dim3(TR_T), 0, h->stream, dws_dev);
            q.linear_solve(LW_SK_SOLVE);
            hipLaunchKernelGGL(lw_tr_step, one, dim3(TR_T), 0, h->stream, dws_dev);
            q.quad(LW_SK_QUAD);
            hipLaunchKernelGGL(lw_tr_model, one, dim3(TR_T), 0, h->stream, dws_dev);
            q.evaluate(1, 0, LW_SK_EVAL);
            hipLaunchKernelGGL(lw_tr_decide, one, dim3(TR_T), 0, h->stream, dws_dev);
            if (it + 1 < max_it) enq_linearize(0);
        }
        HIPCHECK(h, hipGetLastError());
        std::vector<LwCtl> hcs(G);
        std::vector<double> xs(tot_x);
        lap(2);
        HIPCHECK(h, hipMemcpyAsync(hcs.data(), dev + o_ctl, (size_t)G * sizeof(LwCtl), hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipMemcpyAsync(xs.data(), dev + o_x, tot_x * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        lap(3);
        ran_device = true;
        size_t at = 0;
        for (int g = 0; g < G; g++) {
            LwHostWin &w = hws[g];
            w.hc = hcs[g];
            if (std::getenv("VILF_LW_FORCE_FALLBACK")) w.hc.fallback = 1;      // test hook: take the path of a failed factorisation (the host loop redoes the solve from the initial state)
            if (std::getenv("VILF_LW_TRACE")) std::fprintf(stderr, "[vilf lw] device loop, window %d: fallback %d iterations %d successful %d solves %d termination %d cost %.9g -> %.9g\n", g, w.hc.fallback, w.hc.iteration, w.hc.num_successful, w.hc.num_linear_solves, w.hc.termination, w.hc.initial_cost, w.hc.x_cost);
            if (!w.hc.fallback) std::memcpy(w.x.data(), &xs[at], (w.xo + 8) * 8);
            at += w.xo + 8;
        }
    }
    const double tl0 = h->opts.max_solver_time;
    for (int g = 0; g < G; g++) {
        LwHostWin &w = hws[g];
        if (ran_device && !w.hc.fallback) continue;
        const double tlim = tl0 > 0 ? tl0 * (w.in->marginalization_flag == VILF_MARGIN_OLD ? 4.0 / 5.0 : 1.0) : -1.0;   // estimator.cpp:847-850
        // a group under a wall-clock limit: every window's clock starts with its own solve (the reference runs one window per call)
        if ((rc = lw_host_loop(h, c, w, dws_dev + g, tlim, G == 1 ? t_start : std::chrono::steady_clock::now(), ext)) != VILF_OK) return rc;
    }
    // ---- outputs + double2vector (estimator.cpp:549-638)
    std::vector<double> b_pose, b_sb, b_feat, b_ex, b_td, b_ps, b_rs, b_vs, b_bas, b_bgs;
    std::vector<VbState> b_st;
    const size_t sF2 = any_slot ? (size_t)h->batch.Fmax : 0;
    if (contig) { b_pose.resize((size_t)G * 77); b_sb.resize((size_t)G * 99); b_feat.assign((size_t)G * sF2, 0.0); b_ex.resize((size_t)G * 7); b_td.resize(G); b_ps.resize((size_t)G * 33); b_rs.resize((size_t)G * 99); b_vs.resize((size_t)G * 33); b_bas.resize((size_t)G * 33); b_bgs.resize((size_t)G * 33); }
    bool abnormal = false;
    for (int g = 0; g < G; g++) {
        LwHostWin &w = hws[g];
        const vilf_window_in *in = w.in; vilf_window_out *out = w.out;
        const int NF = w.NF, F = w.F; const size_t xo = w.xo;
        {   // the inverse depths back into the caller's feature order
            std::vector<double> tmp(w.x.begin() + 16 * (size_t)NF, w.x.begin() + 16 * (size_t)NF + F);
            for (int f = 0; f < F; f++) w.x[16 * (size_t)NF + f] = tmp[w.fdev[f]];
        }
        const std::vector<double> &x = w.x;
        double R0b[9], P0b[3];
        if (in->gauge_R0) std::memcpy(R0b, in->gauge_R0, 72); else { double q0[4] = {in->para_pose[3], in->para_pose[4], in->para_pose[5], in->para_pose[6]}; h_q2R(q0, R0b); }
        if (in->gauge_P0) std::memcpy(P0b, in->gauge_P0, 24); else std::memcpy(P0b, in->para_pose, 24);
        if (out->para_pose) std::memcpy(out->para_pose, &x[0], 7 * (size_t)NF * 8);
        if (out->para_speed_bias) std::memcpy(out->para_speed_bias, &x[7 * (size_t)NF], 9 * (size_t)NF * 8);
        if (out->para_feature) for (int f = 0; f < F; f++) out->para_feature[f] = x[16 * (size_t)NF + f];
        double R00[9], y0[3], y00[3], rot_diff[9];
        h_q2R(&x[3], R00);
        h_R2ypr(R0b, y0); h_R2ypr(R00, y00);
        const double yd[3] = {y0[0] - y00[0], 0, 0};
        h_ypr2R(yd, rot_diff);
        if (std::fabs(std::fabs(y0[1]) - 90) < 1.0 || std::fabs(std::fabs(y00[1]) - 90) < 1.0)
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += R0b[3 * i + k] * R00[3 * j + k]; rot_diff[3 * i + j] = s; }
        for (int i = 0; i < NF; i++) {
            double Ri[9];
            h_q2R(&x[7 * i + 3], Ri);
            const double dp[3] = {x[7 * i] - x[0], x[7 * i + 1] - x[1], x[7 * i + 2] - x[2]};
            for (int a = 0; a < 3; a++) {
                for (int b = 0; b < 3; b++) { double s = 0; for (int k = 0; k < 3; k++) s += rot_diff[3 * a + k] * Ri[3 * k + b]; out->Rs[9 * i + 3 * a + b] = s; }
                double sp = P0b[a], sv = 0;
                for (int k = 0; k < 3; k++) { sp += rot_diff[3 * a + k] * dp[k]; sv += rot_diff[3 * a + k] * x[7 * NF + 9 * i + k]; }
                out->Ps[3 * i + a] = sp; out->Vs[3 * i + a] = sv;
                out->Bas[3 * i + a] = x[7 * NF + 9 * i + 3 + a]; out->Bgs[3 * i + a] = x[7 * NF + 9 * i + 6 + a];
            }
        }
        for (int k = 0; k < 3; k++) out->tic[k] = x[xo + k];                 // double2vector :607-617: tic / ric / td from para_Ex_Pose / para_Td
        h_q2R(&x[xo + 3], out->ric);
        out->td = est_td ? x[xo + 7] : in->para_td;
        const LwCtl &hc = w.hc;
        out->summary.num_iterations = hc.iteration; out->summary.num_successful_steps = hc.num_successful; out->summary.num_linear_solves = hc.num_linear_solves;
        out->summary.termination = hc.termination; out->summary.initial_cost = hc.initial_cost; out->summary.final_cost = hc.x_cost; out->summary.final_radius = hc.radius;
        if (hc.termination == VILF_TERM_FAILURE) abnormal = true;
        if (w.resident) {                                                    // the marginalization of this window reads the batch buffers of its slot
            const size_t s = w.slot;
            VbState stt;
            std::memset(&stt, 0, sizeof(stt));
            stt.iteration = hc.iteration; stt.num_successful = hc.num_successful; stt.num_linear_solves = hc.num_linear_solves;
            stt.termination = hc.termination; stt.initial_cost = hc.initial_cost; stt.x_cost = hc.x_cost; stt.radius = hc.radius; stt.done = 1;
            if (contig) {
                std::memcpy(&b_pose[(size_t)g * 77], &x[0], 77 * 8); std::memcpy(&b_sb[(size_t)g * 99], &x[77], 99 * 8);
                for (int f = 0; f < F; f++) b_feat[(size_t)g * sF2 + f] = x[16 * (size_t)NF + f];
                std::memcpy(&b_ex[(size_t)g * 7], &x[xo], 56); b_td[g] = out->td;
                std::memcpy(&b_ps[(size_t)g * 33], out->Ps, 33 * 8); std::memcpy(&b_rs[(size_t)g * 99], out->Rs, 99 * 8); std::memcpy(&b_vs[(size_t)g * 33], out->Vs, 33 * 8);
                std::memcpy(&b_bas[(size_t)g * 33], out->Bas, 33 * 8); std::memcpy(&b_bgs[(size_t)g * 33], out->Bgs, 33 * 8);
                b_st.push_back(stt);
            } else {
                HIPCHECK(h, hipMemcpyAsync(h->d[D_POSE].as<double>() + s * 77, &x[0], 77 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_SB].as<double>() + s * 99, &x[77], 99 * 8, hipMemcpyHostToDevice, h->stream));
                if (F) HIPCHECK(h, hipMemcpyAsync(h->d[D_FEAT].as<double>() + s * sF2, &x[16 * (size_t)NF], (size_t)F * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_EX].as<double>() + s * 7, &x[xo], 56, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_TD].as<double>() + s, &out->td, 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_OPS].as<double>() + s * 33, out->Ps, 33 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_ORS].as<double>() + s * 99, out->Rs, 99 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_OVS].as<double>() + s * 33, out->Vs, 33 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_OBAS].as<double>() + s * 33, out->Bas, 33 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_OBGS].as<double>() + s * 33, out->Bgs, 33 * 8, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->batch.st + s, &stt, sizeof(stt), hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipStreamSynchronize(h->stream));              // the sources are locals of this iteration
            }
            std::memcpy(&h->h_ex[s * 7], &x[xo], 56); h->h_td[s] = out->td;
        }
    }
    if (contig) {
        const size_t s0 = hws[0].slot;
        struct Seg { void *dst; const void *src; size_t bytes; };
        const Seg segs[11] = {{h->d[D_POSE].as<double>() + s0 * 77, b_pose.data(), b_pose.size() * 8}, {h->d[D_SB].as<double>() + s0 * 99, b_sb.data(), b_sb.size() * 8},
                              {h->d[D_FEAT].as<double>() + s0 * sF2, b_feat.data(), sF2 ? b_feat.size() * 8 : 0}, {h->d[D_EX].as<double>() + s0 * 7, b_ex.data(), b_ex.size() * 8},
                              {h->d[D_TD].as<double>() + s0, b_td.data(), b_td.size() * 8}, {h->d[D_OPS].as<double>() + s0 * 33, b_ps.data(), b_ps.size() * 8},
                              {h->d[D_ORS].as<double>() + s0 * 99, b_rs.data(), b_rs.size() * 8}, {h->d[D_OVS].as<double>() + s0 * 33, b_vs.data(), b_vs.size() * 8},
                              {h->d[D_OBAS].as<double>() + s0 * 33, b_bas.data(), b_bas.size() * 8}, {h->d[D_OBGS].as<double>() + s0 * 33, b_bgs.data(), b_bgs.size() * 8},
                              {h->batch.st + s0, b_st.data(), b_st.size() * sizeof(VbState)}};
        size_t tot = 0;
        for (const Seg &sg : segs) tot += (sg.bytes + 63) & ~(size_t)63;
        const bool pinned = c->stage && c->stage_cap >= tot;          // the group's input image is idle by now: eleven copies from pinned memory are enqueued in the time of one from pageable memory
        size_t at = 0;
        for (const Seg &sg : segs) {
            if (!sg.bytes) continue;
            const void *src = sg.src;
            if (pinned) { std::memcpy(c->stage + at, sg.src, sg.bytes); src = c->stage + at; at += (sg.bytes + 63) & ~(size_t)63; }
            HIPCHECK(h, hipMemcpyAsync(sg.dst, src, sg.bytes, hipMemcpyHostToDevice, h->stream));
        }
        HIPCHECK(h, hipStreamSynchronize(h->stream));
    }
    lap(4);
    if (trace) std::fprintf(stderr, "[vilf lw] group of %d: validate + priors %.3f ms, pack %.3f, upload + enqueue %.3f, wait %.3f, outputs %.3f\n", G, t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4]);
    const double usec = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count();
    for (LwHostWin &w : hws) w.out->summary.usec_solve = usec;          // the group's wall time: its windows are solved side by side
    return abnormal ? VILF_SOLVER_ABNORMAL : VILF_OK;
}

// batch_slot1 != 0: the window is also resident as slot batch_slot1 - 1 of the 11-frame batch
int vilf_lw_window_solve(vilf_handle *h, const vilf_window_in *in, vilf_window_out *out, int batch_slot1) {
    return vilf_lw_group_solve(h, 1, &in, &out, &batch_slot1);
}

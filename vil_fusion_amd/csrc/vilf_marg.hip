// vilf_marg.hip — device marginalization ≙ estimator.cpp:863-1046 + MarginalizationInfo::{preMarginalize, marginalize,
// getParameterBlocks} (factor/marginalization_factor.cpp:110-319). One 256-thread workgroup per window.
//
//   k_marg_prepare  vector2double() of the post-gauge state; evaluate the factors touching the dropped blocks (prior,
//                   lidarFactor[1], IMUFactor[1], every ProjectionFactor whose feature starts in frame 0) with the Cauchy
//                   corrector; dense-variable normal equations Hd/gd + per-feature arrow rows (the features are 1-dim blocks
//                   coupled only to Pose[0], Pose[j], Ex_Pose and themselves)
//   k_marg_schur    Amm = 0.5 (Amm + Amm^T); eigen-decomposition by a parallel cyclic Jacobi resident in LDS; pseudo-inverse
//                   with the reference's eps = 1e-8 (:267-272); A = Arr - Arm Amm^+ Amr, b = brr - Arm Amm^+ bmm (:275-281),
//                   computed as X'^T f(L) X' with X' = V^T [Amr | bmm] obtained by replaying the rotations (no eigenvector
//                   matrix is formed)
//   k_marg_finish   second eigen-decomposition (with eigenvectors), J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b (:283-291); block table
//                   with the address shift (estimator.cpp:960-971 / :1016-1037); the new prior stays on the device
// Block ordering: dropped blocks then kept blocks, both ascending in block id (the reference iterates address-keyed
// unordered_maps; any order is a permutation of the same result — SURVEY.md §7 "address-keyed bookkeeping").
#include "vilf_device.hpp"
#include "vilf_batch.hpp"

using namespace vd;
#define NT VB_NT
#define VILF_MAX_FEATURES_DEV 1000
__device__ __forceinline__ int pair_index_c(int i, int j) { return j * (j - 1) / 2 + i; }  // i < j

__device__ __forceinline__ void rr_pair(int round, int k, int M, int &p, int &q) {   // round-robin tournament pairing
    int a = (k == 0) ? (M - 1) : (round + k) % (M - 1);
    int c = (round - k + (M - 1)) % (M - 1);
    p = min(a, c); q = max(a, c);
}

// Parallel two-sided cyclic Jacobi on a symmetric M x M matrix (M even, row stride ld). WITH_V: accumulate V <- V J.
// rotlog (optional): (c, s) of every pair of every round, [sweep][round][M/2][2]. Returns the number of sweeps run.
// In-kernel phase stamps: diagnostic build only (make DEFS=-DVILF_STAMPS); workgroup MG_STAMP_WG writes them.
#ifdef VILF_STAMPS
__device__ long long mg_dbg[4 * 32];
#define MG_STAMP_WG 1500
#define MG_STAMP(kid, i) do { if (blockIdx.x == MG_STAMP_WG && threadIdx.x == 0) mg_dbg[(kid) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int vilf_debug_stamps_marg(long long *out128) { return hipMemcpyFromSymbol(out128, HIP_SYMBOL(mg_dbg), sizeof(long long) * 4 * 32) == hipSuccess ? 0 : -1; }
#define MG_ACC_DECL long long macc[6] = {0, 0, 0, 0, 0, 0}; long long macc_last = __builtin_readcyclecounter();
#define MG_ACC(i) do { const long long now_ = __builtin_readcyclecounter(); macc[i] += now_ - macc_last; macc_last = now_; } while (0)
#define MG_ACC_OUT(kid) do { if (blockIdx.x == MG_STAMP_WG && threadIdx.x == 0) for (int i_ = 0; i_ < 6; i_++) mg_dbg[(kid) * 32 + 24 + i_] = macc[i_]; } while (0)
#else
#define MG_STAMP(kid, i) do { } while (0)
#define MG_ACC_DECL
#define MG_ACC(i) do { } while (0)
#define MG_ACC_OUT(kid) do { } while (0)
#endif

template <bool WITH_V>
__device__ int jacobi_eig(double *A, int M, int ld, double *V, double *rotlog, double *s_cs, int *s_flag) {
    const int tid = threadIdx.x, H = M / 2;
    int sweep = 0;
    for (; sweep < MG_SWEEPS; sweep++) {
        if (tid == 0) *s_flag = 0;
        __syncthreads();
        for (int round = 0; round < M - 1; round++) {
            for (int k = tid; k < H; k += NT) {
                int p, q; rr_pair(round, k, M, p, q);
                const double app = A[p * ld + p], aqq = A[q * ld + q], apq = A[p * ld + q];
                double c = 1.0, s = 0.0;
                const double g = 100.0 * fabs(apq);
                if (apq != 0.0) {
                    if (sweep > 3 && fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) { A[p * ld + q] = 0.0; A[q * ld + p] = 0.0; }
                    else {
                        const double theta = (aqq - app) / (2.0 * apq);
                        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(t * t + 1.0); s = t * c;
                        *s_flag = 1;
                    }
                }
                s_cs[2 * k] = c; s_cs[2 * k + 1] = s;
                if (rotlog) { double *l = rotlog + ((size_t)(sweep * (M - 1) + round) * H + k) * 2; l[0] = c; l[1] = s; }
            }
            __syncthreads();
            for (int it = tid; it < H * M; it += NT) {            // rows p, q  <-  J^T A
                const int k = it / M, j = it - k * M;
                int p, q; rr_pair(round, k, M, p, q);
                const double c = s_cs[2 * k], s = s_cs[2 * k + 1];
                const double a = A[p * ld + j], bq = A[q * ld + j];
                A[p * ld + j] = c * a - s * bq; A[q * ld + j] = s * a + c * bq;
            }
            __syncthreads();
            for (int it = tid; it < H * M; it += NT) {            // columns p, q  <-  (.) J
                const int i = it / H, k = it - i * H;
                int p, q; rr_pair(round, k, M, p, q);
                const double c = s_cs[2 * k], s = s_cs[2 * k + 1];
                const double a = A[i * ld + p], bq = A[i * ld + q];
                A[i * ld + p] = c * a - s * bq; A[i * ld + q] = s * a + c * bq;
                if (WITH_V) { const double va = V[i * ld + p], vb = V[i * ld + q]; V[i * ld + p] = c * va - s * vb; V[i * ld + q] = s * va + c * vb; }
            }
            __syncthreads();
        }
        if (!*s_flag) { sweep++; break; }
        __syncthreads();
    }
    return sweep;
}

// ------------------------------------------------------------------------------------------------------------------
typedef double mgp_double4 __attribute__((ext_vector_type(4)));
// TD = estimate_td: the ProjectionTdFactor evaluation costs 100+ VGPRs more than ProjectionFactor; compiled apart so that the KITTI case keeps two workgroups per CU.
template <bool TD>
__device__ __forceinline__ void marg_prepare_body(const VbBatch &b, const VbMarg &g) {
    const int w = blockIdx.x, tid = threadIdx.x;
    MG_STAMP(0, 0);
    __shared__ double s_pose[77], s_sb[99], s_R[99], s_ex[7], s_ric[9], s_dx[VB_PRIOR_LD], s_J[15 * 32], s_r[16], s_lJ[72], s_lr[8];
    __shared__ double s_pm[10 * MG_PAIRM];
    __shared__ int s_off_pose[VB_NF], s_off_sb[2], s_off_ex, s_off_td, s_pmap[VB_PRIOR_LD], s_hdr[8], s_pst[VB_NPAIR], s_pcn[VB_NPAIR], s_pcl[VB_NPAIR];     // pair table: start inside the class list, factor count, class
    __shared__ double s_td;
    __shared__ double s_zero;                                    // a zero in LDS: what a masked lane of the pair products reads
    __shared__ double s_part[4 * MG_PTRI];                       // the four waves' partial pair products (packed lower triangle of 20 x 20)
    __shared__ __attribute__((aligned(16))) double s_rows[MG_GCH * MG_MROW];                  // factor rows of the pair products; afterwards the rank -> feature table of the arrow rows
    __shared__ int s_slots[MG_SLOTS], s_pend[10];                 // Mbuf row of the t-th start-frame-0 factor (evaluation order = pair order); cumulative factor count per pair
    int *info = g.info + (size_t)w * MG_INFO;
    const int F = b.n_feat[w];
    const size_t FM = b.Fmax, FC = b.FACmax;
    const int *phdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    const int mode = g.mflag[w];
    const int have_prior = phdr[0];
    const int *f_start = b.f_start + (size_t)w * FM, *f_nobs = b.f_nobs + (size_t)w * FM, *f_fac0 = b.f_fac0 + (size_t)w * FM;
    int *f0rank = g.f0rank + (size_t)w * FM;
    double *st_feat = g.st_feat + (size_t)w * FM;

    // ---- vector2double() of the post-gauge state (estimator.cpp:866; feature_manager.cpp:150-168,194-216) --------------
    if (tid < VB_NF) {
        const double *P = b.out_Ps + ((size_t)w * VB_NF + tid) * 3, *R = b.out_Rs + ((size_t)w * VB_NF + tid) * 9;
        for (int k = 0; k < 3; k++) s_pose[7 * tid + k] = P[k];
        q_store(s_pose + 7 * tid + 3, q_fromR(R));
        for (int k = 0; k < 9; k++) s_R[9 * tid + k] = R[k];
        for (int k = 0; k < 3; k++) { s_sb[9 * tid + k] = b.out_Vs[((size_t)w * VB_NF + tid) * 3 + k]; s_sb[9 * tid + 3 + k] = b.out_Bas[((size_t)w * VB_NF + tid) * 3 + k]; s_sb[9 * tid + 6 + k] = b.out_Bgs[((size_t)w * VB_NF + tid) * 3 + k]; }
    }
    if (tid == 33) s_td = b.td[w];
    if (tid == 32) {
        const double *ex = b.ex + (size_t)w * 7;
        q_toR(q_load(ex + 3), s_ric);
        for (int k = 0; k < 3; k++) s_ex[k] = ex[k];
        q_store(s_ex + 3, q_fromR(s_ric));
    }
    for (int f = tid; f < F; f += NT) {
        const double est = 1.0 / b.feat[(size_t)w * FM + f];
        const double v = est > 0 ? 1.0 / est : 1.0 / g.init_depth;
        st_feat[f] = v;
        if (f < MG_GCH * MG_MROW) s_rows[f] = v;       // (a copy in the pair products' staging area, idle until then: the factor evaluation below reads it behind the factor's record — from global memory that was a second dependent trip per factor)
    }
    if (tid < VB_NPAIR) { const int *pt = b.pair_off + (size_t)w * VB_PTAB; const int v1 = pt[2 * tid + 1]; s_pst[tid] = pt[2 * tid]; s_pcn[tid] = v1 & 0xffffff; s_pcl[tid] = v1 >> 24; }
    __syncthreads();
    MG_STAMP(0, 1);
    if (tid < 77) g.st_pose[(size_t)w * 77 + tid] = s_pose[tid];
    if (tid < 99) g.st_sb[(size_t)w * 99 + tid] = s_sb[tid];
    if (tid < 7) g.st_ex[(size_t)w * 7 + tid] = s_ex[tid];

    // ---- start-frame-0 features: rank (order of appearance) and longest track, all threads ----------------------------------
    __shared__ int s_fw[NT / 64], s_mf, s_maxobs;
    if (tid == 0) { s_mf = 0; s_maxobs = 0; s_zero = 0.0; }
    __syncthreads();
    MG_STAMP(0, 2);
    if (mode == 0) {
        for (int f0 = 0; f0 < F; f0 += NT) {
            const int f = f0 + tid;
            const int flag = (f < F && f_start[f] == 0) ? 1 : 0;
            const int nob = flag ? f_nobs[f] : 0;
            int incl = flag, mx = nob;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += u; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
            if ((tid & 63) == 63) s_fw[tid >> 6] = incl;
            if ((tid & 63) == 0 && mx > 0) atomicMax(&s_maxobs, mx);
            __syncthreads();
    MG_STAMP(0, 3);
            int off = s_mf;
            for (int k = 0; k < (tid >> 6); k++) off += s_fw[k];
            if (f < F) f0rank[f] = flag ? off + incl - 1 : -1;
            __syncthreads();
    MG_STAMP(0, 4);
            if (tid == 0) { int t = 0; for (int k = 0; k < NT / 64; k++) t += s_fw[k]; s_mf += t; }
            __syncthreads();
    MG_STAMP(0, 5);
        }
    } else for (int f = tid; f < F; f += NT) f0rank[f] = -1;
    // ---- variable tables (thread 0): dense variables = dropped non-feature blocks, then kept blocks ascending in id ------
    if (tid == 0) {
        int status = 0, md = 0, mf = 0, n = 0, nb = 0;
        unsigned present = 0, drop = 0;          // bit per block id (dynamically indexed bool arrays lived in scratch memory: this serial section took 70 k cycles)
#define MG_SET(m, i) ((m) |= 1u << (i))
#define MG_HAS(m, i) (((m) >> (i)) & 1u)
        if (have_prior) for (int i = 0; i < phdr[2]; i++) MG_SET(present, phdr[3 + i]);
        if (mode == 0) {     // MARGIN_OLD
            MG_SET(present, 0); MG_SET(present, 1); MG_SET(drop, 0);
            MG_SET(present, VB_NF); MG_SET(drop, VB_NF);
            if (b.use_lidar) MG_SET(drop, 1);                                           // estimator.cpp:886-895 drop_set {0,1}
            const double *rec = b.imu + ((size_t)w * 10) * IMU_REC;
            if (rec[0] < 10.0) MG_SET(present, VB_NF + 1);                                // :896-905
            mf = s_mf;                                                                   // features observed from frame 0 (:921-950)
            for (int j = 1; j < s_maxobs && j < VB_NF; j++) MG_SET(present, j);
            if (mf > 0) MG_SET(present, 2 * VB_NF);
            if (mf > 0 && TD) MG_SET(present, 2 * VB_NF + 1);                      // para_Td of the ProjectionTdFactors (estimator.cpp:930-935)
        } else {             // MARGIN_SECOND_NEW: only the prior, drop Pose[WINDOW_SIZE-1] (:986-1003)
            if (!have_prior || !MG_HAS(present, VB_NF - 2)) status = 2;                          // nothing to do: prior stays as it is
            MG_SET(drop, VB_NF - 2);
        }
        int off = 0;
        for (int a = 0; a < VB_NF; a++) s_off_pose[a] = -1;
        s_off_sb[0] = s_off_sb[1] = -1; s_off_ex = -1; s_off_td = -1;
        for (int id = 0; id < 2 * VB_NF + 1 && status == 0; id++) if (MG_HAS(present, id) && MG_HAS(drop, id)) {
            if (id < VB_NF) { s_off_pose[id] = off; off += 6; } else if (id < 2 * VB_NF) { if (id - VB_NF < 2) s_off_sb[id - VB_NF] = off; else status = 3; off += 9; }
        }
        md = off;
        for (int id = 0; id < 2 * VB_NF + 2 && status == 0; id++) if (MG_HAS(present, id) && !MG_HAS(drop, id)) {
            int size = 7, loc = 6, sid;
            if (id < VB_NF) { s_off_pose[id] = off; }
            else if (id < 2 * VB_NF) { if (id - VB_NF < 2) s_off_sb[id - VB_NF] = off; else status = 3; size = 9; loc = 9; }
            else if (id == 2 * VB_NF) s_off_ex = off;
            else { s_off_td = off; size = 1; loc = 1; }
            // address shift (estimator.cpp:960-971 MARGIN_OLD: frame i -> i-1; :1016-1037 SECOND_NEW: frame WINDOW_SIZE -> WINDOW_SIZE-1)
            if (id >= 2 * VB_NF) sid = id;
            else if (mode == 0) sid = id - 1;
            else { const int fr = id < VB_NF ? id : id - VB_NF; sid = (fr == VB_NF - 1) ? id - 1 : id; }
            if (nb >= 24) { status = 3; break; }
            info[8 + nb] = sid; info[32 + nb] = size; info[56 + nb] = off - md; info[80 + nb] = id;
            nb++;
            off += loc;
        }
        n = off - md;
        if (n > MG_NK || md > MG_MD) status = 3;
        int M = md + mf; M += (M & 1);
        if (M > g.Mcap) status = 3;
        info[0] = status; info[1] = md; info[2] = mf; info[3] = n; info[4] = md + mf; info[5] = nb; info[6] = M;
        s_hdr[0] = status; s_hdr[1] = md; s_hdr[2] = mf; s_hdr[3] = n;
    }
    __syncthreads();
    MG_STAMP(0, 6);
    if (s_hdr[0] != 0) return;
    const int md = s_hdr[1], n = s_hdr[3], nd = md + n;
    double *Hd = g.Hd + (size_t)w * MG_ND * MG_ND, *gd = g.gd + (size_t)w * MG_ND;
    for (int i = tid; i < MG_ND; i += NT) gd[i] = 0.0;
    // ---- prior factor: dx (marginalization_factor.cpp:345-363) and column map -------------------------------------------
    if (tid < VB_PRIOR_LD) { s_dx[tid] = 0.0; s_pmap[tid] = -1; }
    __syncthreads();
    MG_STAMP(0, 7);
    if (have_prior && tid < phdr[2]) {
        const int id = phdr[3 + tid], size = phdr[27 + tid], idx = phdr[51 + tid];
        const double *x0 = b.prior_x0 + ((size_t)w * 24 + tid) * 9;
        const double *x = id < VB_NF ? s_pose + 7 * id : (id < 2 * VB_NF ? s_sb + 9 * (id - VB_NF) : (id == 2 * VB_NF ? s_ex : &s_td));
        const int doff = id < VB_NF ? s_off_pose[id] : (id < 2 * VB_NF ? s_off_sb[id - VB_NF] : (id == 2 * VB_NF ? s_off_ex : s_off_td));
        if (size == 7) {
            for (int k = 0; k < 3; k++) s_dx[idx + k] = x[k] - x0[k];
            Q dq = q_mul(q_inv(q_load(x0 + 3)), q_load(x + 3));
            const double sgn = (dq.w >= 0) ? 2.0 : -2.0;
            s_dx[idx + 3] = sgn * dq.x; s_dx[idx + 4] = sgn * dq.y; s_dx[idx + 5] = sgn * dq.z;
            for (int k = 0; k < 6; k++) s_pmap[idx + k] = doff + k;
        } else for (int k = 0; k < size; k++) { s_dx[idx + k] = x[k] - x0[k]; s_pmap[idx + k] = doff + k; }
    }
    // ---- lidarFactor[1] and IMUFactor[1] (MARGIN_OLD) ---------------------------------------------------------------------
    for (int i = tid; i < 15 * 32; i += NT) s_J[i] = 0.0;
    if (tid < 72) s_lJ[tid] = 0.0;
    if (tid < 16) s_r[tid] = 0.0;
    if (tid < 8) s_lr[tid] = 0.0;
    __syncthreads();
    MG_STAMP(0, 8);
    // dense index -> prior column (for the fill of Hd below)
    __shared__ int s_pinv[MG_ND];
    if (tid < MG_ND) s_pinv[tid] = -1;
    __syncthreads();
    if (have_prior && tid < phdr[1] && s_pmap[tid] >= 0) s_pinv[s_pmap[tid]] = tid;
    if (mode == 0) {
        if (tid == 0) {
            const double *rec = b.imu + ((size_t)w * 10) * IMU_REC;
            if (rec[0] < 10.0) {
                double r[15];
                imu_raw_eval<true, 32, false>(s_pose, s_sb, s_pose + 7, s_sb + 9, rec, b.G, r, s_J);
                for (int k = 0; k < 15; k++) s_J[32 * k + 30] = r[k];
            }
        }
        if (tid == 64 && b.use_lidar) {
            const double *lc = b.lidar + ((size_t)w * 10) * 7;
            double Ji[36], Jj[36];
            lidar_between_eval<true>(s_pose, s_pose + 7, q_load(b.qil), b.til, q_load(lc), lc + 4, s_lr, Ji, Jj);
            for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 6; c++) { s_lJ[12 * rr + c] = Ji[6 * rr + c]; s_lJ[12 * rr + 6 + c] = Jj[6 * rr + c]; }
        }
    }
    __syncthreads();
    MG_STAMP(0, 9);
    // Hd starts as the prior's H0 = J0^T J0 scattered to the dense order (zero elsewhere): one pass of stores with the loads of eighteen entries in flight, instead of a zero
    // fill and a later read-modify-write pass over the prior's entries. Order of the additions per address: prior, IMU, LiDAR, visual.
    {
        const double *__restrict__ pH0 = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
        double *__restrict__ Hd0 = Hd;
        for (int e0 = tid; e0 < MG_ND * MG_ND; e0 += 18 * NT) {
            double v6[18];
#pragma unroll
            for (int k = 0; k < 18; k++) {
                const int e = min(e0 + k * NT, MG_ND * MG_ND - 1), di = e / MG_ND, dj = e - MG_ND * di;
                const int pi = s_pinv[di], pj = s_pinv[dj];
                v6[k] = (pi >= 0 && pj >= 0) ? pH0[pi * VB_PRIOR_LD + pj] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 18; k++) { const int e = e0 + k * NT; if (e < MG_ND * MG_ND) Hd0[e] = v6[k]; }
        }
    }
    __syncthreads();
    // sqrt_info multiplication of the IMU block (31 columns incl. the residual), in registers then back
    {
        const double *S = b.imu + ((size_t)w * 10) * IMU_REC + IMU_SQRT;
        double acc[2] = {0, 0};
        int e0 = tid, e1 = tid + NT;
        for (int t = 0; t < 2; t++) {
            const int e = t ? e1 : e0;
            if (e < 15 * 31) { const int row = e / 31, col = e - 31 * row; double s = 0; for (int m2 = row; m2 < 15; m2++) s += S[15 * row + m2] * s_J[32 * m2 + col]; acc[t] = s; }
        }
        __syncthreads();
    MG_STAMP(0, 10);
        if (e0 < 15 * 31) s_J[32 * (e0 / 31) + (e0 % 31)] = acc[0];
        if (e1 < 15 * 31) s_J[32 * (e1 / 31) + (e1 % 31)] = acc[1];
    }
    __syncthreads();
    MG_STAMP(0, 11);
    // ---- visual factors of the start-frame-0 features: thread per factor -> Mbuf (slot order) ---------------------------
    double *Mb = g.Mbuf + (size_t)w * MG_MROW * FC;
    if (mode == 0) {
        // the factors of the start-frame-0 features are the pair-sorted segments of the pairs (0, j): walk only those (a scan over all factors
        // paid a chain of three dependent gathers per factor to find them), everything a factor needs in its one 64-byte record
        int ntot = 0;
#pragma unroll
        for (int jj = 0; jj < 10; jj++) ntot += s_pcn[pair_index_c(0, jj + 1)];
        for (int t = tid; t < ntot; t += NT) {
            int rem = t, q = -1;
#pragma unroll
            for (int jj = 0; jj < 10; jj++) {
                const int pj = pair_index_c(0, jj + 1), nseg = s_pcn[pj];
                if (q < 0) { if (rem < nseg) q = VB_SLOT(s_pcl[pj], s_pst[pj] + rem); else rem -= nseg; }
            }
            const double *rec = b.facrec + ((size_t)w * FC + q) * 8;
            double pts_i[3], pts_j[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { pts_i[k] = rec[k]; pts_j[k] = rec[3 + k]; }
            const long long ra = __double_as_longlong(rec[6]), rb = __double_as_longlong(rec[7]);
            const int f = (int)(ra & 0xffffffffll), slot = (int)(ra >> 32), fj = (int)((rb >> 8) & 255);
            if (t < MG_SLOTS) s_slots[t] = slot;
            double r[2], Ji[12], Jj[12], Jf[2], Jex[12], Jtd[2] = {0.0, 0.0};
            if (TD) {         // ProjectionTdFactor (estimator.cpp:930-935): the observations shifted by the pixel velocity over td (+ rolling-shutter row time)
                const int oj = b.ps_obs[(size_t)w * FC + q], oi = b.f_obs0[(size_t)w * FM + f];
                const double *vi = b.obs_vel + ((size_t)w * b.Omax + oi) * 2, *vj = b.obs_vel + ((size_t)w * b.Omax + oj) * 2;
                projection_td_eval<true>(s_pose, s_R, s_pose + 7 * fj, s_R + 9 * fj, s_ric, s_ex, pts_i, pts_j, vi, vj, s_td, b.obs_ctd[(size_t)w * b.Omax + oi], b.obs_ctd[(size_t)w * b.Omax + oj],
                                         b.obs_row[(size_t)w * b.Omax + oi] - b.row_half, b.obs_row[(size_t)w * b.Omax + oj] - b.row_half, b.tr_over_row, st_feat[f], b.sqrt_info, r, Ji, Jj, Jf, Jex, Jtd);
            } else
            projection_eval<true>(s_pose, s_R, s_pose + 7 * fj, s_R + 9 * fj, s_ric, s_ex, pts_i, pts_j, f < MG_GCH * MG_MROW ? s_rows[f] : st_feat[f], b.sqrt_info, r, Ji, Jj, Jf, Jex);
            double rho0, sw;
            cauchy(r[0] * r[0] + r[1] * r[1], b.cauchy_b, rho0, sw);
            // the 42-double row as 21 16-byte stores (a row starts at a multiple of 336 bytes): every lane writes to lines of its own, so the number of store
            // instructions is what the memory pipeline sees — 42 eight-byte stores before
            {
                typedef double mg_double2 __attribute__((ext_vector_type(2)));
                mg_double2 *row2 = reinterpret_cast<mg_double2 *>(Mb + (size_t)slot * MG_MROW);
#pragma unroll
                for (int k = 0; k < 12; k += 2) { row2[k >> 1] = mg_double2{sw * Ji[k], sw * Ji[k + 1]}; row2[6 + (k >> 1)] = mg_double2{sw * Jj[k], sw * Jj[k + 1]}; row2[12 + (k >> 1)] = mg_double2{sw * Jex[k], sw * Jex[k + 1]}; }
                row2[18] = mg_double2{sw * Jf[0], sw * Jf[1]};
                row2[19] = mg_double2{sw * r[0], sw * r[1]};
                row2[20] = mg_double2{sw * Jtd[0], sw * Jtd[1]};
            }
        }
    }
    __syncthreads();
    MG_STAMP(0, 12);
    // ---- per pair (0, j): [J0 Jj Jex Jtd r]^T [J0 Jj Jex Jtd r] (20 x 20, upper triangle; the td column is zero without estimate_td) over the pair's factors. The factor rows of a pair
    // are staged through LDS in chunks (one coalesced 320-byte row per factor) and every entry is one thread's running sum over the
    // factors in pair order — the same summation order as a per-entry gather from global memory, without its dependent loads.
    if (mode == 0) {
        const int *ps_slot = b.ps_slot + (size_t)w * FC;
        // the factor rows in evaluation order (pair after pair), MG_GCH at a time through LDS — a chunk spans several pairs, so the whole walk is a handful of memory
        // round trips (one chunk per pair, its row indices fetched first, was ten times two dependent ones); a thread's running sum is flushed where a pair ends
        if (tid < 10) { int c = 0; for (int k = 0; k <= tid; k++) c += s_pcn[pair_index_c(0, k + 1)]; s_pend[tid] = c; }
        __syncthreads();
        const int ntot = s_pend[9];
        // X^T X per pair on the MFMA: X = the pair's factor rows as 2 n x 20 ([J0 Jj Jex Jtd r], two residual rows per factor), 20 -> 32 columns = the lower tiles (0,0),
        // (1,0), (1,1), one per wave; a k-step is four X rows = two factors read from the staged chunk. (A thread per entry summing over the rows spent 250 k cycles per
        // window in LDS reads.) The accumulators live across chunks and are flushed where a pair ends.
        const int wave = tid >> 6, lane = tid & 63, l16 = lane & 15, l4 = lane >> 4;
        // All four waves work on the SAME pair: the pair's rows inside the chunk are dealt to the waves in trips of eight factors (four k-steps), a wave forms ALL THREE
        // tiles of its trips — the operands of column tile 0 and 1 are read once per k-step and feed three MFMAs ((0,0) = a0 a0, (1,0) = a1 a0, (1,1) = a1 a1), three
        // independent accumulators — and where the pair ends the four partial sums meet in LDS and are added in wave order. (Round 5, first form: whole pairs dealt to
        // the waves — a chunk of 96 rows holds one to three pairs, so one to three waves worked and the rest waited: the barrier behind the MFMAs cost as much as the
        // MFMAs. Until round 5: a wave per tile over all pairs.) An entry is a fixed-order sum over the pair's rows: same bits on every run.
        const int sub = l4 & 1;
        auto comp_of = [&](int c) -> int { return c < 18 ? 12 * (c / 6) + (c % 6) + 6 * sub : (c == 18 ? 40 + sub : (c == 19 ? 38 + sub : -1)); };      // X column -> Mbuf row component
        const int comp0 = comp_of(l16), comp1 = comp_of(16 + l16);
        mgp_double4 T00 = {0.0, 0.0, 0.0, 0.0}, T10 = {0.0, 0.0, 0.0, 0.0}, T11 = {0.0, 0.0, 0.0, 0.0};
        int jj = 0;                                              // the current pair (0, jj + 1), the same in every wave
        int pu = 0;                                              // thread e < 210 adds entry e of the packed triangle: (pu, pv), pv <= pu
        while ((pu + 1) * (pu + 2) / 2 <= min(tid, MG_PTRI - 1)) pu++;
        const int pv = min(tid, MG_PTRI - 1) - pu * (pu + 1) / 2;
        auto flush = [&]() {                                     // collective (two barriers)
            double *mine = s_part + wave * MG_PTRI;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int u0 = l4 + 4 * q, u1 = 16 + l4 + 4 * q, v0 = l16, v1 = 16 + l16;
                if (v0 <= u0) mine[u0 * (u0 + 1) / 2 + v0] = T00[q];
                if (u1 < 20) mine[u1 * (u1 + 1) / 2 + v0] = T10[q];
                if (u1 < 20 && v1 <= u1) mine[u1 * (u1 + 1) / 2 + v1] = T11[q];
            }
            T00 = mgp_double4{0.0, 0.0, 0.0, 0.0}; T10 = T00; T11 = T00;
            lds_barrier();
            if (tid < MG_PTRI) s_pm[jj * MG_PAIRM + 20 * pv + pu] = ((s_part[tid] + s_part[MG_PTRI + tid]) + s_part[2 * MG_PTRI + tid]) + s_part[3 * MG_PTRI + tid];
            lds_barrier();
            jj++;
        };
        // The chunk's rows: all of the thread's loads in flight at once (a plain loop waited for every load before issuing the next: sixteen memory round trips per
        // chunk) — first the row indices (LDS), then the loads (16 bytes each: a row is 21 of them, rows start at multiples of 336 bytes). The loads of chunk t + 1
        // are issued BEFORE the products of chunk t and land in LDS behind them (round 5; the barriers of the loop order LDS only, so they do not wait for them).
        typedef double mg_double2 __attribute__((ext_vector_type(2)));
        constexpr int RW2 = MG_MROW / 2, NLD = (MG_GCH * RW2 + NT - 1) / NT;
        static_assert(MG_MROW % 2 == 0, "rows as 16-byte pieces");
        mg_double2 vals[NLD];
        auto request = [&](int t0) {
            const int nr = min(MG_GCH, ntot - t0);
            int off[NLD];
            if (t0 + nr <= MG_SLOTS) {
#pragma unroll
                for (int k = 0; k < NLD; k++) { const int idx = min(tid + NT * k, nr * RW2 - 1), r = idx / RW2; off[k] = s_slots[t0 + r] * RW2 + (idx - RW2 * r); }
            } else {
#pragma unroll 1
                for (int k = 0; k < NLD; k++) {
                    const int idx = min(tid + NT * k, nr * RW2 - 1), r = idx / RW2;
                    int rem = t0 + r, q = 0;
                    for (int kk = 0; kk < 10; kk++) { const int pj = pair_index_c(0, kk + 1), nseg = s_pcn[pj]; if (rem >= 0) { if (rem < nseg) { q = VB_SLOT(s_pcl[pj], s_pst[pj] + rem); rem = -1; } else rem -= nseg; } }
                    const int o = ps_slot[q] * RW2 + (idx - RW2 * r);
#pragma unroll
                    for (int k2 = 0; k2 < NLD; k2++) if (k2 == k) off[k2] = o;
                }
            }
            const mg_double2 *Mb2 = reinterpret_cast<const mg_double2 *>(Mb);
#pragma unroll
            for (int k = 0; k < NLD; k++) vals[k] = Mb2[(size_t)off[k]];
        };
        MG_ACC_DECL
        if (ntot > 0) request(0);
        for (int t0 = 0; t0 < ntot; t0 += MG_GCH) {
            const int nr = min(MG_GCH, ntot - t0);
            MG_ACC(0);
            {
                mg_double2 *s_rows2 = reinterpret_cast<mg_double2 *>(s_rows);
#pragma unroll
                for (int k = 0; k < NLD; k++) { const int idx = tid + NT * k; if (idx < nr * RW2) s_rows2[idx] = vals[k]; }
            }
            MG_ACC(1);
            lds_barrier();
            if (t0 + MG_GCH < ntot) request(t0 + MG_GCH);
            MG_ACC(2);
            while (jj < 10) {                                    // the pairs that have rows in the chunk (or end empty in front of it)
                const int ps = jj ? s_pend[jj - 1] : 0, pe = s_pend[jj];
                if (pe > ps && ps >= t0 + nr) break;                                  // begins in a later chunk
                const int lo = max(ps, t0) - t0, hi = min(pe, t0 + nr) - t0;         // the pair's rows inside the chunk: [lo, hi)
                for (int fr0 = lo + 8 * wave; fr0 < hi; fr0 += 32) {                 // four k-steps (eight factors) per trip: their eight LDS reads first, then the twelve MFMAs; trips dealt to the waves
                    double a0[4], a1[4];
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {                                  // (rows past the pair's end and the columns 20 .. 31 read a zero: the select sits on the address)
                        const int fr = fr0 + 2 * k4 + (l4 >> 1);
                        a0[k4] = *((fr < hi) ? s_rows + fr * MG_MROW + comp0 : &s_zero);
                        a1[k4] = *((fr < hi && comp1 >= 0) ? s_rows + fr * MG_MROW + comp1 : &s_zero);
                    }
#pragma unroll
                    for (int k4 = 0; k4 < 4; k4++) {
                        T00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[k4], a0[k4], T00, 0, 0, 0);
                        T10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k4], a0[k4], T10, 0, 0, 0);
                        T11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[k4], a1[k4], T11, 0, 0, 0);
                    }
                }
                if (pe <= t0 + nr) flush(); else break;                               // the pair ends in this chunk / runs on into the next one
            }
            MG_ACC(3);
            lds_barrier();
            MG_ACC(4);
        }
        MG_ACC_OUT(0);
        while (jj < 10) flush();
    }
    __syncthreads();
    MG_STAMP(0, 15);
    // ---- dense-variable normal equations: owner-computes over the (nd x nd) entries ---------------------------------------
    {
        const int opose0 = s_off_pose[0], opose1 = s_off_pose[1], osb0 = s_off_sb[0], osb1 = s_off_sb[1], oex = s_off_ex;
        // IMU block variable offsets (local 30 columns): P0 6, SB0 9, P1 6, SB1 9
        auto imu_off = [&](int c) -> int { if (c < 6) return opose0 + c; if (c < 15) return osb0 + (c - 6); if (c < 21) return opose1 + (c - 15); return osb1 < 0 ? -1 : osb1 + (c - 21); };
        for (int e = tid; e < 31 * 30; e += NT) {     // IMU J^T [J r]
            const int u = e / 31, v = e - 31 * u;
            double s = 0;
            for (int row = 0; row < 15; row++) s += s_J[32 * row + u] * s_J[32 * row + (v < 30 ? v : 30)];
            const int du = imu_off(u);
            if (du < 0) continue;
            if (v < 30) { const int dv = imu_off(v); if (dv >= 0) Hd[du * MG_ND + dv] += s; }
            else gd[du] += s;
        }
        __syncthreads();
    MG_STAMP(0, 16);
        for (int e = tid; e < 13 * 12; e += NT) {     // LiDAR between-factor (unweighted jacobian, weighted residual: reference quirk)
            const int u = e / 13, v = e - 13 * u;
            double s = 0;
            for (int row = 0; row < 6; row++) s += s_lJ[12 * row + u] * (v < 12 ? s_lJ[12 * row + v] : s_lr[row]);
            const int du = (u < 6) ? opose0 + u : opose1 + (u - 6);
            if (v < 12) { const int dv = (v < 6) ? opose0 + v : opose1 + (v - 6); Hd[du * MG_ND + dv] += s; }
            else gd[du] += s;
        }
        (void)oex;
    }
    __syncthreads();
    MG_STAMP(0, 17);
    // every pass above/below writes each address at most once and passes are separated by barriers: the summation order per
    // address is fixed (IMU, LiDAR, prior, visual) => bit-reproducible.
    if (have_prior) {
        const int pn = phdr[1];
        const double *pH = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *pg = b.prior_g + (size_t)w * VB_PRIOR_LD;
        for (int i = tid; i < pn; i += NT) {
            const int di = s_pmap[i];
            if (di < 0) continue;
            double s = pg[i];
#pragma unroll 8
            for (int k = 0; k < pn; k++) s += pH[k * VB_PRIOR_LD + i] * s_dx[k];       // H0 = J0^T J0 is symmetric entry by entry: column i read along rows = coalesced across the threads
            gd[di] += s;
        }
    }
    __syncthreads();
    MG_STAMP(0, 18);
    if (mode == 0) {
        // visual dense blocks: P0 x P0, P0 x Pj, P0 x Ex, Pj x Pj, Pj x Ex, Ex x Ex (+ rhs), summed over j in fixed order
        const int o0 = s_off_pose[0], oex = s_off_ex, otd = s_off_td;
        // one thread per entry (u, v) of the pairs' 19 x 20 blocks [J0 Jj Jex Jtd | r] (v == 19: right-hand side): its target in Hd / gd is fixed unless u or v lies in the
        // Pj block, where it moves with the pair. (Walking the nd x (nd + 1) targets instead and searching, per target and pair, for the block column that maps
        // to it was 150 k cycles of integer work per window.) Same sums in the same order: pairs ascending, one += per target.
        for (int e = tid; e < 19 * 20; e += NT) {
            const int u = e / 20, v = e - 20 * u;
            const int ub = u < 6 ? o0 + u : (u < 12 ? -2 : (u < 18 ? (oex >= 0 ? oex + u - 12 : -1) : otd));          // -2: moves with the pair, -1: absent
            const int vb = v == 19 ? -3 : (v < 6 ? o0 + v : (v < 12 ? -2 : (v < 18 ? (oex >= 0 ? oex + v - 12 : -1) : otd)));    // -3: rhs
            if (ub == -1 || vb == -1) continue;
            const int pe = (u <= v) ? 20 * u + v : 20 * v + u;
            if (ub != -2 && vb != -2) {
                double sum = 0; bool any = false;
                for (int jj = 0; jj < 10; jj++) if (s_pcn[pair_index_c(0, jj + 1)] != 0) { sum += s_pm[jj * MG_PAIRM + pe]; any = true; }
                if (any) { if (vb == -3) gd[ub] += sum; else Hd[ub * MG_ND + vb] += sum; }
            } else {
                for (int jj = 0; jj < 10; jj++) {
                    const int oj = s_off_pose[jj + 1];
                    if (s_pcn[pair_index_c(0, jj + 1)] == 0 || oj < 0) continue;
                    const int du = ub == -2 ? oj + u - 6 : ub, dv = vb == -2 ? oj + v - 6 : vb;
                    double sum = 0; sum += s_pm[jj * MG_PAIRM + pe];
                    if (vb == -3) gd[du] += sum; else Hd[du * MG_ND + dv] += sum;
                }
            }
        }
        MG_STAMP(0, 20);
        // per-feature arrow rows: h_f, g_f and the P0 / Ex / td entries are sums over the feature's factors (one thread per (feature, entry), the loads of all its <= 10
        // factors in flight at once); the Pj entries come from one factor each (one thread per (feature, factor, column)). A thread per feature walking its factors
        // with some twenty dependent loads each was 75 k cycles. Same sums in the same order.
        double *Wf = g.Wf + (size_t)w * FM * MG_ND, *hfm = g.hfm + (size_t)w * FM, *gfm = g.gfm + (size_t)w * FM;
        const int mfw = s_hdr[2];
        // rank -> feature, and the feature's factor count and first Mbuf row beside it (round 5: the two loops below fetched them from global memory in front of the rows
        // they address — two dependent trips per pass of a thread, 27 passes): three tables of FRK_CAP ints in the staging area of the pair products
        constexpr int FRK_CAP = 2 * MG_GCH * MG_MROW / 3;
        int *s_frk = reinterpret_cast<int *>(s_rows), *s_fnf = s_frk + FRK_CAP, *s_ff0 = s_fnf + FRK_CAP;
        __syncthreads();
        for (int f = tid; f < F; f += NT) { const int rk = f0rank[f]; if (rk >= 0 && rk < FRK_CAP) { s_frk[rk] = f; s_fnf[rk] = f_nobs[f] - 1; s_ff0[rk] = f_fac0[f]; } }
        for (int e = tid; e < mfw * nd; e += NT) { const int rk = e / nd; Wf[(size_t)rk * MG_ND + (e - nd * rk)] = 0.0; }
        __syncthreads();
        if (mfw <= FRK_CAP) {
            const double *__restrict__ Mr = Mb;
            for (int e = tid; e < mfw * 15; e += NT) {
                const int rk = e / 15, c = e - 15 * rk, nf = s_fnf[rk], f0 = s_ff0[rk];
                const int ca = c == 0 ? 36 : (c == 1 ? 38 : (c == 2 ? 40 : (c < 9 ? c - 3 : 24 + c - 9))), cb = c < 3 ? ca + 1 : ca + 6;
                double a4[10][4];
#pragma unroll
                for (int t = 0; t < 10; t++) {
                    const double *row = Mr + (size_t)(f0 + min(t, max(nf - 1, 0))) * MG_MROW;
                    a4[t][0] = row[36]; a4[t][1] = row[37]; a4[t][2] = row[ca]; a4[t][3] = row[cb];
                }
                double sum = 0;
#pragma unroll
                for (int t = 0; t < 10; t++) if (t < nf) sum += a4[t][2] * a4[t][0] + a4[t][3] * a4[t][1];
                for (int t = 10; t < nf; t++) { const double *row = Mr + (size_t)(f0 + t) * MG_MROW; sum += row[ca] * row[36] + row[cb] * row[37]; }
                double *Wr = Wf + (size_t)rk * MG_ND;
                if (c == 0) hfm[rk] = sum; else if (c == 1) gfm[rk] = sum; else if (c == 2) { if (otd >= 0) Wr[otd] = sum; } else if (c < 9) Wr[o0 + c - 3] = sum; else Wr[oex + c - 9] = sum;
            }
            for (int e0 = tid; e0 < mfw * 60; e0 += 4 * NT) {                // (feature, factor t < 10, column c < 6), four entries' loads in flight
                double a4[4][4]; int dst[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int e = min(e0 + k * NT, mfw * 60 - 1), rk = e / 60, rem = e - 60 * rk, t = rem / 6, c = rem - 6 * t, nf = s_fnf[rk], f0 = s_ff0[rk];
                    const double *row = Mr + (size_t)(f0 + min(t, max(nf - 1, 0))) * MG_MROW;
                    a4[k][0] = row[36]; a4[k][1] = row[37]; a4[k][2] = row[12 + c]; a4[k][3] = row[18 + c];
                    dst[k] = (e0 + k * NT < mfw * 60 && t < nf) ? rk * MG_ND + s_off_pose[1 + t] + c : -1;
                }
#pragma unroll
                for (int k = 0; k < 4; k++) if (dst[k] >= 0) Wf[dst[k]] = a4[k][2] * a4[k][0] + a4[k][3] * a4[k][1];
            }
            for (int f = tid; f < F; f += NT) {                               // factors beyond the tenth of a feature (longer windows than the reference's)
                const int rk = f0rank[f]; if (rk < 0) continue;
                const int nf = f_nobs[f] - 1, f0 = f_fac0[f];
                for (int t = 10; t < nf; t++) { const double *row = Mr + (size_t)(f0 + t) * MG_MROW; for (int c = 0; c < 6; c++) Wf[(size_t)rk * MG_ND + s_off_pose[1 + t] + c] = row[12 + c] * row[36] + row[18 + c] * row[37]; }
            }
        } else
        for (int f = tid; f < F; f += NT) {
            const int rk = f0rank[f];
            if (rk < 0) continue;
            double *Wr = Wf + (size_t)rk * MG_ND;
            double h = 0, gg = 0, w0[6] = {0, 0, 0, 0, 0, 0}, wex[6] = {0, 0, 0, 0, 0, 0}, wtd = 0;
            const int nf = f_nobs[f] - 1, f0 = f_fac0[f];
            for (int t = 0; t < nf; t++) {
                const int slot = f0 + t;
                const double jf0 = Mb[(size_t)slot * MG_MROW + (36)], jf1 = Mb[(size_t)slot * MG_MROW + (37)];
                h += jf0 * jf0 + jf1 * jf1;
                gg += jf0 * Mb[(size_t)slot * MG_MROW + (38)] + jf1 * Mb[(size_t)slot * MG_MROW + (39)];
                wtd += jf0 * Mb[(size_t)slot * MG_MROW + (40)] + jf1 * Mb[(size_t)slot * MG_MROW + (41)];
                const int oj = s_off_pose[1 + t];
                for (int c = 0; c < 6; c++) {
                    w0[c] += Mb[(size_t)slot * MG_MROW + (c)] * jf0 + Mb[(size_t)slot * MG_MROW + (6 + c)] * jf1;
                    wex[c] += Mb[(size_t)slot * MG_MROW + (24 + c)] * jf0 + Mb[(size_t)slot * MG_MROW + (30 + c)] * jf1;
                    Wr[oj + c] = Mb[(size_t)slot * MG_MROW + (12 + c)] * jf0 + Mb[(size_t)slot * MG_MROW + (18 + c)] * jf1;
                }
            }
            for (int c = 0; c < 6; c++) { Wr[o0 + c] = w0[c]; Wr[oex + c] = wex[c]; }
            if (otd >= 0) Wr[otd] = wtd;
            hfm[rk] = h; gfm[rk] = gg;
        }
    }
    MG_STAMP(0, 21);
}

// ------------------------------------------------------------------------------------------------------------------
#define MG_LDS_DOUBLES (MG_MLDS * MG_MLDS)
// launched twice: exact == 0 runs the arrow fast path in ~30 KB of LDS (five workgroups per CU) and flags the windows whose guard
// fails (info[7] = 1); exact == 1 runs the Jacobi path, with the full-size LDS allocation, for the flagged windows only.
extern "C" __global__ __launch_bounds__(NT, 2) void k_marg_prepare(VbBatch b, VbMarg g) { marg_prepare_body<false>(b, g); }
extern "C" __global__ __launch_bounds__(NT) void k_marg_prepare_td(VbBatch b, VbMarg g) { marg_prepare_body<true>(b, g); }

typedef double mg_double4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(NT, 3) void k_marg_schur(VbBatch b, VbMarg g, int exact) {
    const int w = blockIdx.x, tid = threadIdx.x;
    int *info = g.info + (size_t)w * MG_INFO;
    if (info[0] != 0) return;
    if (exact == 2) { if (tid == 0) info[7] = 1; return; }      // test hook: send every window through the exact path
    if (exact && info[7] != 1) return;
    extern __shared__ double s_dyn[];
    __shared__ double s_cs[2 * 512];
    __shared__ int s_flag;
    // the exact path's workspace is a pool: this window's rank among the flagged ones decides the launch that takes it and its slot there
    int slot = w;
    if (exact) {
        __shared__ int s_rank;
        if (tid == 0) s_rank = 0;
        __syncthreads();
        int cnt = 0;
        for (int v = tid; v < w; v += NT) { const int *iv = g.info + (size_t)v * MG_INFO; cnt += (iv[0] == 0 && iv[7] == 1) ? 1 : 0; }
        if (cnt) atomicAdd(&s_rank, cnt);
        __syncthreads();
        const int rank = s_rank;
        if (rank / g.pool != g.pool_round) return;
        slot = rank % g.pool;
    }
    const int md = info[1], mf = info[2], n = info[3], m = info[4], M = info[6], H = M / 2;
    const size_t FM = b.Fmax;
    const double *Hd = g.Hd + (size_t)w * MG_ND * MG_ND, *gd = g.gd + (size_t)w * MG_ND;
    const double *Wf = g.Wf + (size_t)w * FM * MG_ND, *hfm = g.hfm + (size_t)w * FM, *gfm = g.gfm + (size_t)w * FM;
    double *A = (M <= MG_MLDS) ? s_dyn : g.Amm + (size_t)slot * g.Mcap * g.Mcap;
    const int XL = n + 1;
    double *Xg = g.X + (size_t)slot * g.Mcap * (MG_NK + 1);
    // ---- arrow fast path -------------------------------------------------------------------------------------------------
    // Amm = [[Hdd, Wd^T], [Wd, D]] with D = diag(h_f): when Amm is positive definite with every eigenvalue above the reference's
    // threshold (1e-8, marginalization_factor.cpp:270) the pseudo-inverse IS the inverse and X^T Amm^-1 X follows from eliminating
    // the diagonal D and a Cholesky of the md x md Schur complement S — no eigen-decomposition of the (md + mf)-dimensional block.
    // Guard: D > 0, S > 0 (Cholesky pivots) and trace(Amm^-1) < 1e8 (the trace bounds the largest eigenvalue of the inverse, i.e.
    // lambda_min(Amm) > 1e-8: nothing would be truncated). Otherwise fall through to the Jacobi path below.
    if (!exact) {
        double *s_S = s_dyn, *s_Y = s_S + MG_MD * MG_MD, *s_ih = s_Y + MG_MD * (MG_NK + 1), *s_red = s_ih + VILF_MAX_FEATURES_DEV;   // [md][md], [md][XL], [mf], [NT]
        __shared__ int s_ok;
        __shared__ double s_linv[MG_MD + 3];                       // 1 / L_jj of the arrow Cholesky
        MG_STAMP(2, 0);
        if (tid == 0) s_ok = (md > 0 && md <= MG_MD && mf <= VILF_MAX_FEATURES_DEV && md + n + 1 <= 16 * 7) ? 1 : 0;      // C = 7 x 7 tiles (the reference's prior: n <= 76)
        __syncthreads();
        // Every product of the fast path with the arrow rows is a block of ONE symmetric matrix C = R^T diag(1 / h_f) R, R = [W_f (md + n columns) | g_f]: the md x md block
        // of S, the md x (n + 1) block of Y and the feature part of Arr and b_r. C (<= 112 x 112, 28 lower 16 x 16 tiles, seven per wave) is accumulated by
        // v_mfma_f64_16x16x4_f64 over the features, the rows staged through LDS 16 at a time. (Three scalar loops used to read every row from global memory once per
        // ENTRY of their result: 2.5 ms per step.) Tile t of wave w is the (ti, tj) of index 7 w + t in row-major lower-triangle order.
        mg_double4 T[7];
        int tti[7], ttj[7];
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int q = 7 * (tid >> 6) + t;
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= q) ti++;
            tti[t] = ti; ttj[t] = q - ti * (ti + 1) / 2;
            T[t] = mg_double4{0.0, 0.0, 0.0, 0.0};
        }
        const int RW = md + n + 1, lane = tid & 63, l16 = lane & 15, l4 = lane >> 4;
        double *s_w = s_red + NT;                                   // [MG_FCH][MG_RWP]
        // (round 5) the rows of chunk t + 1 are requested before the products of chunk t and go to LDS behind them; the loop's barriers order LDS only. Until then
        // every chunk of sixteen rows was a full memory round trip in front of 28 MFMAs: fifteen trips per window, the larger part of the launch.
        constexpr int NLD = (MG_FCH * MG_RWP + NT - 1) / NT;
        double vals[NLD];
        auto request = [&](int f0) {
#pragma unroll
            for (int k = 0; k < NLD; k++) {
                const int e = tid + NT * k, f = e / MG_RWP, c = e - MG_RWP * f;
                const bool in = e < MG_FCH * MG_RWP && f0 + f < mf && c < RW;
                const double *src = in ? ((c < RW - 1) ? Wf + (size_t)(f0 + f) * MG_ND + c : gfm + f0 + f) : gfm;      // (a dead entry reads a live word and is zeroed below: no branch around the load)
                const double v = *src;
                vals[k] = in ? v : 0.0;
            }
        };
        if (mf > 0 && s_ok) request(0);             // the first chunk's trip runs under the 1 / h and the S, Y fills that follow
        for (int f = tid; f < mf; f += NT) { const double h = hfm[f]; if (!(h > 0.0)) s_ok = 0; s_ih[f] = 1.0 / h; }
        // the dense part of S and Y goes to LDS now — 0.5 (Hdd + Hdd^T) and [Hdr | g_d], a thread per entry, coalesced along a row — so that its memory trip runs under
        // the product loop; the tiles subtract their part of C in place afterwards. (Until round 5 every accumulator element fetched its own entries behind the loop,
        // a branch and a trip each: 30 k cycles.)
        if (s_ok) {
            const int XLc = n + 1;
            for (int e = tid; e < md * (md + XLc); e += NT) {
                const int i = e / (md + XLc), c = e - (md + XLc) * i;
                if (c < md) s_S[i * md + c] = 0.5 * (Hd[min(i, c) * MG_ND + max(i, c)] + Hd[max(i, c) * MG_ND + min(i, c)]);
                else { const int k = c - md; s_Y[i * XLc + k] = (k < n) ? Hd[i * MG_ND + md + k] : gd[i]; }
            }
        }
        __syncthreads();
        MG_STAMP(2, 1);
        if (s_ok) {
            for (int f0 = 0; f0 < mf; f0 += MG_FCH) {
                lds_barrier();
#pragma unroll
                for (int k = 0; k < NLD; k++) { const int e = tid + NT * k; if (e < MG_FCH * MG_RWP) s_w[e] = vals[k]; }
                lds_barrier();
                if (f0 + MG_FCH < mf) request(f0 + MG_FCH);
#pragma unroll
                for (int ks = 0; ks < MG_FCH / 4; ks++) {
                    const int k = 4 * ks + l4;
                    const double ihk = (f0 + k < mf) ? s_ih[f0 + k] : 0.0;
                    const double *row = s_w + k * MG_RWP;
#pragma unroll
                    for (int t = 0; t < 7; t++) T[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(row[16 * tti[t] + l16], row[16 * ttj[t] + l16] * ihk, T[t], 0, 0, 0);
                }
            }
            __syncthreads();
            MG_STAMP(2, 2);
            // S = 0.5 (Hdd + Hdd^T) - C_dd and Y = [Hdr | g_d] - C_d,(r | rhs) to LDS; the rest of C stays in the tiles for Arr below. Element q of a tile: row l4 + 4 q, column l16.
#pragma unroll
            for (int t = 0; t < 7; t++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = 16 * tti[t] + l4 + 4 * q, c = 16 * ttj[t] + l16, lo = min(r, c), hi = max(r, c);
                    if (r < c) continue;                                      // a diagonal tile holds (r, c) and (c, r): the lower one acts
                    if (hi < md) { const double v = s_S[r * md + c] - T[t][q]; s_S[r * md + c] = v; s_S[c * md + r] = v; }
                    else if (lo < md && hi < RW) { const int k = hi - md; s_Y[lo * XL + k] -= T[t][q]; }
                }
            __syncthreads();
            MG_STAMP(2, 3);
        }
        // ---- Cholesky of S, Z = L^-1 Y and the trace guard as ONE column-by-column elimination over "items", a thread each, the item's md entries in registers:
        //   items 0 .. md - 1            the rows of S            -> L (row r: entries 0 .. r)
        //   md .. md + XL - 1            the columns of Y         -> Z = L^-1 Y
        //   then md unit vectors e_c                              -> columns of L^-1 (their squares: |L^-1|_F^2)
        //   then the mf arrow rows w_f                            -> L^-1 w_f (trace(Amm^-1) = |L^-1|_F^2 + sum_f (1 / h_f + |L^-1 w_f|^2 / h_f^2))
        // Step j: the S rows publish column j (row j's entry = the pivot), ONE barrier, then every item does z_j *= 1 / sqrt(pivot) and z_c -= z_j L_cj for c > j — its
        // 20 - j multiply-adds are independent, their operands broadcast LDS reads. Until round 5 these were three phases: an in-place Cholesky with three barriers per
        // column (63), then a thread per column of Y and a thread per trace column each running the row-oriented substitution — 220 DEPENDENT multiply-adds with an LDS read
        // in front of every one: 27 + 27 + 26 k cycles of a 200 k-cycle launch. Every entry sees the same operations in the same order (the trace now multiplies by
        // 1 / L_jj where it divided: it is only compared with 1e8). Items beyond the 256 threads run the same elimination afterwards with the finished L.
        __shared__ double s_col[2][MG_MD + 3];
        double tr = 0;
        if (s_ok) {                                                  // (uniform: the flag was last written in front of a barrier)
            const int n_items = md + XL + md + mf;
            double z[MG_MD];
            auto load_item = [&](int item) {
#pragma unroll
                for (int i = 0; i < MG_MD; i++) {
                    double v = 0.0;
                    if (i < md) {
                        if (item < md) v = (i <= item) ? s_S[item * md + i] : 0.0;
                        else if (item < md + XL) v = s_Y[i * XL + (item - md)];
                        else if (item < md + XL + md) v = (i == item - md - XL) ? 1.0 : 0.0;
                        else if (item < n_items) v = Wf[(size_t)(item - md - XL - md) * MG_ND + i];
                    }
                    z[i] = v;
                }
            };
            auto store_item = [&](int item) {
                if (item < md) {
#pragma unroll
                    for (int i = 0; i < MG_MD; i++) if (i <= item) s_S[item * md + i] = z[i];
                } else if (item < md + XL) {
#pragma unroll
                    for (int i = 0; i < MG_MD; i++) if (i < md) s_Y[i * XL + (item - md)] = z[i];
                } else if (item < n_items) {
                    double sq = 0;
#pragma unroll
                    for (int i = 0; i < MG_MD; i++) if (i < md) sq += z[i] * z[i];
                    const int f = item - md - XL - md;
                    tr += (f < 0) ? sq : s_ih[f] + sq * s_ih[f] * s_ih[f];
                }
            };
            load_item(tid);
            if (tid < md) s_col[0][tid] = z[0];
            bool okp = true;
#pragma unroll
            for (int j = 0; j < MG_MD; j++) {
                if (j < md) {                                       // (uniform)
                    __syncthreads();
                    const double *col = s_col[j & 1];
                    const double piv = col[j];
                    okp = okp && (piv > 0.0);
                    const double inv = rsqrt_nr(piv > 0.0 ? piv : 1.0);
                    if (tid == 0) s_linv[j] = inv;
                    const double zj = z[j] * inv;
                    z[j] = zj;
#pragma unroll
                    for (int c = j + 1; c < MG_MD; c++) if (c < md) z[c] -= zj * (col[c] * inv);
                    if (j + 1 < MG_MD) { if (tid < md && tid >= j + 1) s_col[(j + 1) & 1][tid] = z[j + 1]; }
                }
            }
            if (!okp && tid == 0) s_ok = 0;
            __syncthreads();
            if (s_ok) store_item(tid);
            __syncthreads();
            if (s_ok) {
#pragma unroll 1
                for (int item = tid + NT; item < n_items; item += NT) {      // the items beyond the first 256: L is complete (s_S, s_linv)
                    load_item(item);
#pragma unroll
                    for (int j = 0; j < MG_MD; j++) {
                        if (j < md) {
                            const double zj = z[j] * s_linv[j];
                            z[j] = zj;
#pragma unroll
                            for (int c = j + 1; c < MG_MD; c++) if (c < md) z[c] -= zj * s_S[c * md + j];
                        }
                    }
                    store_item(item);
                }
            }
        }
        if (s_ok) {
            MG_STAMP(2, 5);
            MG_STAMP(2, 6);
            s_red[tid] = tr;
            __syncthreads();
            for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) s_red[tid] += s_red[tid + st]; __syncthreads(); }
            if (tid == 0 && !(s_red[0] < 1e8)) s_ok = 0;
            __syncthreads();
        }
        if (s_ok) {
            double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK, *br = g.br + (size_t)w * MG_NK;
            MG_STAMP(2, 7);
            // Arr - (feature part, in the tiles) - Z^T Z: the second product by MFMA too (K = md, rows of Z = L^-1 Y from LDS, shifted by md against the tile grid)
#pragma unroll
            for (int ks = 0; ks < (MG_MD + 3) / 4; ks++) {
                const int k = 4 * ks + l4;
#pragma unroll
                for (int t = 0; t < 7; t++) {
                    const int ca = 16 * tti[t] + l16 - md, cb = 16 * ttj[t] + l16 - md;
                    const double za = (k < md && ca >= 0 && ca < XL) ? s_Y[k * XL + ca] : 0.0, zb = (k < md && cb >= 0 && cb < XL) ? s_Y[k * XL + cb] : 0.0;
                    T[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(za, zb, T[t], 0, 0, 0);
                }
            }
            MG_STAMP(2, 8);
#pragma unroll
            for (int t = 0; t < 7; t++) {
                // the tile's entries of Hd / g_d first, all eight in flight (a dead element reads a live word: no branch around a load), then the stores
                double ha[4], hb[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = 16 * tti[t] + l4 + 4 * q - md, j = 16 * ttj[t] + l16 - md;       // C index -> kept index (n = the right-hand side column / row)
                    const bool in = !(i < 0 || j < 0 || i > n || j > n);
                    const bool a_on = in && i < n, b_on = in && tti[t] != ttj[t] && j < n;
                    ha[q] = *(a_on ? ((j < n) ? Hd + (md + i) * MG_ND + md + j : gd + md + i) : gd);
                    hb[q] = *(b_on ? ((i < n) ? Hd + (md + j) * MG_ND + md + i : gd + md + j) : gd);
                }
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int i = 16 * tti[t] + l4 + 4 * q - md, j = 16 * ttj[t] + l16 - md;
                    if (i < 0 || j < 0 || i > n || j > n) continue;
                    const double sacc = T[t][q];
                    if (i < n) { if (j < n) Ar[i * MG_NK + j] = ha[q] - sacc; else br[i] = ha[q] - sacc; }
                    if (tti[t] != ttj[t] && j < n) { if (i < n) Ar[j * MG_NK + i] = hb[q] - sacc; else br[j] = hb[q] - sacc; }
                }
            }
            MG_STAMP(2, 9);
            if (tid == 0) info[7] = 0;
            return;
        }
        if (tid == 0) info[7] = 1;      // guard failed: the exact launch takes this window
        return;
    }
    // Amm = 0.5 (Amm + Amm^T) (marginalization_factor.cpp:267), arrow structure: dense md x md block, feature diagonal
    for (int e = tid; e < M * M; e += NT) {
        const int i = e / M, j = e - M * i;
        double v = 0;
        if (i < m && j < m) {
            if (i < md && j < md) v = 0.5 * (Hd[i * MG_ND + j] + Hd[j * MG_ND + i]);
            else if (i < md) v = Wf[(size_t)(j - md) * MG_ND + i];
            else if (j < md) v = Wf[(size_t)(i - md) * MG_ND + j];
            else if (i == j) v = hfm[i - md];
        }
        A[e] = v;
    }
    for (int e = tid; e < M * XL; e += NT) {        // X = [Amr | bmm]
        const int i = e / XL, k = e - XL * i;
        double v = 0;
        if (i < md) v = (k < n) ? Hd[i * MG_ND + md + k] : gd[i];
        else if (i < m) v = (k < n) ? Wf[(size_t)(i - md) * MG_ND + md + k] : gfm[i - md];
        Xg[e] = v;
    }
    __syncthreads();
    double *rot = g.rot + (size_t)slot * MG_SWEEPS * (size_t)(g.Mcap - 1) * g.Mcap;
    const int sweeps = jacobi_eig<false>(A, M, M, nullptr, rot, s_cs, &s_flag);
    double *lam = g.lam + (size_t)slot * g.Mcap;
    for (int i = tid; i < M; i += NT) lam[i] = A[i * M + i];
    __syncthreads();
    // replay the rotations on X (X' = V^T X); X lives in LDS when it fits
    double *X = (M * XL <= MG_LDS_DOUBLES) ? s_dyn : Xg;
    if (X != Xg) { for (int e = tid; e < M * XL; e += NT) X[e] = Xg[e]; }
    __syncthreads();
    for (int sw = 0; sw < sweeps; sw++)
        for (int round = 0; round < M - 1; round++) {
            const double *l = rot + ((size_t)(sw * (M - 1) + round) * H) * 2;
            for (int it = tid; it < H * XL; it += NT) {
                const int k = it / XL, j = it - k * XL;
                int p, q; rr_pair(round, k, M, p, q);
                const double c = l[2 * k], s = l[2 * k + 1];
                const double a = X[p * XL + j], bq = X[q * XL + j];
                X[p * XL + j] = c * a - s * bq; X[q * XL + j] = s * a + c * bq;
            }
            __syncthreads();
        }
    // A = Arr - Arm Amm^+ Amr ; b = brr - Arm Amm^+ bmm  with Amm^+ = V diag(lam > eps ? 1/lam : 0) V^T  (:267-281)
    const double eps = 1e-8;
    double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK, *br = g.br + (size_t)w * MG_NK;
    for (int e = tid; e < n * XL; e += NT) {
        const int i = e / XL, j = e - XL * i;
        double s = 0;
        for (int t = 0; t < M; t++) { const double lt = lam[t]; if (lt > eps) s += X[t * XL + i] * X[t * XL + j] / lt; }
        if (j < n) Ar[i * MG_NK + j] = Hd[(md + i) * MG_ND + md + j] - s;
        else br[i] = gd[md + i] - s;
    }
    (void)mf;
}

// Symmetric eigen-decomposition of the n x n matrix held in V (LDS, row-major, leading dimension ld) by Householder
// tridiagonalisation + implicit QL — the EISPACK tred2 / tql2 pair the CPU restatement uses (oracle/omath.cpp sym_eigen), restated
// for one workgroup: the O(n^2) inner loops of every step run across the threads with the SAME per-element operation order
// (ascending-k dot products), the three length-i reductions of a tred2 step by wave 0, the scalar QL recurrence by thread 0 and
// the rotations of one QL iteration applied to all rows of V in parallel. On exit V(:, j) is eigenvector j of eigenvalue d[j]
// (unsorted). d, e: LDS arrays of n doubles; s_cs: 2 n doubles; s_sc: 4 doubles; s_ctl: 2 ints.
__device__ __forceinline__ double mg_wave_sum(double v) { return vilf_wave_sum64(v); }       // every caller has its whole wave active
#define VV(r, c) V[(r) * ld + (c)]
// Householder tridiagonalisation with accumulated transformations (EISPACK tred2): d = diagonal, e[0 .. n-2] = sub-diagonal, V = Q
__device__ void tred2_part(double *V, int n, int ld, double *d, double *e, double *s_sc) {
    // Per Householder step the work falls into two wide phases (the mat-vec and the rank-2 update, all threads) and two narrow ones (the three length-i reductions with
    // the element-wise passes between them): the narrow ones run inside wave 0 — lane-strided, ordered by wave-level syncs — while the other waves wait at ONE barrier,
    // so a step is four workgroup barriers (it used to be nine: every reduction and every element-wise pass was its own barrier phase). A one-wave-per-window form
    // (no barriers at all) was measured 1.5 x SLOWER: one wave cannot keep enough LDS requests in flight, and the 46 KB matrix allows three windows per CU either way.
    const int tid = threadIdx.x;
#define W0SYNC __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier()
    if (tid < 64) for (int k = tid; k < n; k += 64) d[k] = VV(n - 1, k);
    MG_ACC_DECL
    for (int i = n - 1; i > 0; i--) {
        // ---- narrow phase A (wave 0): scale, scaled d, h, e[i], d[i-1] (d was loaded by this wave at the end of the previous step)
        if (tid < 64) {
            W0SYNC;
            double a = 0;
            for (int k = tid; k < i; k += 64) a += fabs(d[k]);
            const double scale = mg_wave_sum(a);
            if (scale == 0.0) {
                if (tid == 0) { e[i] = d[i - 1]; s_sc[0] = 0.0; }
            } else {
                a = 0;
                for (int k = tid; k < i; k += 64) { const double t = d[k] / scale; d[k] = t; a += t * t; }
                a = mg_wave_sum(a);
                W0SYNC;
                if (tid == 0) { double h = a; const double f = d[i - 1]; double gq = sqrt(h); if (f > 0) gq = -gq; e[i] = scale * gq; h = h - f * gq; d[i - 1] = f - gq; s_sc[1] = h; s_sc[0] = scale; }
            }
        }
        MG_ACC(0);
        __syncthreads();
        MG_ACC(1);
        const double scale = s_sc[0];
        if (scale == 0.0) {
            if (tid < i) { VV(i, tid) = 0.0; VV(tid, i) = 0.0; }
            __syncthreads();
            if (tid < 64) { for (int k = tid; k < i; k += 64) d[k] = VV(i - 1, k); if (tid == 0) d[i] = 0.0; }
            continue;
        }
        const double h = s_sc[1];
        {                                    // e = A_sub d with the symmetric matrix read from its lower triangle; V(:, i) keeps the reflector.
            // Row j's dot product is split over `parts` adjacent lanes (interleaved k) and combined by shuffles: the dependent chain is i / parts long
            const int lg = (4 * i <= NT) ? 2 : ((2 * i <= NT) ? 1 : 0), parts = 1 << lg;
            const int j = tid >> lg, part = tid & (parts - 1);
            double gq = 0;
            if (j < i) {
                int k = part;
                for (; k + 7 * parts < i; k += 8 * parts) {             // eight terms per trip, their loads issued together
                    double a4[8], d4[8];
#pragma unroll
                    for (int u = 0; u < 8; u++) { const int kk = k + u * parts; a4[u] = (kk <= j ? VV(j, kk) : VV(kk, j)); d4[u] = d[kk]; }
#pragma unroll
                    for (int u = 0; u < 8; u++) gq += a4[u] * d4[u];
                }
                for (; k + 3 * parts < i; k += 4 * parts) {
                    double a4[4], d4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int kk = k + u * parts; a4[u] = (kk <= j ? VV(j, kk) : VV(kk, j)); d4[u] = d[kk]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) gq += a4[u] * d4[u];
                }
                for (; k < i; k += parts) gq += (k <= j ? VV(j, k) : VV(k, j)) * d[k];
            }
            if (lg >= 1) gq += vilf_dpp_f64<0xB1>(gq);      // lane ^ 1 by DPP (every lane takes part)
            if (lg >= 2) gq += vilf_dpp_f64<0x4E>(gq);      // lane ^ 2
            if (j < i && part == 0) { e[j] = gq / h; VV(j, i) = d[j]; }
        }
        MG_ACC(2);
        __syncthreads();
        MG_ACC(1);
        // ---- narrow phase B (wave 0): hh and e -= hh d
        if (tid < 64) {
            double a = 0;
            for (int k = tid; k < i; k += 64) a += e[k] * d[k];
            const double hh = mg_wave_sum(a) / (h + h);
            for (int k = tid; k < i; k += 64) e[k] -= hh * d[k];
        }
        MG_ACC(3);
        __syncthreads();
        MG_ACC(1);
        {   // rank-2 update of the lower triangle only: rows r and i-1-r together fill one row of an (i+1)-wide rectangle; the row index comes from a
            // float reciprocal (exact for these sizes) instead of an integer division per element. (The kernel is bound by LDS throughput — three windows share a
            // CU's LDS pipe — so what counts is the number of wave-level LDS instructions: a row-per-wave form with d[j], e[j] in registers issues more of them,
            // half its lanes idle, and was 1.7 x slower.)
            const int w = i + 1, nr = (i + 1) >> 1;
            const float invw = 1.0f / (float)w;
            for (int t = tid; t < nr * w; t += NT) {
                const int r = (int)(((float)t + 0.5f) * invw), c = t - r * w;
                int k, j;
                if (c <= r) { k = r; j = c; } else { k = i - 1 - r; j = c - (r + 1); if (k == r) continue; }
                VV(k, j) -= (d[j] * e[k] + e[j] * d[k]);
            }
        }
        MG_ACC(4);
        __syncthreads();
        MG_ACC(1);
        // the next step's d = row i - 1 (wave 0, which is the one that reads it next); row i of V is cleared by everyone
        if (tid < 64) { for (int k = tid; k < i; k += 64) d[k] = VV(i - 1, k); if (tid == 0) d[i] = h; }
        if (tid < i) VV(i, tid) = 0.0;
    }
    __syncthreads();
#undef W0SYNC
    MG_ACC(5);
    MG_ACC_OUT(1);
    for (int i = 0; i < n - 1; i++) {        // accumulate the transformations
        if (tid == 0) { VV(n - 1, i) = VV(i, i); VV(i, i) = 1.0; }
        const double h = d[i + 1];
        __syncthreads();
        if (h != 0.0) {
            if (tid <= i) d[tid] = VV(tid, i + 1) / h;
            __syncthreads();
            {                                // column j of the accumulated transformation, split over `parts` adjacent lanes like the mat-vec above
                const int lg = (4 * (i + 1) <= NT) ? 2 : ((2 * (i + 1) <= NT) ? 1 : 0), parts = 1 << lg;
                const int j = tid >> lg, part = tid & (parts - 1);
                double gq = 0;
                if (j <= i) {
                    int k = part;
                    for (; k + 3 * parts <= i; k += 4 * parts) {
                        double a4[4], b4[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { a4[u] = VV(k + u * parts, i + 1); b4[u] = VV(k + u * parts, j); }
#pragma unroll
                        for (int u = 0; u < 4; u++) gq += a4[u] * b4[u];
                    }
                    for (; k <= i; k += parts) gq += VV(k, i + 1) * VV(k, j);
                }
                if (lg >= 1) gq += vilf_dpp_f64<0xB1>(gq);      // lane ^ 1 by DPP (every lane takes part)
                if (lg >= 2) gq += vilf_dpp_f64<0x4E>(gq);      // lane ^ 2
                if (j <= i) {
                    int k = part;
                    for (; k + 3 * parts <= i; k += 4 * parts) {
                        double v4[4], d4[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) { v4[u] = VV(k + u * parts, j); d4[u] = d[k + u * parts]; }
#pragma unroll
                        for (int u = 0; u < 4; u++) VV(k + u * parts, j) = v4[u] - gq * d4[u];
                    }
                    for (; k <= i; k += parts) VV(k, j) -= gq * d[k];
                }
            }
            __syncthreads();
        }
        if (tid <= i) VV(tid, i + 1) = 0.0;
        __syncthreads();
    }
    if (tid < n) { d[tid] = VV(n - 1, tid); VV(n - 1, tid) = 0.0; }
    __syncthreads();
    if (tid == 0) { VV(n - 1, n - 1) = 1.0; for (int i = 1; i < n; i++) e[i - 1] = e[i]; e[n - 1] = 0.0; }
    __syncthreads();
    MG_STAMP(1, 10);
}
// implicit QL on the tridiagonal matrix, rotations applied to the rows of V (EISPACK tql2), one workgroup
__device__ void tql2_part(double *V, int n, int ld, double *d, double *e, double *s_cs, int *s_ctl) {
    const int tid = threadIdx.x;
    double f = 0.0, tst1 = 0.0;              // live in thread 0 only
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; l++) {
        if (tid == 0) {
            tst1 = fmax(tst1, fabs(d[l]) + fabs(e[l]));
            int m = l;
            while (m < n) { if (fabs(e[m]) <= eps * tst1) break; m++; }
            s_ctl[0] = m;
        }
        __syncthreads();
        const int m = s_ctl[0];
        if (m > l) {
            int iter = 0;
            for (;;) {
                if (tid == 0) {
                    iter++;
                    double gq = d[l];
                    double p = (d[l + 1] - gq) / (2.0 * e[l]);
                    double r = hypot(p, 1.0);
                    if (p < 0) r = -r;
                    d[l] = e[l] / (p + r);
                    d[l + 1] = e[l] * (p + r);
                    const double dl1 = d[l + 1];
                    double hq = gq - d[l];
                    for (int i = l + 2; i < n; i++) d[i] -= hq;
                    f += hq;
                    p = d[m];
                    double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                    const double el1 = e[l + 1];
                    for (int i = m - 1; i >= l; i--) {
                        c3 = c2; c2 = c; s2 = s;
                        gq = c * e[i];
                        hq = c * p;
                        {   // r = hypot(p, e[i]), s = e[i] / r, c = p / r through one reciprocal square root: this recurrence is the serial
                            // critical path of the whole decomposition (the magnitudes here are far from overflow)
                            const double ei = e[i], rr = p * p + ei * ei;
                            const double inv = rr > 0.0 ? rsqrt_nr(rr) : 0.0;
                            r = rr * inv;
                            e[i + 1] = s * r;
                            s = ei * inv;
                            c = p * inv;
                        }
                        p = c * d[i] - s * gq;
                        d[i + 1] = hq + s * (c * gq + s * d[i]);
                        s_cs[2 * i] = c; s_cs[2 * i + 1] = s;
                    }
                    p = -s * s2 * c3 * el1 * e[l] / dl1;
                    e[l] = s * p;
                    d[l] = c * p;
                    s_ctl[1] = (fabs(e[l]) > eps * tst1 && iter < 64) ? 1 : 0;
                }
                __syncthreads();
                const int more = s_ctl[1];
                if (tid < n) {                   // row k of the eigenvector matrix through the whole rotation sequence; the running column stays in a register
                    const int k = tid;
                    double hq = VV(k, m);
                    for (int i = m - 1; i >= l; i--) {
                        const double c = s_cs[2 * i], sn = s_cs[2 * i + 1];
                        const double vi = VV(k, i);
                        VV(k, i + 1) = sn * vi + c * hq;
                        hq = c * vi - sn * hq;
                    }
                    VV(k, l) = hq;
                }
                __syncthreads();
                if (!more) break;
            }
        }
        if (tid == 0) { d[l] += f; e[l] = 0.0; }
        __syncthreads();
    }
}
__device__ void tred2_tql2(double *V, int n, int ld, double *d, double *e, double *s_cs, double *s_sc, int *s_ctl) {
    tred2_part(V, n, ld, d, e, s_sc);
    tql2_part(V, n, ld, d, e, s_cs, s_ctl);
}

// block table with the address shift + x0 = parameter_block_data (getParameterBlocks, :299-319) of the new prior of window w
__device__ void mf_table(const VbBatch &b, const VbMarg &g, int w, int n, int nb, const int *info) {
    const int tid = threadIdx.x;
    int *hdr = g.prior_hdr_out + (size_t)w * VB_PRIOR_HDR;
    if (tid == 0) { hdr[0] = 1; hdr[1] = n; hdr[2] = nb; hdr[75] = info[4]; }     // [75] = m (marginalized dimension, informative)
    if (tid < 24) {
        hdr[3 + tid] = tid < nb ? info[8 + tid] : 0; hdr[27 + tid] = tid < nb ? info[32 + tid] : 0; hdr[51 + tid] = tid < nb ? info[56 + tid] : 0;
        if (tid < nb) {
            const int id = info[80 + tid];
            const double *x = id < VB_NF ? g.st_pose + (size_t)w * 77 + 7 * id : (id < 2 * VB_NF ? g.st_sb + (size_t)w * 99 + 9 * (id - VB_NF) : (id == 2 * VB_NF ? g.st_ex + (size_t)w * 7 : b.td + w));
            double *x0 = g.prior_x0_out + ((size_t)w * 24 + tid) * 9;
            const int size = info[32 + tid];
            for (int k = 0; k < 9; k++) x0[k] = k < size ? x[k] : 0.0;
        }
    }
}
// launched twice: windows with n_lo <= n < n_hi only. The usual kept dimension (n <= 77) needs < 48 KB of LDS for its n x n matrix,
// so three workgroups share a CU; the rare larger priors go through the second launch with the full-size allocation.
// eigenvalues in s_lam, eigenvectors in the columns of V -> the new prior: J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b, block table, x0
__device__ void mf_tail(const VbBatch &b, const VbMarg &g, int w, const double *V, int N, int n, int nb, const int *info, const double *s_lam, const double *s_br, int *s_rank) {
    const int tid = threadIdx.x;
    if (tid < n) {      // ascending order like Eigen::SelfAdjointEigenSolver
        int rk = 0;
        for (int j = 0; j < n; j++) if (s_lam[j] < s_lam[tid] || (s_lam[j] == s_lam[tid] && j < tid)) rk++;
        s_rank[tid] = rk;
    }
    __syncthreads();
    const double eps = 1e-8;
    double *Jo = g.prior_J_out + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *ro = g.prior_r_out + (size_t)w * VB_PRIOR_LD;
    for (int e = tid; e < n * n; e += NT) {          // linearized_jacobians = sqrt(S) V^T  (:283-290), leading dimension n
        const int i = e / n, k = e - n * i;
        const double S = s_lam[i] > eps ? s_lam[i] : 0.0;
        Jo[(size_t)s_rank[i] * n + k] = sqrt(S) * V[k * N + i];
    }
    if (tid < n) {                                  // linearized_residuals = S^-1/2 V^T b  (:291)
        const double Sinv = s_lam[tid] > eps ? 1.0 / s_lam[tid] : 0.0;
        double s = 0;
        for (int k = 0; k < n; k++) s += V[k * N + tid] * s_br[k];
        ro[s_rank[tid]] = sqrt(Sinv) * s;
    }
    mf_table(b, g, w, n, nb, info);
}

// ---- kept block without an eigen-decomposition ------------------------------------------------------------------------------------
// marginalize() ends with A = V S V^T, J0 = sqrt(S) V^T, r0 = S^-1/2 V^T b, eigenvalues below 1e-8 truncated (marginalization_factor.cpp:283-291).
// Everything downstream of the prior — MarginalizationFactor::Evaluate's r0 + J0 dx inside a least-squares cost (:333-381), the next marginalization's
// J^T J / J^T r — sees (J0, r0) only through J0^T J0, J0^T r0 and |r0|^2, which are invariant under an orthogonal transform of the rows. When NO eigenvalue is
// truncated, J0^T J0 = A and J0^T r0 = b, so the Cholesky factor serves as well: A = L L^T, J0 = L^T, r0 = L^-1 b (|r0|^2 = b^T A^-1 b in both forms).
// Guard: every pivot positive and trace(A^-1) = |L^-1|_F^2 < 1e8, which bounds the largest eigenvalue of A^-1, i.e. lambda_min(A) > 1e-8: the reference would
// truncate nothing. A window that fails the guard (the gauge-deficient priors of a window chain that started without a prior: four eigenvalues at rounding level)
// is left to the eigen-solver launches below (qlInfo[3] = 0).
// One workgroup per window, everything in one n x N LDS array: the lower triangle becomes L column by column (right-looking); the strict upper triangle holds
// X = L^-1 transposed (X[i][c] at V[c][i]), built by the fan-out form of the forward substitution IN THE SAME column loop: once column j of L is final, row j of X is
// final too (X[j][c] /= L[j][j], X[j][j] = 1 / L[j][j]) and every later row takes X[i][c] -= L[i][j] X[j][c] (c <= j) beside the trailing update
// A[i][c] -= L[i][j] L[c][j] (j < c <= i). Two barriers per column, no storage besides the matrix itself; the diagonal of L lives in s_dg (V[j][j] keeps the pivot).
extern "C" __global__ __launch_bounds__(NT) void k_mf_chol(VbBatch b, VbMarg g, int n_lo, int n_hi, int disable) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int *info = g.info + (size_t)w * MG_INFO;
    int *qi = g.qlInfo + (size_t)w * 4;
    if (info[0] != 0) { if (tid == 0 && n_lo == 0) qi[3] = 0; return; }     // no new prior for this window: the flag must not keep an earlier call's value (k_prior_prep reads it)
                                                                            // (n_lo > 0 in the first launch: k_mf_chol_tiles ran before and has cleared it)
    if (info[3] < n_lo || info[3] >= n_hi) return;
    if (disable) { if (tid == 0) qi[3] = 0; return; }          // test hook: every window through the eigen-solver
    extern __shared__ double s_dyn[];
    __shared__ double s_dg[MG_NK + 2], s_br[MG_NK + 2], s_red[NT / 64];
    __shared__ int s_ok;
    const int n = info[3], nb = info[5], N = n | 1;
    double *V = s_dyn;
    const double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK, *br = g.br + (size_t)w * MG_NK;
    for (int e = tid; e < n * n; e += NT) { const int i = e / n, j = e - n * i; V[i * N + j] = (j <= i) ? 0.5 * (Ar[i * MG_NK + j] + Ar[j * MG_NK + i]) : 0.0; }
    if (tid < n) s_br[tid] = br[tid];
    if (tid == 0) s_ok = 1;
    __syncthreads();
    const int tr = tid >> 2, tc = tid & 3;                       // rows j + 1 + tr (+ 64); every fourth element of a row's run
    for (int j = 0; j < n; j++) {
        // phase 1: column j of L and row j of X. Every thread takes the pivot from LDS itself (no broadcast phase).
        const double d = V[j * N + j];
        // 1 / sqrt(d) by v_rsq_f64 + two Newton steps, the column scaled by a multiplication: an IEEE square root, a division for 1 / l and a division per entry were
        // ~35 dependent fp64 operations at 36 cycles each at the head of every one of the n columns (this Cholesky form is this library's own — the reference's
        // eigen-decomposition is the fallback — and the results differ in the last bit only)
        const double dd = d > 0.0 ? d : 1.0, linv = rsqrt_nr(dd), l = dd * linv;
        if (tid < n) {
            if (tid != j) V[tid * N + j] *= linv;                // tid > j: L[tid][j]; tid < j: X[j][tid] (stored transposed)
            else { s_dg[j] = l; if (!(d > 0.0)) s_ok = 0; }
        }
        __syncthreads();
        if (!s_ok) break;                                        // uniform: written before the barrier, read by every thread after it
        // phase 2: rows i > j; nothing here writes column j, which both loops read. Four elements per trip, every operand loaded before the first store: a loop of
        // load - multiply - store trips on one LDS array is kept in program order by the compiler (possible aliasing) and pays the LDS latency per element
        for (int i = j + 1 + tr; i < n; i += 64) {
            const double lij = V[i * N + j];
            for (int c0 = tc; c0 < j; c0 += 16) {                                             // X[i][c] -= L[i][j] X[j][c]
                double v[4], xj[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int c = min(c0 + 4 * u, j - 1); v[u] = V[c * N + i]; xj[u] = V[c * N + j]; }
#pragma unroll
                for (int u = 0; u < 4; u++) if (c0 + 4 * u < j) V[(c0 + 4 * u) * N + i] = v[u] - lij * xj[u];
            }
            for (int c0 = j + 1 + tc; c0 <= i; c0 += 16) {                                    // A[i][c] -= L[i][j] L[c][j]
                double v[4], lc[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { const int c = min(c0 + 4 * u, i); v[u] = V[i * N + c]; lc[u] = V[c * N + j]; }
#pragma unroll
                for (int u = 0; u < 4; u++) if (c0 + 4 * u <= i) V[i * N + c0 + 4 * u] = v[u] - lij * lc[u];
            }
            if (tc == 0) V[j * N + i] -= lij * linv;                                          // X[i][j] -= L[i][j] X[j][j]
        }
        __syncthreads();
    }
    if (!s_ok) { if (tid == 0) qi[3] = 0; return; }
    // trace(A^-1) = |X|_F^2
    double sq = 0.0;
    for (int e = tid; e < n * n; e += NT) { const int c = e / n, i = e - n * c; if (i > c) { const double x = V[c * N + i]; sq += x * x; } }
    if (tid < n) { const double x = 1.0 / s_dg[tid]; sq += x * x; }
    sq = mg_wave_sum(sq);
    if ((tid & 63) == 0) s_red[tid >> 6] = sq;
    __syncthreads();
    double trc = 0.0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) trc += s_red[k];
    if (!(trc < 1e8)) { if (tid == 0) qi[3] = 0; return; }
    // the new prior: linearized_jacobians = L^T (leading dimension n), linearized_residuals = L^-1 b
    double *Jo = g.prior_J_out + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *ro = g.prior_r_out + (size_t)w * VB_PRIOR_LD;
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, k = e - n * i;
        Jo[e] = k > i ? V[k * N + i] : (k == i ? s_dg[i] : 0.0);
    }
    if (tid < n) {
        double s = s_br[tid] / s_dg[tid];
        for (int c = 0; c < tid; c++) s += V[c * N + tid] * s_br[c];
        ro[tid] = s;
    }
    // H0 = J0^T J0 = L L^T = A and g0 = J0^T r0 = b: written from the kept block itself, k_prior_prep skips this window
    double *Ho = g.prior_H_out + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *go = g.prior_g_out + (size_t)w * VB_PRIOR_LD;
    for (int e = tid; e < n * n; e += NT) { const int i = e / n, j = e - n * i; Ho[i * VB_PRIOR_LD + j] = 0.5 * (Ar[i * MG_NK + j] + Ar[j * MG_NK + i]); }
    if (tid < n) go[tid] = s_br[tid];
    mf_table(b, g, w, n, nb, info);
    if (tid == 0) qi[3] = 1;
}

// ---- the kept block's Cholesky form for n <= 75, on the matrix cores (round 5) ---------------------------------------------------------------------------------------
// k_mf_chol above is a column-by-column Cholesky that carries X = L^-1 along (for the guard trace(A^-1) = |L^-1|_F^2 and r0 = L^-1 b): two barriers and two passes over
// the trailing matrix per column, 410 k cycles per window. This kernel factors the AUGMENTED matrix [A; b^T; I] the way k_solve_sb factors its dense block: the trailing
// matrix lives in MFMA accumulator tiles (16 x 16, ten per wave), per 4-column panel the panel's columns go through LDS, a thread per ROW factors the 4 x 4 diagonal block
// itself and solves its own row strip, and the rank-4 update is one MFMA per tile. The rows under A are right-hand sides: row 75 = b^T comes out as (L^-1 b)^T = r0, the
// identity rows 80 + c as row c of L^-T — their squares are summed where the row threads produce them: |L^-1|_F^2 without ever storing L^-1. n < 75 is padded with an
// identity block (pivots 1, no coupling). Same guard, same outputs as k_mf_chol (J0 = L^T, r0, H0 = A, g0 = b, block table); a window that fails the guard is left to the
// eigen-solver launches exactly as before. ~34 KB of LDS: four workgroups per CU.
#define MFT_ROWS 160                                      // 80 (A, b, padding) + 80 (identity rows, padding)
#define MFT_NT10 10                                       // tiles per wave: 40 = 15 (rows 0..79, lower) + 25 (rows 80..159, all five column tiles)
#define MFT_LDS_DOUBLES (SB_NR * (SB_NR + 1) / 2 + 2 * 4 * MFT_ROWS + 16)
extern "C" __global__ __launch_bounds__(NT, 2) void k_mf_chol_tiles(VbBatch b, VbMarg g, int disable) {
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int *info = g.info + (size_t)w * MG_INFO;
    int *qi = g.qlInfo + (size_t)w * 4;
    if (info[0] != 0) { if (tid == 0) qi[3] = 0; return; }       // no new prior for this window: the flag must not keep an earlier call's value (k_prior_prep reads it)
    if (info[3] > SB_ND) return;                                  // wider kept blocks: k_mf_chol
    if (disable) { if (tid == 0) qi[3] = 0; return; }             // test hook: every window through the eigen-solver
    extern __shared__ double s_dyn[];
    double *s_P = s_dyn, *s_pan = s_P + SB_NR * (SB_NR + 1) / 2, *s_lp = s_pan + 4 * MFT_ROWS;      // packed lower rows 0..75; the panel's columns [row][4]; the panel's factor rows [row][4]
    __shared__ double s_red[NT / 64];
    __shared__ int s_ok;
    const int n = info[3], nb = info[5];
    const double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK, *br = g.br + (size_t)w * MG_NK;
    auto prow = [](int r) { return r * (r + 1) / 2; };
    // the packed lower triangle: A symmetrised (as k_mf_chol), identity padding for rows n..74, row 75 = b. The thread's twelve entries: both loads of every entry in flight
    // at once (a dead entry reads a live word), then the stores — and H0 = A, g0 = b go out from here, symmetrised entry by entry, instead of from a second pass over A behind
    // the factorisation (whose loop of two loads and a store per trip was a memory round trip per entry: the larger part of this kernel until round 5). A window that fails
    // the guard below is taken by the eigen-solver launches, whose k_prior_prep rewrites H0 and g0.
    double *Ho = g.prior_H_out + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *go = g.prior_g_out + (size_t)w * VB_PRIOR_LD;
    MG_STAMP(3, 0);
    {
        constexpr int NPK = (SB_NR * (SB_NR + 1) / 2 + NT - 1) / NT;
        double va[NPK], vb[NPK];
        int rr[NPK], cc[NPK];
#pragma unroll
        for (int k = 0; k < NPK; k++) {
            const int e = min(tid + NT * k, SB_NR * (SB_NR + 1) / 2 - 1);
            int r = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            if (prow(r) > e) r--;
            if (prow(r + 1) <= e) r++;
            const int c = e - prow(r);
            rr[k] = r; cc[k] = c;
            va[k] = *((r < n) ? Ar + r * MG_NK + c : ((r >= SB_ND && c < n) ? br + c : br));
            vb[k] = *((r < n) ? Ar + c * MG_NK + r : br);
        }
#pragma unroll
        for (int k = 0; k < NPK; k++) {
            const int e = tid + NT * k, r = rr[k], c = cc[k];
            if (e < SB_NR * (SB_NR + 1) / 2) {
                double v;
                if (r < n) { v = 0.5 * (va[k] + vb[k]); Ho[r * VB_PRIOR_LD + c] = v; Ho[c * VB_PRIOR_LD + r] = v; }
                else if (r < SB_ND) v = (c == r) ? 1.0 : 0.0;
                else { v = (c < n) ? va[k] : 0.0; if (c < n) go[c] = v; }       // (75, 75) = 0: the right-hand side row's own diagonal is not a pivot
                s_P[e] = v;
            }
        }
    }
    if (tid == 0) s_ok = 1;
    for (int e = tid; e < 4 * MFT_ROWS; e += NT) { s_pan[e] = 0.0; s_lp[e] = 0.0; }
    __syncthreads();
    MG_STAMP(3, 1);
    // this wave's ten tiles: tile k = wave + 4 i of the list (R, C), R = 0..9, C = 0..min(R, 4)
    int tR[MFT_NT10], tC[MFT_NT10];
#pragma unroll
    for (int i = 0; i < MFT_NT10; i++) {
        const int k = wave + 4 * i;
        int R, C;
        if (k < 15) { R = 0; while ((R + 1) * (R + 2) / 2 <= k) R++; C = k - R * (R + 1) / 2; }
        else { R = 5 + (k - 15) / 5; C = (k - 15) % 5; }
        tR[i] = R; tC[i] = C;
    }
    const int c16 = lane & 15, g4 = lane >> 4;
    typedef double mft_double4 __attribute__((ext_vector_type(4)));
    mft_double4 T[MFT_NT10];
#pragma unroll
    for (int i = 0; i < MFT_NT10; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 16 * tR[i] + g4 + 4 * q, c = 16 * tC[i] + c16;
            double v = 0.0;
            if (r < SB_NR) { if (c <= r) v = s_P[prow(r) + c]; }                    // A, padding, b
            else if (r >= 80 && r - 80 == c && c < SB_ND) v = 1.0;                  // the identity rows
            T[i][q] = v;
        }
    __syncthreads();
    double sq = 0.0;                                                               // this thread's row of L^-T, squared (rows 80 ..)
    MG_STAMP(3, 2);
    MG_ACC_DECL
#pragma unroll 1
    for (int bj = 0; bj < 19; bj++) {
        MG_ACC(0);
        const int j0 = 4 * bj, tc = bj >> 2, sp = bj & 3;
        // 1. the panel's four columns, rows >= j0, to LDS
        if ((c16 >> 2) == sp) {
#pragma unroll
            for (int i = 0; i < MFT_NT10; i++)
                if (tC[i] == tc) {
#pragma unroll
                    for (int q = 0; q < 4; q++) { const int r = 16 * tR[i] + g4 + 4 * q; if (r >= j0) s_pan[4 * r + (c16 & 3)] = T[i][q]; }
                }
        }
        MG_ACC(1);
        __syncthreads();
        MG_ACC(2);
        // 2. one thread per row: Cholesky of the diagonal block, the row's strip of the factor (k_solve_sb's step, rows 76 .. 159 are right-hand sides / padding)
        if (tid >= j0 && tid < MFT_ROWS) {
            const double *dg = s_pan + 4 * j0, *rp = s_pan + 4 * tid;
            const double d00 = dg[0], d10 = dg[4], d11 = dg[5], d20 = dg[8], d21 = dg[9], d22 = dg[10], d30 = dg[12], d31 = dg[13], d32 = dg[14], d33 = dg[15];
            const double r0v = rp[0], r1v = rp[1], r2v = rp[2], r3v = rp[3];
            const double i0 = rsqrt_h3(d00), l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
            const double t11 = d11 - l10 * l10, i1 = rsqrt_h3(t11), l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
            const double t22 = d22 - l20 * l20 - l21 * l21, i2 = rsqrt_h3(t22), l32 = (d32 - l30 * l20 - l31 * l21) * i2;
            const double t33 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
            const bool last = j0 + 3 >= SB_ND;                   // panel 18: column 75 is the right-hand side row's own diagonal — not a pivot
            const double i3 = last ? 0.0 : rsqrt_h3(t33);
            const double x0 = r0v * i0, x1 = (r1v - x0 * l10) * i1, x2 = (r2v - x0 * l20 - x1 * l21) * i2, x3 = (r3v - x0 * l30 - x1 * l31 - x2 * l32) * i3;
            const int k = tid - j0;                               // rows of the diagonal block keep their lower part only
            if (tid < SB_NR) {
                double *Lr = s_P + prow(tid) + j0;
                Lr[0] = x0;
                if (k >= 1) Lr[1] = x1;
                if (k >= 2) Lr[2] = x2;
                if (k >= 3 && j0 + 3 < SB_ND) Lr[3] = x3;
            }
            double *lp = s_lp + 4 * tid;
            lp[0] = x0; lp[1] = (k >= 1) ? x1 : 0.0; lp[2] = (k >= 2) ? x2 : 0.0; lp[3] = (k >= 3) ? x3 : 0.0;
            if (tid >= 80) sq += x0 * x0 + x1 * x1 + x2 * x2 + x3 * x3;                       // (x3 = 0 in the last panel)
            if (k == 0 && (!(d00 > 0.0) || !(t11 > 0.0) || !(t22 > 0.0) || (!last && !(t33 > 0.0)))) s_ok = 0;
        }
        MG_ACC(3);
        __syncthreads();
        MG_ACC(4);
        // 3. rank-4 update of the tiles that reach beyond the panel: one MFMA each
#pragma unroll
        for (int i = 0; i < MFT_NT10; i++)
            if (16 * tC[i] + 15 >= j0 + 4) {
                const double av = -s_lp[4 * (16 * tR[i] + c16) + g4], bv = s_lp[4 * min(16 * tC[i] + c16, 79) + g4];
                T[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, T[i], 0, 0, 0);
            }
    }
    MG_ACC_OUT(3);
    __syncthreads();
    MG_STAMP(3, 3);
    if (!s_ok) { if (tid == 0) qi[3] = 0; return; }
    // trace(A^-1) = |L^-1|_F^2 (the padding's identity rows contribute 75 - n ones: taken off)
    sq = mg_wave_sum(sq);
    if (lane == 0) s_red[wave] = sq;
    __syncthreads();
    double trc = 0.0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) trc += s_red[k];
    trc -= (double)(SB_ND - n);
    if (!(trc < 1e8)) { if (tid == 0) qi[3] = 0; return; }
    // the new prior: linearized_jacobians = L^T (leading dimension n), linearized_residuals = L^-1 b = row 75 of the factor
    double *Jo = g.prior_J_out + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *ro = g.prior_r_out + (size_t)w * VB_PRIOR_LD;
    // (the entries above the diagonal read a zero that lives in LDS: a select on the loaded value is compiled into a branch around the read, a round trip per entry)
    if (tid == 0) s_pan[0] = 0.0;
    __syncthreads();
    for (int e0 = 0; e0 < n * n; e0 += 8 * NT) {
        double jv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = min(e0 + NT * u + tid, n * n - 1), i = e / n, k = e - n * i; jv[u] = *((k >= i) ? s_P + prow(k) + i : s_pan); }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = e0 + NT * u + tid; if (e < n * n) Jo[e] = jv[u]; }
    }
    if (tid < n) ro[tid] = s_P[prow(SB_ND) + tid];
    // (H0 = J0^T J0 = L L^T = A and g0 = J0^T r0 = b went out with the fill at the top: k_prior_prep skips this window)
    MG_STAMP(3, 4);
    mf_table(b, g, w, n, nb, info);
    MG_STAMP(3, 5);
    if (tid == 0) qi[3] = 1;
}

// ---- the eigen-solver split in three launches --------------------------------------------------------------------------------------
// tql2's rotation recurrence is one lane's serial chain: inside k_marg_finish the other 255 lanes of the window's workgroup wait for it
// (~70 % of that kernel). The recurrence touches only d / e — not V — so it runs here for EVERY window at once, one lane per window
// (k_mf_ql), logging the (c, s) rotations; k_mf_tridiag before it does tred2 per workgroup, k_mf_apply after it replays the log on the
// rows of V (no dependent chain left) and writes the prior. Same arithmetic, same order, per window. A window whose log does not fit
// is flagged and taken by k_marg_finish.
extern "C" __global__ __launch_bounds__(NT) void k_mf_tridiag(VbBatch b, VbMarg g, int n_lo, int n_hi) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int *info = g.info + (size_t)w * MG_INFO;
    if (info[0] != 0 || info[3] < n_lo || info[3] >= n_hi || g.qlInfo[(size_t)w * 4 + 3]) return;
    extern __shared__ double s_dyn[];
    __shared__ double s_lam[MG_NK + 2], s_e[MG_NK + 2], s_sc[4];
    const int n = info[3], N = n | 1;
    double *V = s_dyn;
    const double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK;
    for (int e = tid; e < n * n; e += NT) { const int i = e / n, j = e - n * i; V[i * N + j] = 0.5 * (Ar[i * MG_NK + j] + Ar[j * MG_NK + i]); }
    __syncthreads();
    tred2_part(V, n, N, s_lam, s_e, s_sc);
    double *Vg = g.qlV + (size_t)w * MG_NK * (MG_NK + 1), *dg = g.qlD + (size_t)w * 2 * (MG_NK + 2);
    for (int e = tid; e < n * N; e += NT) Vg[e] = V[e];
    if (tid < n) { dg[tid] = s_lam[tid]; dg[MG_NK + 2 + tid] = s_e[tid]; }
}
// one lane per window; d / e live in LDS as [i][lane]
extern "C" __global__ __launch_bounds__(64) void k_mf_ql(VbBatch b, VbMarg g, int force_overflow) {
    extern __shared__ double s_de[];
    // QL_LPW windows per wave: the recurrence is one serial chain per window and lanes in different sweeps diverge (the wave pays the longest
    // sweep of its lanes every time), so fewer windows per wave on more CUs is faster than full waves on a quarter of the chip
    const int lane = threadIdx.x, w = blockIdx.x * QL_LPW + lane;
    if (lane >= QL_LPW || w >= b.B) return;
    const int *info = g.info + (size_t)w * MG_INFO;
    int *qi = g.qlInfo + (size_t)w * 4;
    qi[0] = 0; qi[1] = 0; qi[2] = 0;
    if (info[0] != 0 || qi[3]) return;
    const int n = info[3];
    double *d = s_de + lane, *e = s_de + (MG_NK + 2) * QL_LPW + lane;      // element i at [QL_LPW * i]: 6 KB of LDS per wave, several waves per CU
    double *dg = g.qlD + (size_t)w * 2 * (MG_NK + 2);
    for (int i = 0; i < n; i++) { d[QL_LPW * i] = dg[i]; e[QL_LPW * i] = dg[MG_NK + 2 + i]; }
    double *lg = g.qlLog + (size_t)w * 2 * QL_RCAP;
    int *itab = g.qlIt + (size_t)w * QL_ICAP;
    int ni = 0, nr = 0;
    bool over = force_overflow != 0;              // test hook: send every window through the single-workgroup fallback
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n && !over; l++) {
        tst1 = fmax(tst1, fabs(d[QL_LPW * l]) + fabs(e[QL_LPW * l]));
        int m = l;
        while (m < n) {                              // first negligible sub-diagonal element at or after l, eight candidates per LDS round trip
            double t8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) t8[u] = e[QL_LPW * min(m + u, n - 1)];
            int first = 8;
#pragma unroll
            for (int u = 7; u >= 0; u--) if (m + u < n && fabs(t8[u]) <= eps * tst1) first = u;
            if (first < 8) { m += first; break; }
            m += 8;
        }
        m = min(m, n);
        if (m > l) {
            int iter = 0;
            for (;;) {
                if (ni >= QL_ICAP || nr + (m - l) > QL_RCAP) { over = true; break; }
                iter++;
                double gq = d[QL_LPW * l];
                double p = (d[QL_LPW * (l + 1)] - gq) / (2.0 * e[QL_LPW * l]);
                double r = hypot(p, 1.0);
                if (p < 0) r = -r;
                d[QL_LPW * l] = e[QL_LPW * l] / (p + r);
                d[QL_LPW * (l + 1)] = e[QL_LPW * l] * (p + r);
                const double dl1 = d[QL_LPW * (l + 1)];
                double hq = gq - d[QL_LPW * l];
                {   // the shift of the remaining diagonal, eight loads in flight (one dependent LDS round trip per element cost a third of the kernel)
                    int i = l + 2;
                    for (; i + 7 < n; i += 8) {
                        double t8[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) t8[u] = d[QL_LPW * (i + u)];
#pragma unroll
                        for (int u = 0; u < 8; u++) d[QL_LPW * (i + u)] = t8[u] - hq;
                    }
                    for (; i < n; i++) d[QL_LPW * i] -= hq;
                }
                f += hq;
                p = d[QL_LPW * m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[QL_LPW * (l + 1)];
                double ei = e[QL_LPW * (m - 1)], di = d[QL_LPW * (m - 1)];           // operands of the next step are fetched one step ahead: the LDS
                for (int i = m - 1; i >= l; i--) {                            // latency stays off the dependent chain p -> rr -> inv -> c -> p
                    const int ip = max(i - 1, l);
                    const double ei_n = e[QL_LPW * ip], di_n = d[QL_LPW * ip];
                    c3 = c2; c2 = c; s2 = s;
                    gq = c * ei;
                    hq = c * p;
                    const double rr = p * p + ei * ei;
                    const double inv = rr > 0.0 ? rsqrt_h3(rr) : 0.0;
                    r = rr * inv;
                    e[QL_LPW * (i + 1)] = s * r;
                    s = ei * inv;
                    c = p * inv;
                    p = c * di - s * gq;
                    d[QL_LPW * (i + 1)] = hq + s * (c * gq + s * di);
                    *reinterpret_cast<double2 *>(lg + 2 * nr) = make_double2(c, s); nr++;   // rotation of columns (i, i + 1), logged in application order
                    ei = ei_n; di = di_n;
                }
                p = -s * s2 * c3 * el1 * e[QL_LPW * l] / dl1;
                e[QL_LPW * l] = s * p;
                d[QL_LPW * l] = c * p;
                itab[ni++] = l | (m << 8);
                if (!(fabs(e[QL_LPW * l]) > eps * tst1 && iter < 64)) break;
            }
        }
        if (!over) { d[QL_LPW * l] += f; e[QL_LPW * l] = 0.0; }
    }
    for (int i = 0; i < n; i++) dg[i] = d[QL_LPW * i];                          // eigenvalues
    qi[0] = ni; qi[1] = nr; qi[2] = over ? 1 : 0;
}
#define MFA_CH 64            // rotations per staged chunk of k_mf_apply (two buffers of 1 KB)
extern "C" __global__ __launch_bounds__(NT) void k_mf_apply(VbBatch b, VbMarg g, int n_lo, int n_hi) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int *info = g.info + (size_t)w * MG_INFO;
    if (info[0] != 0 || info[3] < n_lo || info[3] >= n_hi) return;
    const int *qi = g.qlInfo + (size_t)w * 4;
    if (qi[2] || qi[3]) return;                                             // log overflow: k_marg_finish redoes this window; [3]: k_mf_chol has written this prior
    extern __shared__ double s_dyn[];
    __shared__ double s_lam[MG_NK + 2], s_br[MG_NK + 2];
    __shared__ int s_rank[MG_NK + 2];
    const int n = info[3], nb = info[5], N = n | 1;
    double *V = s_dyn;
    const double *Vg = g.qlV + (size_t)w * MG_NK * (MG_NK + 1), *dg = g.qlD + (size_t)w * 2 * (MG_NK + 2), *br = g.br + (size_t)w * MG_NK;
    for (int e = tid; e < n * N; e += NT) V[e] = Vg[e];
    if (tid < n) { s_lam[tid] = dg[tid]; s_br[tid] = br[tid]; }
    __syncthreads();
    // Row k goes through every logged rotation; rows are independent, one lane each (waves 0-1). The rotation log streams through LDS in
    // chunks of MFA_CH rotations, double-buffered: waves 2-3 fetch chunk c + 1 (one double per lane) while the row waves apply chunk c — the
    // rows never wait on a global / scalar load (a 64-byte s_load per four rotations cost ~200 cycles per rotation before; holding the chunk
    // in registers and broadcasting with v_readlane was measured slower than the LDS broadcast reads). The iteration table
    // (l | m << 8 per QL sweep) sits in LDS as 16-bit entries.
    {
        double *s_logd = V + n * N;                                         // [2][2 * MFA_CH] (c, s) pairs; 8-byte reads (16-byte broadcast reads measured slower)
        unsigned short *s_it = reinterpret_cast<unsigned short *>(s_logd + 4 * MFA_CH);   // [QL_ICAP]
        const double *lg = g.qlLog + (size_t)w * 2 * QL_RCAP;
        const int *itab = g.qlIt + (size_t)w * QL_ICAP;
        const int ni = qi[0], R = qi[1], nch = (R + MFA_CH - 1) / MFA_CH;
        for (int e = tid; e < ni; e += NT) s_it[e] = (unsigned short)itab[e];
        if (tid >= 128 && nch > 0) s_logd[tid - 128] = lg[min(tid - 128, 2 * R - 1)];
        __syncthreads();
        if (tid >= 128) {                                                   // loader waves
            for (int c = 0; c < nch; c++) {
                if (c + 1 < nch) s_logd[((c + 1) & 1) * 2 * MFA_CH + tid - 128] = lg[min((c + 1) * 2 * MFA_CH + tid - 128, 2 * R - 1)];
                __syncthreads();
            }
        } else {                                                            // row waves: control flow is uniform, only the LDS accesses are predicated
            const int k = tid, ld = N;
            const bool act = k < n;
            int it = 0, i = -1, l = 0, p = 0;
            double hq = 0.0;
            for (int c = 0; c < nch; c++) {
                const double *cs = s_logd + (c & 1) * 2 * MFA_CH - 2 * c * MFA_CH;      // cs[2 p], cs[2 p + 1] = (c, s) of rotation p
                const int pend = min(R, (c + 1) * MFA_CH);
                while (p < pend && (i >= l || it < ni)) {
                    if (i < l) { const int lm = s_it[it++]; l = lm & 255; const int m = lm >> 8; if (act) hq = VV(k, m); i = m - 1; }
                    int cnt = min(i - l + 1, pend - p);
                    if (act) {
                        for (; cnt >= 4; cnt -= 4, i -= 4, p += 4) {        // four rotations per trip: their operands are independent loads, only hq chains
                            double cc[4], sn[4], vi[4];
#pragma unroll
                            for (int u = 0; u < 4; u++) { cc[u] = cs[2 * (p + u)]; sn[u] = cs[2 * (p + u) + 1]; vi[u] = VV(k, i - u); }
#pragma unroll
                            for (int u = 0; u < 4; u++) { VV(k, i - u + 1) = sn[u] * vi[u] + cc[u] * hq; hq = cc[u] * vi[u] - sn[u] * hq; }
                        }
                        for (; cnt > 0; cnt--, i--, p++) {
                            const double c1 = cs[2 * p], s1 = cs[2 * p + 1], vi = VV(k, i);
                            VV(k, i + 1) = s1 * vi + c1 * hq;
                            hq = c1 * vi - s1 * hq;
                        }
                        if (i < l) VV(k, l) = hq;
                    } else { i -= cnt; p += cnt; }
                }
                __syncthreads();
            }
        }
    }
    __syncthreads();
    mf_tail(b, g, w, V, N, n, nb, info, s_lam, s_br, s_rank);
}

extern "C" __global__ __launch_bounds__(NT) void k_marg_finish(VbBatch b, VbMarg g, int n_lo, int n_hi, int only_flagged) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int *info = g.info + (size_t)w * MG_INFO;
    if (info[0] != 0 || info[3] < n_lo || info[3] >= n_hi) return;
    if (g.qlInfo[(size_t)w * 4 + 3] || (only_flagged && !g.qlInfo[(size_t)w * 4 + 2])) return;
    extern __shared__ double s_dyn[];
    __shared__ double s_cs[2 * (MG_NK + 2)], s_lam[MG_NK + 2], s_e[MG_NK + 2], s_br[MG_NK + 2], s_sc[4];
    __shared__ int s_rank[MG_NK + 2], s_ctl[2];
    const int n = info[3], nb = info[5], N = n | 1;          // odd leading dimension: row / column walks stay off the same LDS banks
    double *V = s_dyn;
    const double *Ar = g.Ar + (size_t)w * MG_NK * MG_NK, *br = g.br + (size_t)w * MG_NK;
    for (int e = tid; e < n * n; e += NT) {
        const int i = e / n, j = e - n * i;
        V[i * N + j] = 0.5 * (Ar[i * MG_NK + j] + Ar[j * MG_NK + i]);
    }
    if (tid < n) s_br[tid] = br[tid];
    __syncthreads();
    tred2_tql2(V, n, N, s_lam, s_e, s_cs, s_sc, s_ctl);     // eigenvalues -> s_lam, eigenvectors -> columns of V
    __syncthreads();
    mf_tail(b, g, w, V, N, n, nb, info, s_lam, s_br, s_rank);
}

#undef VV

// The new prior is written to a second set of buffers (the set that was read stays intact: a rewind to it is a pointer swap, no copy). A window whose prior this call
// leaves as it is (SECOND_NEW without Pose[WINDOW_SIZE - 1] in it, estimator.cpp:982-983, or an unsupported block table) carries it over here.
extern "C" __global__ __launch_bounds__(NT) void k_prior_keep(VbBatch b, VbMarg g) {
    const int w = blockIdx.x, tid = threadIdx.x;
    if (g.info[(size_t)w * MG_INFO] == 0) return;
    for (int i = tid; i < VB_PRIOR_HDR; i += NT) g.prior_hdr_out[(size_t)w * VB_PRIOR_HDR + i] = b.prior_hdr[(size_t)w * VB_PRIOR_HDR + i];
    for (int i = tid; i < 24 * 9; i += NT) g.prior_x0_out[(size_t)w * 24 * 9 + i] = b.prior_x0[(size_t)w * 24 * 9 + i];
    for (int i = tid; i < VB_PRIOR_LD; i += NT) g.prior_r_out[(size_t)w * VB_PRIOR_LD + i] = b.prior_r[(size_t)w * VB_PRIOR_LD + i];
    for (int i = tid; i < VB_PRIOR_LD * VB_PRIOR_LD; i += NT) g.prior_J_out[(size_t)w * VB_PRIOR_LD * VB_PRIOR_LD + i] = b.prior_J[(size_t)w * VB_PRIOR_LD * VB_PRIOR_LD + i];
}


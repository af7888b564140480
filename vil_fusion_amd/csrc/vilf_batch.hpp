// vilf_batch.hpp — device-resident layout of a batch of independent sliding windows (one workgroup per window).
//
// Memory layout in HBM (all per-window arrays are [B][stride] with strides fixed per upload):
//   state      pose[B][11*7]  sb[B][11*9]  feat[B][Fmax]          (+ *_init copies for rewind, cand_* trial point)
//   features   f_start/f_nobs/f_obs0/f_fac0/f_const [B][Fmax]     (CSR over observations, feature_manager order)
//   obs        obs[B][Omax][3]                                     (feature_per_frame[k].point)
//   factors    facrec / ps_obs / ps_slot [B][FACmax]               (one slot per (feature, later observation); the factors of a frame pair (i<j) are contiguous inside
//                                                                   the list of the pair's wave class, the four class lists interleaved 56 slots at a time (VB_SLOT);
//                                                                   unused slots carry a null record; ps_slot = index in feature-major order)
//   pairs      pair_off[B][VB_PTAB]                                (per pair: start inside its class list, factor count, class)
//   imu        imu[B][10][288]  (delta_*, bias jacobians, 15x15 sqrt_info precomputed once: imu_factor.h:64)
//   lidar      lidar[B][10][7]
//   prior      hdr[B][80] x0[B][24][9] J[B][160*160] r[B][160] H=J^T J [B][160*160] g=J^T r [B][160]
//   workspace  facw[B][8][FACmax] (per factor: Ji^T Jf (6), Jf^T Jf, Jf^T r)  Hpp[B][66][36] (visual pose-pose blocks)
//              W[B][Fmax][80] (H_pf rows zero-initialised at upload; the feature's frame range + column 66 = g_f are
//              rewritten every linearization) hf gf [B][Fmax]
//              imuH[B][10][900] imug[B][10][30] lidH[B][10][144] lidg[B][10][12] g[B][165]
//              scale/diag/grad/gn[B][165+Fmax] st[B]
#pragma once
#include <stdint.h>

#define VB_NF 11            // frames in the window (WINDOW_SIZE + 1)
#define VB_P 165            // reduced tangent dimension: 11 * (6 + 9)
#define VB_NPOSE 66
#define VB_NPAIR 55         // frame pairs i<j
#define VB_PAIRD 120        // per pair: JjJj(36) JjJi(36) JiJi(36) Jj^T r(6) Ji^T r(6)
#define VB_FACW 8           // per-factor Schur partials
#define VB_WLD 80           // row stride of W: 66 pose columns, column 66 = g_f, 67..79 zero (5 MFMA column tiles)
#define VB_XLD 13           // LDS row stride (doubles) of the factor chunk [Jj(6) Ji(6) r]; the MFMA operand load masks columns 13..15
#define VB_CLS 56           // factor slots per wave class in a chunk: the 55 frame pairs are dealt to the 4 waves of k_linearize by factor count (LPT), and every chunk
                            // holds 56 slots of each class — each wave finds an equal share of MFMA rows in every chunk instead of one wave owning the chunk's big pair
#define VB_CHUNK (4 * VB_CLS) // factor slots per LDS chunk (one per thread, threads 224..255 sit the evaluation out): 448 rows x 13 doubles = 46.6 KB
                            // -> k_linearize needs < 80 KB of LDS and two workgroups share a CU
#define VB_LIN_LDS_DOUBLES (2 * VB_CHUNK * VB_XLD + 8)   // >= 10 * 512 (the IMU staging area that precedes the chunk loop)
#define VB_LIN_SHARED_BYTES 33024                        // the tables behind the chunk (struct LinShared, vilf_kernels.hip): all of k_linearize's LDS is dynamic
#define VB_LIN_LDS_BYTES (VB_LIN_LDS_DOUBLES * 8 + VB_LIN_SHARED_BYTES)
#define VB_PTAB (2 * VB_NPAIR + 2)   // per-window pair table: [2 p] = start of pair p inside its class list, [2 p + 1] = factor count | class << 24
#define VB_SLOT(cls, x) (VB_CHUNK * ((x) / VB_CLS) + VB_CLS * (cls) + ((x) % VB_CLS))     // class-local position -> slot of the chunk-interleaved factor arrays
#define VB_NT 256           // threads per window workgroup
#define VB_NTILE 11         // 16x16 tiles per dimension (176 padded)
#define VB_NPAD 176
#define VB_PRIOR_HDR 80     // valid, n, nblocks, ids[24], sizes[24], idx[24]
#define VB_PRIOR_LD 160

// ---- k_solve_sb: speed-bias-first elimination (vilf_kernels.hip) -------------------------------------------------------
// After the feature Schur complement the speed-bias part of the reduced system is block tridiagonal (an IMU factor couples consecutive frames only,
// estimator.cpp:742-749; the prior holds SpeedBias[0] alone). SpeedBias[1..10] ("chain" blocks) are eliminated first, newest frame first; the poses and
// SpeedBias[0] (which the prior couples to every pose) form the dense block: 66 + 9 = 75 variables, row 75 = right-hand side, row 76 = Cauchy-point row.
#define SB_ND 75            // dense variables (permuted indices 0..74: poses, SpeedBias[0])
#define SB_NR 76            // rows of the packed lower-triangular dense system (SB_ND + the right-hand side row)
#define SB_NCH 10           // chain blocks: SpeedBias[1..10]
// band_a = H~(SpeedBias[a], dense): [Pose a-1 | Pose a | Pose a+1 | rhs] (stride 20, rhs at 18); a = 1 also holds SpeedBias[0]: [.. | SpeedBias[0] | rhs] (stride 28, rhs at 27)
#define SB_BSTR(a) ((a) == 1 ? 28 : 20)
#define SB_BRHS(a) ((a) == 1 ? 27 : 18)
#define SB_BOFF(a) ((a) == 1 ? 0 : 252 + 180 * ((a) - 2))
#define SB_OFF_P 0
#define SB_OFF_D (SB_OFF_P + SB_NR * (SB_NR + 1) / 2)            // 2926: D_a -> L_a -> M_a = L_a^-1 (lower, 9 x 9 each)
#define SB_OFF_E (SB_OFF_D + SB_NCH * 81)                         // 3736: E_a -> B_(a+1) -> N_a
#define SB_OFF_BAND (SB_OFF_E + (SB_NCH - 1) * 81)                // 4465: band_a -> M_a band_a
#define SB_OFF_VEC (SB_OFF_BAND + 252 + 180 * (SB_NCH - 1))       // 6337: g~, diagonal_, scale, y, v (168 each)
#define SB_VLD 168
#define SB_OFF_T (SB_OFF_VEC + 5 * SB_VLD)                        // Cauchy-point row (80)
#define SB_OFF_LINV (SB_OFF_T + 80)                               // 1 / L_ii of the chain blocks (96)
#define SB_OFF_DINV (SB_OFF_LINV + 96)                            // inverses of the 4 x 4 diagonal blocks of the dense factor (19 x 16)
#define SB_OFF_PAN (SB_OFF_DINV + 19 * 16)                        // dense factorisation: current 4-column panel [76][4], then the factor rows of the panel as MFMA operands [80][4]
#define SB_OFF_U (SB_OFF_PAN + 2 * 304 + 16)                      // chain: forward solution (96)
#define SB_OFF_W (SB_OFF_U + 96)                                  // chain: M r -> backward vectors (96)
#define SB_OFF_RED (SB_OFF_W + 96)                                // block-sum scratch (16)
#define SB_LDS_DOUBLES (SB_OFF_RED + 16)

struct VbState {            // per-window trust-region state (ceres TrustRegionMinimizer + DoglegStrategy members)
    double x_cost, cand_cost, initial_cost;
    double radius, mu, alpha, dogleg_step_norm;
    double x_norm, gradient_max_norm;
    double grad_sqnorm;     // ||gradient_||^2            (scaled space)
    double Jg2;             // v^T H v, v = gradient_/diagonal_
    double gy;              // g~^T y  (y = (H~ + mu D^2)^-1 g~)
    double gn_sqnorm;       // ||gauss_newton_step_||^2 = sum diag^2 y^2
    double mu_used;         // mu the current Gauss-Newton step was computed with
    double model_cost_change, relative_decrease;
    int iteration, num_successful, num_linear_solves, num_consecutive_invalid;
    int termination, done, reuse, need_linearize, solve_failed, scaling_ready, started;
    int dev_error;          // set by a kernel that gave up waiting for another workgroup (k_linearize_split's bounded carry wait): every reader of the results returns
                            // VILF_ERR_DEVICE for this window instead of numbers computed from a carry that never came
    int ws;                 // which of the two linearisation workspaces (W, Hpp, imuH, pairD, g ...) belongs to the current state x: k_linearize writes the OTHER one at the
                            // candidate and flips this when the step is accepted
};

// ---- marginalization workspace (vilf_marg.hip) ---------------------------------------------------------------------
#define MG_RWP 113          // staged row length: 7 tiles of 16 columns (+ 1: odd stride, no bank conflicts between the four k of an operand read)
#define MG_FCH 16           // arrow rows staged per chunk in the fast path of k_marg_schur (16 x 118 doubles of LDS)
#define MG_MD 21            // dropped non-feature variables: Pose[0] 6 + Pose[1] 6 (USE_LIDAR_CONST, estimator.cpp:891) + SpeedBias[0] 9
#define MG_NK 96            // kept (prior) dimension capacity; the reference's prior never exceeds 75 + td
#define MG_ND (MG_MD + MG_NK)
#define MG_MROW 42          // per visual factor: J_P0[12] J_Pj[12] J_Ex[12] J_f[2] r[2] J_td[2] (the last two only with estimate_td)
#define MG_PAIRM 400        // per pair (0,j): [J0 Jj Jex Jtd r]^T [..] upper triangle (20 columns: td = 18, r = 19), stored 20x20
#define QL_RCAP 12288        // rotations logged per window by k_mf_ql (typical: ~5 k for n = 75)
#define QL_ICAP 768          // QL iterations logged per window
#define QL_LPW 16           // windows per wave of k_mf_ql (8: faster at 2048 windows, slower at 4096)
#define MG_SLOTS 1024       // start-frame-0 factor rows whose Mbuf index is kept in LDS by k_marg_prepare (beyond: looked up again)
#define MG_GCH 80           // factor rows staged per chunk in the pair gather of k_marg_prepare (27 KB of LDS; 96 until the per-wave partial products needed 6.7 KB)
#define MG_PTRI 210         // entries of the packed lower triangle of a 20 x 20 pair product
#define MG_MLDS 136         // largest Amm held in LDS by the Jacobi eigen-solver
#define MG_INFO 128         // per window: [0] status [1] md [2] mf [3] n [4] m [5] nblocks [6] M(padded) [8..31] shifted ids [32..55] sizes
                            //             [56..79] idx [80..103] original ids
#define MG_SWEEPS 24
struct VbMarg {
    int Mcap;               // capacity (padded, even) of m = md + mf over the batch
    double init_depth;
    const int *mflag;       // [B] marginalization_flag
    int *info;              // [B][MG_INFO]
    int *f0rank;            // [B][Fmax] rank among the start-frame-0 features, or -1
    double *st_pose, *st_sb, *st_feat, *st_ex;   // linearization point = vector2double() of the post-gauge state (para_Td: VbBatch::td)
    double *Mbuf;           // [B][FACmax][MG_MROW]   (one 40-double record per visual factor, feature-major slot order)
    double *Hd, *gd;        // [B][MG_ND*MG_ND], [B][MG_ND]    dense-variable normal equations
    double *Wf;             // [B][Fmax][MG_ND]  arrow rows of the start-0 features (indexed by rank)
    double *hfm, *gfm;      // [B][Fmax]
    double *Amm;            // [pool][Mcap*Mcap]    only used when m > MG_MLDS
    double *X;              // [B][Mcap][MG_NK+1]
    double *rot;            // [pool][MG_SWEEPS][Mcap-1][Mcap]   (c,s) log of the Jacobi rotations
    int pool, pool_round;   // Amm / X / rot / lam are a pool of `pool` slots for the windows of the exact (Jacobi) path: launch `pool_round` takes the flagged windows of rank
                            // [pool_round * pool, (pool_round + 1) * pool) (a slot per window would be 20 MB per window at 300 dropped features)
    double *lam;            // [B][Mcap]
    double *Ar, *br;        // [B][MG_NK*MG_NK], [B][MG_NK]
    double *qlV, *qlD, *qlLog;   // eigen-solver split: [B][MG_NK*(MG_NK+1)] tridiagonalising transform, [B][2*(MG_NK+2)] d / e, [B][2*QL_RCAP] (c, s) rotations
    int *qlIt, *qlInfo;     // [B][QL_ICAP] l | m << 8 per QL iteration, [B][4] iterations, rotations, overflow
    int *prior_hdr_out;     // = batch prior arrays (written by k_marg_finish)
    double *prior_x0_out, *prior_J_out, *prior_r_out;
    double *prior_H_out, *prior_g_out;   // H0 = J0^T J0, g0 = J0^T r0 of the new prior: k_mf_chol writes them itself (they ARE A and b there); k_prior_prep fills the rest
};

#define VB_SPLIT_MAXCH 12        // factor chunks a split window may have (12 x 224 slots)
#define VB_SPLIT_MAXB 8          // largest batch that is split. Measured (tools/dev_small_batch_sweep.sh, ms per 8-iteration solve of the batch, split / one workgroup per window):
                                 // B = 1: 1.11 / 1.33, 4: 1.16 / 1.37, 8: 1.27 / 1.37, 16: 1.51 / 1.41, 32: 2.09 / 1.46 — beyond ~100 workgroups the roles get in each other's way
#define VB_SPLIT_CTL 64
#define VB_SPLIT_CNT 52          // split_ctl[52 + (generation & 7)]: the arrival counter of the launch with that generation; [1 + 4 chunk + wave]: "carry written" = the generation
#define VB_SPLIT_SPIN_MAX (1 << 20)   // bounded wait for a carry: ~2^20 x (s_sleep 1 + an L2 load) ~ 0.1 s, four orders of magnitude beyond the longest chunk; then dev_error
#define VB_SPLIT_CARRY (VB_SPLIT_MAXCH * 4 * 64 * 8)
#define VB_SPLIT_DBL (VB_SPLIT_CARRY + (VB_SPLIT_MAXCH + 3) * 256)
struct VbBatch {
    int B, Fmax, Omax, FACmax;
    int w0;                 // first window of this launch (a batch may be enqueued in parts: window = blockIdx.x + w0)
    // Live-window lists (converging batches): a window that has stopped leaves every later launch; with the windows addressed through a dense list of the ones still
    // running, the finished ones' workgroups sit at the END of the grid and exit at once wherever they were in the batch (without it the saving depended on where the
    // short workgroups fell among the long ones: 0 .. 40 %). Launches of iteration i (k_solve_sb, k_linearize; live_it = i): live_ctl[i] = 1 once any window of the
    // batch has stopped by the end of iteration i; k_linearize of iteration i appends the windows it leaves unfinished to list i (arrival order: which workgroup takes
    // which window is free, a window's arithmetic is untouched) only when live_ctl[i - 1] is set, and iteration i + 1 addresses its windows through list i under the
    // same condition — a batch in which nothing has stopped pays two scalar loads per workgroup and no atomic. live_ctl: [64] flags, [64] list lengths; live_buf: 2 x B.
    int *live_ctl, *live_buf; int live_it;
    // Small batches (the real-time case is ONE window): k_linearize_split gives a window split_nr workgroups — one per chunk of VB_CHUNK factor slots, one for the IMU
    // factors, one for the LiDAR factors and the prior's cost — and the one that arrives last assembles (vilf_kernels.hip). split_ctl: per window VB_SPLIT_CTL ints
    // ([0] arrivals, [1 + 4 chunk + wave] "carry written"); split_buf: per window VB_SPLIT_DBL doubles (carries [chunk][wave][lane][8], cost partials [3 + chunks][256]).
    int split_nr; int *split_ctl; double *split_buf;
    int split_fault;        // test hook (VILF_SPLIT_FAULT=1): chunk 0 never publishes its carries — the consumers' bounded wait must end the launch with dev_error
    int split_gen;          // generation of this launch (host counter, never 0): flags carry it, so nothing a dead or earlier launch left behind can be mistaken for a hand-over
    // options
    double sqrt_info, cauchy_b, G[3];
    double qil[4], til[3];  // RIC*RCL as quaternion (xyzw), RIC*TCL+TIC (lidar_factor.h:28-29)
    int use_lidar, max_iterations;
    double min_relative_decrease, function_tolerance, gradient_tolerance, parameter_tolerance;
    double min_radius, initial_radius, min_lm_diagonal, max_lm_diagonal;
    // sizes
    const int *n_feat, *n_fac;
    // state
    double *pose, *sb, *feat;
    double *cand_pose, *cand_sb, *cand_feat;
    const double *pose_init, *sb_init, *feat_init;
    const double *ex;
    const double *gauge_R0, *gauge_P0;
    // problem description
    const int *f_start, *f_nobs, *f_obs0, *f_fac0;
    const uint8_t *f_const;
    const double *obs;
    const double *obs_vel, *obs_ctd, *obs_row;   // estimate_td only: [B][Omax][2] pixel velocity, [B][Omax] td at capture, [B][Omax] image row (ProjectionTdFactor, projection_td_factor.cpp:6-21)
    double *td;             // [B] para_Td
    int est_td; double tr_over_row, row_half;
    const int *ps_feat, *ps_obs, *ps_slot;
    const int *pair_off;
    const double *facrec;   // [B][FACmax][8]: per pair-sorted factor {pts_i[3], pts_j[3], (feature | slot << 32), (frame_i | frame_j << 8 | const << 16)} — one
                            // coalesced 64-byte record instead of five dependent gathers
    const double *imu, *lidar;
    const int *sb_tab;      // k_solve_sb's gather index table (k_sb_table, built once per handle): SB_TAB_ROWS int4 rows x 256 threads
    const int *lut_imu, *lut_lid, *lut_vis;   // static scatter tables: source element -> LDS tile offset (or -1)
    double *cf;             // [B][Fmax] per-feature Schur coefficient s_f / sqrt(h~_f'), then W_f . (S y)_p (k_solve_sb)
    const int *prior_hdr;
    const double *prior_x0, *prior_J, *prior_r, *prior_H, *prior_g;
    // workspace
    double *facw, *Hpp, *W, *hf, *gf, *imuH, *imug, *lidH, *lidg, *g;
    double *pairD;          // [B][55][120] per-frame-pair visual products (accumulated over the factor chunks; L2-resident)
    double *diagH;          // [B][165] diagonal of the (unscaled) reduced system, frame-major (Jacobi scaling / dogleg diagonal)
    double *scale, *diag, *grad, *gn;
    VbState *st;
    // outputs of finalize
    double *out_Ps, *out_Rs, *out_Vs, *out_Bas, *out_Bgs;
    long long *dbg;         // optional phase stamps (s_memtime) of window 0: [kernel 0..2][32]; NULL in production
};

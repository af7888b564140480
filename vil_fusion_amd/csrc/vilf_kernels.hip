// vilf_kernels.hip — hand-written gfx950 kernels of the sliding-window solve (one 256-thread workgroup per window).
//
//   k_imu_prep     one-off: sqrt_info = LLT(cov^-1).L^T per IMU factor            (imu_factor.h:64)
//   k_prior_prep   one-off: H0 = J0^T J0, g0 = J0^T r0 of the marginalization prior (marginalization_factor.cpp:333-381)
//   k_linearize    every factor's residual + Jacobian at x, robust corrector, per-frame-pair J^T J / J^T r blocks,
//                  per-feature Schur vectors (H_pf, H_ff, g_f), IMU / LiDAR / prior blocks, cost, gradient
//                  (≙ ceres Evaluate inside Solve, estimator.cpp:852; factors: factor/*.h)
//   k_solve        assemble the 165x165 reduced system as 16x16 fp64 tiles in LDS, Jacobi scaling, dogleg diagonal,
//                  Cauchy point, MFMA Schur reduce (-= W~^T W~), MFMA blocked Cholesky, Gauss-Newton step
//                  (≙ DoglegStrategy::ComputeStep + DENSE_SCHUR, Ceres 2.0)
//   (k_step)       dogleg interpolation, Plus(), trial cost, accept / reject, radius update: since round 2 the prologue / epilogue of k_linearize, which linearises at
//                  the candidate into the second workspace and flips VbState::ws on acceptance
//                  (≙ TrustRegionMinimizer loop body)
//   k_finalize     double2vector() gauge fix (estimator.cpp:549-596)
//
// All arithmetic is fp64 (the reference is double throughout). MFMA: v_mfma_f64_16x16x4_f64.
#include "vilf_device.hpp"
#include "vilf_batch.hpp"

using namespace vd;

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

#define NT VB_NT
#define VILF_MAX_FEATURES_DEV 1000
// In-kernel phase stamps are a diagnostic build only (make DEFS=-DVILF_STAMPS): the production kernels carry no clock reads.
#ifdef VILF_STAMPS
#define STAMP(kid, i) do { if (b.dbg && blockIdx.x == 0 && threadIdx.x == 0) b.dbg[(kid) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
#define TICK() __builtin_readcyclecounter()
#else
#define STAMP(kid, i) do { } while (0)
#define TICK() 0LL
#endif
#define LSTAMP(i) do { if (JAC) STAMP(0, i); } while (0)      // the step-only launch shares the body: it must not overwrite the stamps of the linearisation
__device__ __forceinline__ int pair_index(int i, int j) { return j * (j - 1) / 2 + i; }  // i < j
// tangent index (frame a, local l in [0,15)) -> P-first permuted index: poses 0..65, speed-bias 66..164
__device__ __forceinline__ int perm_index(int a, int l) { return l < 6 ? 6 * a + l : 66 + 9 * a + (l - 6); }
__device__ __forceinline__ void unperm(int p, int &a, int &l) {
    if (p < 66) { a = p / 6; l = p - 6 * a; } else { int q = p - 66; a = q / 9; l = 6 + q - 9 * a; }
}
__device__ __forceinline__ int tile_index(int ta, int tb) { return ta * (ta + 1) / 2 + tb; }  // ta >= tb
// XOR swizzle inside a 16x16 fp64 tile: element (r, c) at 16*r + (c ^ r). Row reads stay one contiguous 128-B line, column reads
// (fixed c, 16 rows) hit 16 distinct bank pairs instead of 2 (ds_read_b64: 64 banks x 4 B; unswizzled row stride = 128 B).
#define TIX(r, c) (16 * (r) + ((c) ^ (r)))

// block-wide sum / max with a fixed reduction tree (deterministic)
// block reductions of the 256-thread kernels: the fixed tree of vilf_wave_sum64 inside each wave, then the four wave results in wave order by every thread — three
// barriers (the shared-memory halving tree they replace had ten per call)
__device__ __forceinline__ double block_sum(double v, double *s_red) {
    const int tid = threadIdx.x;
    v = vilf_wave_sum64(v);
    lds_barrier();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    lds_barrier();
    double r = 0;
#pragma unroll
    for (int k = 0; k < NT / 64; k++) r += s_red[k];
    lds_barrier();
    return r;
}
__device__ __forceinline__ double block_max(double v, double *s_red) {
    const int tid = threadIdx.x;
    v = vilf_wave_max64(v);
    lds_barrier();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    lds_barrier();
    double r = s_red[0];
#pragma unroll
    for (int k = 1; k < NT / 64; k++) r = fmax(r, s_red[k]);
    lds_barrier();
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// one-off preparation
// cov: [n][225] in, out: rec + IMU_SQRT of each factor: sqrt_info = LLT(cov^-1).L^T (imu_factor.h:64), once per upload.
// 16 lanes per factor (lane = matrix row, 4 factors per 64-thread block), matrices in LDS; every element sees the same sequence
// of operations as the row-serial algorithm (Gauss-Jordan with partial pivoting = Eigen PartialPivLU inverse, left-looking LLT).
extern "C" __global__ __launch_bounds__(64) void k_imu_prep(int n, const double *cov, double *work, double *imu_rec) {
    __shared__ double sA[4][15][16], sI[4][15][16], s_piv[4];
    const int grp = threadIdx.x >> 4, r = threadIdx.x & 15, id = blockIdx.x * 4 + grp;
    const bool row = id < n && r < 15;
    (void)work;
    if (row) for (int j = 0; j < 15; j++) { sA[grp][r][j] = cov[(size_t)id * 225 + 15 * r + j]; sI[grp][r][j] = (j == r) ? 1.0 : 0.0; }
    __syncthreads();
    for (int k = 0; k < 15; k++) {
        if (id < n && r == 0) {                       // partial pivoting: first row with the largest |A[i][k]|, i >= k
            int p = k; double best = fabs(sA[grp][k][k]);
            for (int i = k + 1; i < 15; i++) { const double v = fabs(sA[grp][i][k]); if (v > best) { best = v; p = i; } }
            if (p != k) for (int j = 0; j < 15; j++) { double t = sA[grp][k][j]; sA[grp][k][j] = sA[grp][p][j]; sA[grp][p][j] = t; t = sI[grp][k][j]; sI[grp][k][j] = sI[grp][p][j]; sI[grp][p][j] = t; }
            s_piv[grp] = sA[grp][k][k];
        }
        __syncthreads();
        if (row && r > k) {
            const double f = sA[grp][r][k] / s_piv[grp];
            for (int j = k; j < 15; j++) sA[grp][r][j] -= f * sA[grp][k][j];
            for (int j = 0; j < 15; j++) sI[grp][r][j] -= f * sI[grp][k][j];
        }
        __syncthreads();
    }
    for (int k = 14; k >= 0; k--) {
        if (row && r == k) { const double piv = sA[grp][k][k]; for (int j = 0; j < 15; j++) sI[grp][k][j] /= piv; }
        __syncthreads();
        if (row && r < k) { const double f = sA[grp][r][k]; for (int j = 0; j < 15; j++) sI[grp][r][j] -= f * sI[grp][k][j]; }
        __syncthreads();
    }
    // lower Cholesky of the inverse (in place), sqrt_info = L^T
    for (int j = 0; j < 15; j++) {
        if (row && r == j) { double sacc = sI[grp][j][j]; for (int k = 0; k < j; k++) sacc -= sI[grp][j][k] * sI[grp][j][k]; sI[grp][j][j] = sqrt(sacc); }
        __syncthreads();
        if (row && r > j) { double t = sI[grp][r][j]; for (int k = 0; k < j; k++) t -= sI[grp][r][k] * sI[grp][j][k]; sI[grp][r][j] = t / sI[grp][j][j]; }
        __syncthreads();
    }
    if (row) { double *S = imu_rec + (size_t)id * IMU_REC + IMU_SQRT; for (int j = 0; j < 15; j++) S[15 * r + j] = (j >= r) ? sI[grp][j][r] : 0.0; }
}

// done4 (nullable): [B][4] flags of the marginalization; [4 w + 3] != 0: k_mf_chol has written this window's H0 / g0 already
extern "C" __global__ __launch_bounds__(NT) void k_prior_prep(VbBatch b, double *prior_H, double *prior_g, unsigned lds_bytes, const int *done4) {
    const int w = blockIdx.x, tid = threadIdx.x;
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    if (!hdr[0] || (done4 && done4[4 * (size_t)w + 3])) return;
    const int n = hdr[1];
    const double *J = b.prior_J + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    const double *r = b.prior_r + (size_t)w * VB_PRIOR_LD;
    double *H = prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    double *g = prior_g + (size_t)w * VB_PRIOR_LD;
    // J (n x n) staged in LDS once: every entry of H = J^T J reads two of its columns (2 n reads per entry from L2 otherwise); same order of additions
    extern __shared__ double s_J[];
    const int ld = n | 1;
    if ((size_t)n * ld * sizeof(double) <= lds_bytes) {
        for (int e = tid; e < n * n; e += NT) { const int k = e / n, i = e - k * n; s_J[k * ld + i] = J[e]; }
        __syncthreads();
        for (int e = tid; e < n * n; e += NT) {
            int i = e / n, j = e - i * n;
            double s = 0;
            for (int k = 0; k < n; k++) s += s_J[k * ld + i] * s_J[k * ld + j];
            H[i * VB_PRIOR_LD + j] = s;
        }
        for (int i = tid; i < n; i += NT) {
            double s = 0;
            for (int k = 0; k < n; k++) s += s_J[k * ld + i] * r[k];
            g[i] = s;
        }
        return;
    }
    for (int e = tid; e < n * n; e += NT) {
        int i = e / n, j = e - i * n;
        double s = 0;
        for (int k = 0; k < n; k++) s += J[k * n + i] * J[k * n + j];
        H[i * VB_PRIOR_LD + j] = s;
    }
    for (int i = tid; i < n; i += NT) {
        double s = 0;
        for (int k = 0; k < n; k++) s += J[k * n + i] * r[k];
        g[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// prior helpers: column map (tangent index frame-major -> prior column) and dx (marginalization_factor.cpp:345-363)
__device__ void prior_setup(const VbBatch &b, int w, const double *pose, const double *sb, int *s_pcol, double *s_dx, int tid) {
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    for (int i = tid; i < VB_P; i += NT) s_pcol[i] = -1;
    for (int i = tid; i < VB_PRIOR_LD; i += NT) s_dx[i] = 0.0;
    __syncthreads();
    if (!hdr[0]) return;
    const int nb = hdr[2];
    if (tid < nb) {
        const int id = hdr[3 + tid], size = hdr[27 + tid], idx = hdr[51 + tid];
        const double *x0 = b.prior_x0 + ((size_t)w * 24 + tid) * 9;
        if (id < VB_NF) {
            const double *x = pose + 7 * id;
            for (int k = 0; k < 3; k++) s_dx[idx + k] = x[k] - x0[k];
            Q dq = q_mul(q_inv(q_load(x0 + 3)), q_load(x + 3));
            double sgn = (dq.w >= 0) ? 2.0 : -2.0;
            s_dx[idx + 3] = sgn * dq.x; s_dx[idx + 4] = sgn * dq.y; s_dx[idx + 5] = sgn * dq.z;
            for (int k = 0; k < 6; k++) s_pcol[15 * id + k] = idx + k;
        } else if (id < 2 * VB_NF) {
            const int a = id - VB_NF;
            const double *x = sb + 9 * a;
            for (int k = 0; k < 9; k++) { s_dx[idx + k] = x[k] - x0[k]; s_pcol[15 * a + 6 + k] = idx + k; }
        } else if (id == 2 * VB_NF) {
            // Ex_Pose: constant in the solve (estimate_extrinsic = 0): contributes dx to the residual only
            const double *x = b.ex + (size_t)w * 7;
            for (int k = 0; k < 3; k++) s_dx[idx + k] = x[k] - x0[k];
            Q dq = q_mul(q_inv(q_load(x0 + 3)), q_load(x + 3));
            double sgn = (dq.w >= 0) ? 2.0 : -2.0;
            s_dx[idx + 3] = sgn * dq.x; s_dx[idx + 4] = sgn * dq.y; s_dx[idx + 5] = sgn * dq.z;
        }
        (void)size;
    }
    __syncthreads();
}

// 0.5 * || r0 + J0 dx ||^2 summed over the block's threads (each thread returns its partial)
__device__ double prior_cost_partial(const VbBatch &b, int w, const double *s_dx, int tid) {
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    if (!hdr[0]) return 0.0;
    const int n = hdr[1];
    const double *J = b.prior_J + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    const double *r0 = b.prior_r + (size_t)w * VB_PRIOR_LD;
    double acc = 0;
    for (int row = tid; row < n; row += NT) {
        double s = r0[row];
        for (int k = 0; k < n; k++) s += J[row * n + k] * s_dx[k];
        acc += 0.5 * s * s;
    }
    return acc;
}

// The prior's two matrix-vector products, J0 dx (cost) and H0 dx (gradient), with every row dealt to THREE threads: a row is a chain of n = 75 loads from lines of its
// own (row-per-lane access: nothing coalesces), eight in flight — ten dependent trips to L2 for a third of the workgroup while the rest waits at the next barrier. A third
// of a row is three trips, and all 256 threads issue. The three partial sums are added in a fixed order (part 0, 1, 2). Both contain LDS-only barriers: all threads.
// 3 n > 256 (a prior wider than 85): the row-per-thread forms.
// (Also built and measured, same box: J0 staged in LDS for both products — 17 k cycles dearer in the cost phase than it saved in the gradient phase; both products on
//  the matrix cores, dx in column 0 of the B operand, H0 read along its rows — 30 k cycles for the pair under load against 36 k here, 11 k against 7 k alone on the
//  chip, 252 registers: the loads' address arithmetic at one fp64-rate VALU instruction per ~12 cycles and wave costs what the coalescing saves.)
__device__ __forceinline__ double prior_cost_split(const VbBatch &b, int w, const double *s_dx, double *s_tmp, int tid) {
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    if (!hdr[0]) return 0.0;
    const int n = hdr[1];
    if (3 * n > NT) return prior_cost_partial(b, w, s_dx, tid);
    const double *J = b.prior_J + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    const int row = tid / 3, part = tid - 3 * row, per = (n + 2) / 3, c0 = part * per, c1 = min(n, c0 + per);
    const double r0v = b.prior_r[(size_t)w * VB_PRIOR_LD + min(row, n - 1)];
    double acc = 0;
    if (row < n) {
        const double *Jr = J + (size_t)row * n;
#pragma unroll 8
        for (int k = c0; k < c1; k++) acc += Jr[k] * s_dx[k];
    }
    s_tmp[tid] = acc;
    lds_barrier();
    double cost = 0;
    if (row < n && part == 0) { double t = r0v; t += s_tmp[tid]; t += s_tmp[tid + 1]; t += s_tmp[tid + 2]; cost = 0.5 * t * t; }
    lds_barrier();
    return cost;
}
__device__ __forceinline__ double prior_grad_split(const VbBatch &b, int w, const double *s_dx, const int *s_pcol, double *s_tmp, int tid) {
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    if (!hdr[0]) return 0.0;
    const int n = hdr[1];
    const int pc = (tid < VB_P) ? s_pcol[tid] : -1;
    if (3 * n > NT) {
        if (pc < 0) return 0.0;
        double t = b.prior_g[(size_t)w * VB_PRIOR_LD + pc];
        const double *pH = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD + (size_t)pc * VB_PRIOR_LD;
#pragma unroll 8
        for (int k = 0; k < n; k++) t += pH[k] * s_dx[k];
        return t;
    }
    const int col = tid / 3, part = tid - 3 * col, per = (n + 2) / 3, c0 = part * per, c1 = min(n, c0 + per);
    const double g0 = b.prior_g[(size_t)w * VB_PRIOR_LD + max(pc, 0)];
    double acc = 0;
    if (col < n) {
        const double *pH = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD + (size_t)col * VB_PRIOR_LD;
#pragma unroll 8
        for (int k = c0; k < c1; k++) acc += pH[k] * s_dx[k];
    }
    s_tmp[tid] = acc;
    lds_barrier();
    double g = 0;
    if (pc >= 0) { g = g0; g += s_tmp[3 * pc]; g += s_tmp[3 * pc + 1]; g += s_tmp[3 * pc + 2]; }
    lds_barrier();
    return g;
}

// The prior through J0 alone, every load coalesced: t = r0 + J0 dx with a WAVE per row (lanes = columns: two loads of consecutive addresses, a fixed-tree wave sum),
// cost 0.5 |t|^2; the gradient as Ceres forms it, J0^T t, with a LANE per column and the rows dealt to three waves (a load = consecutive columns of one row). Against
// the row-per-thread forms (prior_cost_split / prior_grad_split: every lane on a line of its own, J0 AND H0 = 90 KB per window and launch) this is a sixth of the
// requests and half the bytes — under a full device the two products took 36 k of k_linearize's 250 k cycles, 7 k alone on the chip: they wait for the memory system,
// and what loads it is the number of requests. n <= 85 (three partial sums of n columns in 256 doubles); wider priors: the split forms.
// s_t: >= 85 doubles (t, kept for the gradient). Contains LDS-only barriers: all threads.
template <bool GRAD>
__device__ __forceinline__ void prior_products_j(const VbBatch &b, int w, const double *s_dx, const int *s_pcol, double *s_tmp, double *s_t, int tid, double &cost, double &gprior) {
    cost = 0.0; gprior = 0.0;
    const int *hdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    if (!hdr[0]) return;
    const int n = hdr[1];
    if (n > 85) {
        cost = prior_cost_split(b, w, s_dx, s_tmp, tid);
        if (GRAD) gprior = prior_grad_split(b, w, s_dx, s_pcol, s_tmp, tid);
        return;
    }
    const double *J = b.prior_J + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, *r0 = b.prior_r + (size_t)w * VB_PRIOR_LD;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = min(lane, n - 1), c1 = min(lane + 64, n - 1);
    const double dx0 = s_dx[lane], dx1 = s_dx[lane + 64];               // zero from n on (prior_setup): a clamped column adds nothing
    const double r0v = r0[min(wave + 4 * min(lane, 31), n - 1)];         // lane k holds r0 of this wave's k-th row
    {
        double a0[6], a1[6];
#pragma unroll 1
        for (int rb = wave; rb < n; rb += 24) {                          // six rows of the wave at a time: their twelve loads in flight
#pragma unroll
            for (int u = 0; u < 6; u++) { const double *Jr = J + (size_t)min(rb + 4 * u, n - 1) * n; a0[u] = Jr[c0]; a1[u] = Jr[c1]; }
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int r = rb + 4 * u;
                const double sum = vilf_wave_sum64(a0[u] * dx0 + a1[u] * dx1);
                const double t = readlane_f64(r0v, (r - wave) >> 2) + sum;
                if (r < n && lane == 0) { s_t[r] = t; cost += 0.5 * t * t; }
            }
        }
    }
    if (GRAD) {
        lds_barrier();
        const int pc = (tid < VB_P) ? s_pcol[tid] : -1;
        if (wave < 3) {
            const int per = (n + 2) / 3, ra = wave * per, rb = min(n, ra + per);
            double g0 = 0, g1 = 0;
#pragma unroll 8
            for (int r = ra; r < rb; r++) { const double t = s_t[r]; const double *Jr = J + (size_t)r * n; g0 += Jr[c0] * t; g1 += Jr[c1] * t; }
            s_tmp[85 * wave + c0] = g0;                                    // (clamped columns rewrite column n - 1 with its own value: lane n - 1 - 64 / the lanes >= n hold the same sum)
            if (lane + 64 < n) s_tmp[85 * wave + lane + 64] = g1;
        }
        lds_barrier();
        if (pc >= 0) { double g = s_tmp[pc]; g += s_tmp[85 + pc]; g += s_tmp[170 + pc]; gprior = g; }
        lds_barrier();
    }
}

// the window of this workgroup: blockIdx.x + w0, or through the dense list of the windows still running (-1: the list is shorter than the grid)
__device__ __forceinline__ int vb_window(const VbBatch &b) {
    const int i = b.live_it;
    if (!b.live_ctl || i < 3 || !b.live_ctl[i - 2]) return blockIdx.x + b.w0;      // list i - 1 exists iff something had stopped by the end of iteration i - 2
    if ((int)blockIdx.x >= b.live_ctl[64 + i - 1]) return -1;
    return b.live_buf[(size_t)((i - 1) & 1) * b.B + blockIdx.x];
}
// ------------------------------------------------------------------------------------------------------------------
// k_linearize
//
// LDS plan (dynamic, one 46.6 KB region; with the static arrays the kernel stays under 80 KB => two workgroups per CU):
//                     s_U  IMU stage: 10 x [16 rows][32 cols] = [S*Jraw | S*r | 0], then the same region becomes
//                     s_X  [2*VB_CHUNK rows][VB_XLD]  the chunk of robustified factor rows [Jj(6) | Ji(6) | r]
// pairD (55 x 120 per-frame-pair products) is accumulated in global memory (L2-resident, 53 KB per window).
// Factors arrive SORTED BY FRAME PAIR (host), so each pair's rows are contiguous in a chunk and
//   [Jj Ji r]^T [Jj Ji r]  (13 x 13: JjJj, JjJi, JiJi, Jj^T r, Ji^T r)
// is one v_mfma_f64_16x16x4_f64 accumulation per 4 rows: the block-sparse J^T J / J^T r accumulate on the matrix cores,
// deterministic (each pair is owned by one wave, fixed order).
#define LIN_LDS_DOUBLES VB_LIN_LDS_DOUBLES

// pairD element of pair p, zero when the pair has no factor (unconditional load + select: a conditional load would serialise on
// vmcnt(0); untouched slots hold garbage that never becomes an arithmetic input)
__device__ __forceinline__ double pd_get(const double *pd, const int *s_pcn, int p, int e) { const double v = pd[p * VB_PAIRD + e]; return (s_pcn[p] > 0) ? v : 0.0; }

__device__ __forceinline__ int pair_elem(int row, int col) {   // element of the 16x16 X^T X tile -> slot in pairD (or -1)
    if (row < 6) {
        if (col < 6) return 6 * row + col;
        if (col < 12) return 36 + 6 * row + (col - 6);
        if (col == 12) return 108 + row;
        return -1;
    }
    if (row < 12) {
        if (col >= 6 && col < 12) return 72 + 6 * (row - 6) + (col - 6);
        if (col == 12) return 114 + (row - 6);
    }
    return -1;
}

// JAC = false: the launch after the LAST solve of the iteration budget. Its linearisation would never be used (Ceres tests max_num_iterations before the gradient),
// so it only takes the step: candidate, cost of every factor (residuals only), accept / reject — a ninth of the linearisations of a solve.
// FUSED (k_iter): the window and its workspace slot are handed in (a persistent workgroup takes windows from a queue and keeps ONE workspace for all of them);
// iteration_zero = 2 then means "linearise at x again" (the workspace of x is gone — it was this workgroup's scratch — and an invalid step asks for another solve from it).
// Returns 0: linearised (a solve may follow), 1: the step was invalid (nothing linearised), 2: the window had stopped before.
// Everything that used to be static LDS lives behind the factor chunk in the dynamic region, so that a kernel that runs the solve in the same workgroup can overlay it.
struct LinShared {
    double pose[77], sb[99], R[99], ric[9], tic[3];
    double lidJ[10 * 72], lidr[64], grad[176];
    double pt[VB_NPAIR * PT_LD];
    double dx[VB_PRIOR_LD];
    double red[NT];
    double def[2];
    int pcol[VB_P], pst[VB_NPAIR], pcn[VB_NPAIR];
    int last, defi[2];
};
static_assert(sizeof(LinShared) <= VB_LIN_SHARED_BYTES, "VB_LIN_SHARED_BYTES (vilf_batch.hpp) must cover LinShared");
template <bool JAC, bool SPLIT = false, bool FUSED = false>
__device__ __forceinline__ int linearize_body(const VbBatch &b, int iteration_zero, int w_in = 0, size_t ww_in = 0, int tid_in = 0) {
    // SPLIT (small batches, k_linearize_split): the window's work is dealt to NR = chunks + 2 workgroups — role s < NCH evaluates the factor slots of chunk s and forms
    // their pair products, role NCH the IMU factors, role NCH + 1 the LiDAR factors and the prior's cost; every role runs the set-up (state, candidate, tables) itself
    // and changes nothing in VbState. The workgroup that arrives last (a counter) has the same set-up in its LDS and everything else in global memory: it applies the
    // trust-region bookkeeping the others left undone and runs the phases behind the chunk loop. The result is the one of the single workgroup to the bit: a pair whose
    // factors span chunks hands its MFMA accumulators from chunk to chunk (global memory + a flag, only to a HIGHER workgroup index: the dispatcher starts workgroups in
    // order, so the writer is always running or done), and the per-thread cost sums are added up by the last workgroup in the single workgroup's order.
    const int tid = FUSED ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    const int NR = SPLIT ? b.split_nr : 1, NCH = NR - 2;
    const int w = FUSED ? w_in : (SPLIT ? (int)blockIdx.x / NR : vb_window(b));
    const int role = SPLIT ? (int)blockIdx.x - w * NR : 0;
    if (w < 0) return 2;
    const bool do_vis = !SPLIT || role < NCH, do_imu = !SPLIT || role == NCH, do_lp = !SPLIT || role == NCH + 1;
    int *sctl = SPLIT ? b.split_ctl + (size_t)w * VB_SPLIT_CTL : nullptr;
    double *sbuf = SPLIT ? b.split_buf + (size_t)w * VB_SPLIT_DBL : nullptr, *scost = SPLIT ? sbuf + VB_SPLIT_CARRY : nullptr;
    extern __shared__ double s_dyn[];
    LinShared &LS = *reinterpret_cast<LinShared *>(s_dyn + LIN_LDS_DOUBLES);
    int &s_last = LS.last;
    double *s_def = LS.def;
    int *s_defi = LS.defi;
    auto arrive = [&]() -> bool {                          // all threads; true in the workgroup that arrives last
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = (__hip_atomic_fetch_add(sctl + VB_SPLIT_CNT + (b.split_gen & 7), 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == NR - 1) ? 1 : 0;
        __syncthreads();
        if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        return s_last != 0;
    };
    VbState *st = b.st + w;
    // the arrival counter four generations ahead is cleared by every workgroup of this launch (idempotent; no launch in flight uses it): a launch that died half way
    // leaves a counter that is wiped long before its slot comes round again
    if (SPLIT && tid == 0) sctl[VB_SPLIT_CNT + ((b.split_gen + 4) & 7)] = 0;
    const int lit = b.live_it;
    const bool lists = !iteration_zero && JAC && b.live_ctl != nullptr && lit >= 1;
    const int stopped_before = lists ? b.live_ctl[lit - 1] : 0;          // has any window of the batch stopped by the end of the previous iteration?
    if (!iteration_zero && st->done) { if (lists && tid == 0) b.live_ctl[lit] = 1; return 2; }
    // thread 0, at the launch's exits: pass the flag on, and — once something has stopped — put this window on the next iteration's list
    auto still_live = [&]() {
        if (!lists) return;
        if (st->done || stopped_before) b.live_ctl[lit] = 1;
        if (stopped_before && !st->done) b.live_buf[(size_t)(lit & 1) * b.B + atomicAdd(b.live_ctl + 64 + lit, 1)] = w;
    };
    // thread 0: what the trust-region step computation writes (from s_def / s_defi): DoglegStrategy's step norm and model cost change; HandleInvalidStep + StepIsInvalid
    auto step_bookkeeping = [&]() {
        const int valid = s_defi[0];
        if (!s_defi[1]) { st->dogleg_step_norm = s_def[0]; st->model_cost_change = s_def[1]; }
        if (!valid) {
            st->num_consecutive_invalid += 1;
            st->mu *= 10.0;
            st->reuse = 0;
            st->solve_failed = 0;
            if (st->num_consecutive_invalid >= 5) { st->done = 1; st->termination = 4; }
        } else st->num_consecutive_invalid = 0;
    };
    // Two linearisation workspaces per window. Iteration zero fills set 0 at the initial state. Later launches ARE the trust-region step (what k_step was): they form
    // the dogleg step from the last solve, linearise at the CANDIDATE into the set that does not belong to x — the candidate's cost falls out of the same pass over
    // the factors, so there is no separate cost-only pass — and flip st->ws when the step is accepted; a rejected step leaves x, its cost and its set untouched.
    const int wset = iteration_zero ? 0 : (st->ws ^ 1);
    const size_t ww = FUSED ? ww_in : (size_t)wset * b.B + w;              // window index inside the written set (FUSED: the workgroup's own slot)

    double *s_X = s_dyn;
    double *s_U = s_dyn;                      // the IMU staging area (10 x 512) shares the region with the factor chunk that follows it
    double *pd = b.pairD + ww * VB_NPAIR * VB_PAIRD;
    double *s_pose = LS.pose, *s_sb = LS.sb, *s_R = LS.R, *s_ric = LS.ric, *s_tic = LS.tic;
    double *s_lidJ = LS.lidJ, *s_lidr = LS.lidr, *s_grad = LS.grad;
    double *s_pt = LS.pt;
    double *s_dx = LS.dx;
    int *s_pcol = LS.pcol, *s_pst = LS.pst, *s_pcn = LS.pcn;           // pair table: start inside the class list, factor count
    double *s_red = LS.red;

    LSTAMP(0);
    const int F = b.n_feat[w], nfac = b.n_fac[w];
    const size_t FM = b.Fmax, FC = b.FACmax;
    double *pose_g = b.pose + (size_t)w * 77, *sb_g = b.sb + (size_t)w * 99;
    double *feat_x = b.feat + (size_t)w * FM, *cfeat = b.cand_feat + (size_t)w * FM;
    const double *feat = iteration_zero ? feat_x : cfeat;                    // the point of this linearisation
    const double *ex = b.ex + (size_t)w * 7;
    const uint8_t *f_const0 = b.f_const + (size_t)w * FM;

    if (tid < 77) s_pose[tid] = pose_g[tid];
    if (tid < 99) s_sb[tid] = sb_g[tid];
    if (FUSED && JAC) {
        // the slot's W rows carry another window's feature ranges: zero the rows this window uses (coalesced 16-byte stores; the factor lanes write behind several barriers)
        double2_t *Wz = reinterpret_cast<double2_t *>(b.W + ww * FM * VB_WLD);
        const int nz = min((F + 3) & ~3, (int)FM) * (VB_WLD / 2);
        for (int i = tid; i < nz; i += NT) Wz[i] = double2_t{0.0, 0.0};
    }
    double stepsq = 0;
    if (!iteration_zero) {
        // ---- DoglegStrategy::ComputeTraditionalDoglegStep + model cost change (thread 0), then the candidate x (+) delta -----------------------------
        int *s_flagi = reinterpret_cast<int *>(s_red);                       // [0] valid step; the coefficients follow as doubles
        if (tid == 0) {
            int valid = 1;
            double ca = 0, cb = 0;
            const int sf0 = st->solve_failed;
            double norm0 = 0, mcc0 = 0;
            if (sf0) valid = 0;
            else {
                const double radius = st->radius, alpha = st->alpha;
                const double gradient_norm = sqrt(st->grad_sqnorm), gn_norm = sqrt(st->gn_sqnorm);
                double norm;
                if (gn_norm <= radius) { ca = 0; cb = 1; norm = gn_norm; }
                else if (gradient_norm * alpha >= radius) { ca = -(radius / gradient_norm); cb = 0; norm = radius; }
                else {
                    const double b_dot_a = alpha * st->gy;                  // -alpha * gradient_ . gauss_newton_step_
                    const double a_sq = (alpha * gradient_norm) * (alpha * gradient_norm);
                    const double bma = a_sq - 2 * b_dot_a + gn_norm * gn_norm;
                    const double c = b_dot_a - a_sq;
                    const double d = sqrt(c * c + bma * (radius * radius - a_sq));
                    const double beta = (c <= 0) ? (d - c) / bma : (radius * radius - a_sq) / (d + c);
                    ca = -alpha * (1.0 - beta); cb = beta;
                    norm = sqrt(ca * ca * st->grad_sqnorm - 2 * ca * cb * st->gy + cb * cb * st->gn_sqnorm);
                }
                norm0 = norm;
                // step = ca * v - cb * y ; model_cost_change = -step.g~ - 0.5 step^T H~ step, with (H~ + mu D^2) y = g~
                const double mu = st->mu_used, G2 = st->grad_sqnorm;
                const double sg = ca * G2 - cb * st->gy;
                const double vHy = G2 - mu * st->gy;
                const double yHy = st->gy - mu * st->gn_sqnorm;
                const double sHs = ca * ca * st->Jg2 - 2 * ca * cb * vHy + cb * cb * yHy;
                const double mcc = -sg - 0.5 * sHs;
                mcc0 = mcc;
                if (!(mcc > 0.0)) valid = 0;
            }
            s_def[0] = norm0; s_def[1] = mcc0; s_defi[0] = valid; s_defi[1] = sf0;
            if (!SPLIT) step_bookkeeping();                                  // (SPLIT: by the workgroup that arrives last — the others still read these fields)
            s_flagi[0] = valid;
            s_red[1] = ca; s_red[2] = cb;
            if (!SPLIT && !valid) still_live();                              // an invalid step: the window goes on with a larger mu (unless that was the fifth)
        }
        __syncthreads();
        if (!s_flagi[0]) {
            if (SPLIT) { if (arrive()) { if (tid == 0) step_bookkeeping(); } }
            if (!FUSED) return 1;
            // FUSED: the invalid step raised mu and asks for another solve from the linearisation at x — which was this workgroup's scratch and is gone. x is linearised
            // again (s_pose / s_sb still hold x, the features are read from feat_x): the same arithmetic on the same state gives the same bits.
            if (st->done) return 2;                        // (thread 0 wrote it before the barrier above: the fifth invalid step in a row ends the window)
            iteration_zero = 2;
            feat = feat_x;
        }
        if (!iteration_zero) {
        const double ca = s_red[1], cb = s_red[2];
        const double *scale_g = b.scale + (size_t)w * (VB_P + FM), *diag_g = b.diag + (size_t)w * (VB_P + FM);
        const double *grad_g = b.grad + (size_t)w * (VB_P + FM), *gn_g = b.gn + (size_t)w * (VB_P + FM);
        // delta = (ca * gradient_ + cb * gauss_newton_step_) ./ diagonal_ .* jacobian_scaling   (s_grad is free until the gradient phase)
        if (tid < VB_P) {
            int a, l; unperm(tid, a, l);
            s_grad[15 * a + l] = (ca * grad_g[tid] + cb * gn_g[tid]) / diag_g[tid] * scale_g[tid];
        }
        lds_barrier();
        if (tid < VB_NF) {
            double xp[7];
            pose_plus(s_pose + 7 * tid, s_grad + 15 * tid, xp);
            for (int k = 0; k < 7; k++) { const double d = s_pose[7 * tid + k] - xp[k]; stepsq += d * d; s_pose[7 * tid + k] = xp[k]; }
            for (int k = 0; k < 9; k++) { const double d = s_grad[15 * tid + 6 + k]; stepsq += d * d; s_sb[9 * tid + k] += d; }
        }
        for (int f = tid; f < F; f += NT) {
            double v = feat_x[f];
            if (!f_const0[f]) {
                const double d = (ca * grad_g[VB_P + f] + cb * gn_g[VB_P + f]) / diag_g[VB_P + f] * scale_g[VB_P + f];
                stepsq += d * d;
                v += d;
            }
            cfeat[f] = v;
        }
        __threadfence_block();                                               // cfeat is read back by other threads of this workgroup below (after the next barrier)
        }
    }
    // pair table: lane p of every wave keeps pair p's start inside its class list and its factor count in registers, and the wave the set of its own class's pairs
    // as a bit mask — the chunk loop walks the mask and fetches a pair's entry with v_readlane (three dependent LDS reads per pair and chunk before)
    unsigned long long cls_mask;
    int my_pst, my_pcn;
    {
        const int *pt = b.pair_off + (size_t)w * VB_PTAB;
        const int pp = min(lane, VB_NPAIR - 1), v1 = pt[2 * pp + 1];
        my_pst = pt[2 * pp]; my_pcn = v1 & 0xffffff;
        if (wave == 0 && lane < VB_NPAIR) { s_pst[lane] = my_pst; s_pcn[lane] = my_pcn; }
        const bool mine = lane < VB_NPAIR && (v1 >> 24) == wave && my_pcn > 0;
        cls_mask = __ballot(mine);
    }
    if (JAC && do_imu) for (int i = tid; i < 10 * 512; i += NT) s_U[i] = 0.0;
    __syncthreads();
    if (tid < VB_NF) q_toR(q_load(s_pose + 7 * tid + 3), s_R + 9 * tid);
    if (tid == 32) { q_toR(q_load(ex + 3), s_ric); s_tic[0] = ex[0]; s_tic[1] = ex[1]; s_tic[2] = ex[2]; }
    prior_setup(b, w, s_pose, s_sb, s_pcol, s_dx, tid);   // contains __syncthreads
    if (tid >= 128 && tid < 128 + VB_NPAIR) {               // per-frame-pair geometry tables (wave 2)
        const int p = tid - 128;
        int j = 1; while (j * (j + 1) / 2 <= p) j++;
        const int i = p - j * (j - 1) / 2;
        pair_table(s_pose + 7 * i, s_R + 9 * i, s_pose + 7 * j, s_R + 9 * j, s_ric, s_tic, s_pt + PT_LD * p);
    }
    LSTAMP(1);

    double cost_local = 0;
    // ---- IMU raw (wave 3, lanes 0..9) and LiDAR between-factors (wave 1, lanes 0..9) ------------------------------
    if (do_imu && tid >= 192 && tid < 202) {
        const int k = tid - 192;
        const double *rec = b.imu + ((size_t)w * 10 + k) * IMU_REC;
        if (rec[287] != 0.0) {
            double r[15];
            if (JAC) {
                imu_raw_eval<true, 32, false>(s_pose + 7 * k, s_sb + 9 * k, s_pose + 7 * (k + 1), s_sb + 9 * (k + 1), rec, b.G, r, s_U + 512 * k);
                for (int m = 0; m < 15; m++) s_U[512 * k + 32 * m + 30] = r[m];
            } else {
                imu_raw_eval<false>(s_pose + 7 * k, s_sb + 9 * k, s_pose + 7 * (k + 1), s_sb + 9 * (k + 1), rec, b.G, r, nullptr);
                double acc = 0;
                for (int i = 0; i < 15; i++) { double sq = 0; for (int m = i; m < 15; m++) sq += rec[IMU_SQRT + 15 * i + m] * r[m]; acc += sq * sq; }
                cost_local += 0.5 * acc;
            }
        }
    }
    if (do_lp && tid >= 64 && tid < 74) {
        const int k = tid - 64;
        if (b.use_lidar) {
            const double *lc = b.lidar + ((size_t)w * 10 + k) * 7;
            if (JAC) {
                double Ji[36], Jj[36];
                lidar_between_eval<true>(s_pose + 7 * k, s_pose + 7 * (k + 1), q_load(b.qil), b.til, q_load(lc), lc + 4, s_lidr + 6 * k, Ji, Jj);
                for (int rr = 0; rr < 6; rr++) for (int c = 0; c < 6; c++) { s_lidJ[72 * k + 12 * rr + c] = Ji[6 * rr + c]; s_lidJ[72 * k + 12 * rr + 6 + c] = Jj[6 * rr + c]; }
            } else lidar_between_eval<false>(s_pose + 7 * k, s_pose + 7 * (k + 1), q_load(b.qil), b.til, q_load(lc), lc + 4, s_lidr + 6 * k, nullptr, nullptr);
        } else {
            if (JAC) for (int e = 0; e < 72; e++) s_lidJ[72 * k + e] = 0;
            for (int e = 0; e < 6; e++) s_lidr[6 * k + e] = 0;
        }
    }
    lds_barrier();
    LSTAMP(2);
    // ---- IMU: X = sqrt_info * [Jraw | r] on MFMA, in place (imu_factor.h:64,93,126,145,160) -------------------------
    if (JAC && do_imu) for (int task = wave; task < 20; task += 4) {
        const int k = task >> 1, ct = task & 1;
        double *Xk = s_U + 512 * k;
        const double *S = b.imu + ((size_t)w * 10 + k) * IMU_REC + IMU_SQRT;
        double av[4], bv[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) {
            const int i = lane & 15, kk = 4 * s4 + (lane >> 4);
            av[s4] = (i < 15 && kk < 15) ? S[15 * i + kk] : 0.0;
            bv[s4] = Xk[32 * kk + 16 * ct + (lane & 15)];
        }
        double4_t acc = {0, 0, 0, 0};
#pragma unroll
        for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; q++) Xk[32 * ((lane >> 4) + 4 * q) + 16 * ct + (lane & 15)] = acc[q];
    }
    lds_barrier();
    LSTAMP(3);
    // ---- IMU: [J r]^T [J r] on MFMA -> imuH (30x30), imug (30), cost; LiDAR blocks on the VALU ------------------------
    if (JAC) {
        double *imuH = b.imuH + ww * 9000, *imug = b.imug + ww * 300;
        if (do_imu)
        for (int task = wave; task < 30; task += 4) {
            const int k = task / 3, tt = task - 3 * k;
            const int ti = (tt == 0) ? 0 : 1, tj = (tt == 2) ? 1 : 0;
            const double *Xk = s_U + 512 * k;
            double4_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int s4 = 0; s4 < 4; s4++) {
                const int row = 4 * s4 + (lane >> 4);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xk[32 * row + 16 * ti + (lane & 15)], Xk[32 * row + 16 * tj + (lane & 15)], acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int p = 16 * ti + (lane >> 4) + 4 * q, c = 16 * tj + (lane & 15);
                if (p < 30 && c < 30) { imuH[900 * k + 30 * p + c] = acc[q]; if (ti != tj) imuH[900 * k + 30 * c + p] = acc[q]; }
                if (p == 30 && c < 30) imug[30 * k + c] = acc[q];
                if (p == 30 && c == 30) cost_local += 0.5 * acc[q];
            }
        }
        LSTAMP(10);
        double *lidH = b.lidH + ww * 1440, *lidg = b.lidg + ww * 120;
        if (do_lp) {
        for (int idx = tid; idx < 1440; idx += NT) {
            const int k = idx / 144, e = idx - 144 * k, p = e / 12, q = e - 12 * p;
            double s = 0;
            for (int row = 0; row < 6; row++) s += s_lidJ[72 * k + 12 * row + p] * s_lidJ[72 * k + 12 * row + q];
            lidH[idx] = s;
        }
        for (int idx = tid; idx < 120; idx += NT) {
            const int k = idx / 12, p = idx - 12 * k;
            double s = 0;
            for (int row = 0; row < 6; row++) s += s_lidJ[72 * k + 12 * row + p] * s_lidr[6 * k + row];
            lidg[idx] = s;
        }
        }
    }
    if (SPLIT && do_imu) scost[tid] = cost_local;                          // cost partials of the roles, per thread: [0] IMU, [1] LiDAR, [2] prior, [3 + s] chunk s
    if (SPLIT) cost_local = 0;
    if (do_lp && tid >= 64 && tid < 74) { const int k = tid - 64; double s = 0; for (int m = 0; m < 6; m++) s += s_lidr[6 * k + m] * s_lidr[6 * k + m]; cost_local += 0.5 * s; }
    LSTAMP(11);
    if (SPLIT && do_lp) { scost[256 + tid] = cost_local; cost_local = 0; }
    // the prior: cost, and — kept in a register until the gradient phase — the gradient entry g0 + H0 dx of this thread's reduced variable (SPLIT: the cost by the
    // LiDAR / prior role, the gradient by the workgroup that arrives last)
    double gprior = 0;
    if (!SPLIT || do_lp) {
        double pcost;
        prior_products_j<(JAC && !SPLIT)>(b, w, s_dx, s_pcol, s_red, s_grad, tid, pcost, gprior);
        if (SPLIT) scost[512 + tid] = pcost; else cost_local += pcost;
    }
    LSTAMP(13);
    lds_barrier();                                                       // the IMU staging area is dead: the region becomes s_X. (LDS-only barriers from here to the end of the
                                                                           // chunk loop: the imuH / lidH stores above drain under the visual factors instead of being waited for here)
    if (tid == 0) s_X[2 * VB_CHUNK * VB_XLD] = 0.0;                       // the zero the masked lanes of the pair products read (behind the rows; nothing else writes it)
    LSTAMP(4);
    // ---- visual factors: chunks of 256 pair-sorted factors -> LDS rows -> MFMA X^T X per pair ------------------------
    const int *f_start = b.f_start + (size_t)w * FM;
    const uint8_t *f_const = b.f_const + (size_t)w * FM;
    const double *facrec = b.facrec + (size_t)w * FC * 8;
    double *facw = b.facw + (FUSED ? ww : (size_t)w) * VB_FACW * FC;
    double *W = b.W + ww * FM * VB_WLD;
    long long t_eval = 0, t_sync1 = 0, t_mfma = 0, t_sync2 = 0, t_a = 0;
    int pe[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) pe[q4] = pair_elem((lane >> 4) + 4 * q4, lane & 15);   // C-tile register -> pairD slot (fixed per lane)
    double4_t cacc = {0, 0, 0, 0}, cacc1 = {0, 0, 0, 0};            // the open pair of this wave's class list across chunk boundaries
    unsigned long long mrem = cls_mask;                              // the pairs of this wave's class not finished yet, ascending = class-list order
    if (!JAC) {
        // residuals only: one lane per factor slot, no rows, no products
        for (int fac = tid; fac < nfac; fac += NT) {
            const double4_t *rp = reinterpret_cast<const double4_t *>(facrec + (size_t)fac * 8);
            const double4_t ra = rp[0], rb = rp[1];
            const double pts_i[3] = {ra[0], ra[1], ra[2]}, pts_j[3] = {ra[3], rb[0], rb[1]};
            const unsigned long long ia = __double_as_longlong(rb[2]), ib = __double_as_longlong(rb[3]);
            if ((ib >> 17) & 1) continue;                     // unused slot of the chunk-interleaved layout
            const int f = (int)(ia & 0xffffffffu), fi = (int)(ib & 0xff), fj = (int)((ib >> 8) & 0xff);
            double r[2];
            projection_eval_pair<false>(s_pt + PT_LD * pair_index(fi, fj), s_ric, s_tic, pts_i, pts_j, feat[f], b.sqrt_info, r, nullptr, nullptr, nullptr);
            double rho0, sw;
            cauchy(r[0] * r[0] + r[1] * r[1], b.cauchy_b, rho0, sw);
            cost_local += 0.5 * rho0;
        }
    } else {
    // one coalesced 64-byte record per factor slot (points, feature, slot, frames, const flag, null flag); only the inverse depth is a gather. The NEXT chunk's records
    // are requested behind the evaluation (its registers are free then) and arrive under the pair products; the inverse depths follow behind the products: the
    // evaluation of a chunk no longer starts with a trip to HBM and a dependent gather (a fifth of the chunk loop)
    const int c_begin = SPLIT ? role * VB_CHUNK : 0, c_end = SPLIT ? (do_vis ? min(nfac, (role + 1) * VB_CHUNK) : 0) : nfac;
    double4_t ra_n = {0, 0, 0, 0}, rb_n = {0, 0, 0, 0};
    double fv_n = 0;
    if (c_begin < c_end) {
        const double4_t *rp = reinterpret_cast<const double4_t *>(facrec + (size_t)min(c_begin + tid, nfac - 1) * 8);
        ra_n = rp[0]; rb_n = rp[1];
        fv_n = feat[(int)(__double_as_longlong(rb_n[2]) & 0xffffffffu)];
    }
    for (int c0 = c_begin; c0 < c_end; c0 += VB_CHUNK) {
        t_a = TICK();
        const int q = c0 + tid;
        double *x0 = s_X + (2 * min(tid, VB_CHUNK - 1)) * VB_XLD, *x1 = x0 + VB_XLD;
        const double4_t ra = ra_n, rb = rb_n;
        const double fv = fv_n;
        const unsigned long long ia = __double_as_longlong(rb[2]), ib = __double_as_longlong(rb[3]);
        if (tid >= VB_CHUNK) {
        } else if (q < nfac && !((ib >> 17) & 1)) {
            const double pts_i[3] = {ra[0], ra[1], ra[2]}, pts_j[3] = {ra[3], rb[0], rb[1]};
            const int f = (int)(ia & 0xffffffffu), slot = (int)(ia >> 32), fi = (int)(ib & 0xff), fj = (int)((ib >> 8) & 0xff);
            const bool fc = ((ib >> 16) & 1) != 0;
            double r[2], Ji[12], Jj[12], Jf[2];
            projection_eval_pair<true>(s_pt + PT_LD * pair_index(fi, fj), s_ric, s_tic, pts_i, pts_j, fv, b.sqrt_info, r, Ji, Jj, Jf);
            double rho0, sw;
            cauchy(r[0] * r[0] + r[1] * r[1], b.cauchy_b, rho0, sw);
            cost_local += 0.5 * rho0;
#pragma unroll
            for (int c = 0; c < 6; c++) { x0[c] = sw * Jj[c]; x0[6 + c] = sw * Ji[c]; x1[c] = sw * Jj[6 + c]; x1[6 + c] = sw * Ji[6 + c]; }
            x0[12] = sw * r[0]; x1[12] = sw * r[1];
            if (!fc) {
                const double jf0 = sw * Jf[0], jf1 = sw * Jf[1];
                // 16-byte stores (the row's six doubles start at a multiple of 48 bytes, a record at a multiple of 64): 7 store instructions per lane instead of 14 —
                // every lane of a wave writes to lines of its own, so the instruction count is what the memory pipeline sees
                double2_t *Wr = reinterpret_cast<double2_t *>(W + (size_t)f * VB_WLD + 6 * fj);
                double2_t *fw = reinterpret_cast<double2_t *>(facw + (size_t)slot * VB_FACW);
#pragma unroll
                for (int c = 0; c < 6; c += 2) {
                    Wr[c >> 1] = double2_t{x0[c] * jf0 + x1[c] * jf1, x0[c + 1] * jf0 + x1[c + 1] * jf1};                               // H_pf block of frame j (exclusive owner)
                    fw[c >> 1] = double2_t{x0[6 + c] * jf0 + x1[6 + c] * jf1, x0[7 + c] * jf0 + x1[7 + c] * jf1};                       // partial of the anchor-frame block (one 64-byte record per factor)
                }
                fw[3] = double2_t{jf0 * jf0 + jf1 * jf1, jf0 * x0[12] + jf1 * x1[12]};
            }
        } else {
#pragma unroll
            for (int c = 0; c < VB_XLD; c++) { x0[c] = 0; x1[c] = 0; }
        }
        { long long t_b = TICK(); t_eval += t_b - t_a; t_a = t_b; }
        lds_barrier();                            // (the rows in LDS; the W / facw stores of this chunk are read after the loop, behind a full barrier)
        { long long t_b = TICK(); t_sync1 += t_b - t_a; t_a = t_b; }
        if (c0 + VB_CHUNK < c_end) {
            const double4_t *rp = reinterpret_cast<const double4_t *>(facrec + (size_t)min(c0 + VB_CHUNK + tid, nfac - 1) * 8);
            ra_n = rp[0]; rb_n = rp[1];
        }
        const int x0c = VB_CLS * (c0 / VB_CHUNK);                              // this chunk holds the class-local positions [x0c, x0c + VB_CLS) of every class
        if constexpr (SPLIT) {
            // This workgroup has ONE chunk. The pairs of this wave's class that have factors in it, in an order that keeps the chain between the chunks short: first
            // the pair that runs on into the next chunk (its accumulators are published as soon as this chunk's rows are in), then the pairs that lie inside, last the
            // pair that began in an earlier chunk (its accumulators come from the previous chunk's workgroup). Within a pair the rows are accumulated in the same order
            // as by the single workgroup.
            double *carry_out = sbuf + ((size_t)(role * 4 + wave) * 64 + lane) * 8;
            const double *carry_in = sbuf + ((size_t)(max(role, 1) * 4 - 4 + wave) * 64 + lane) * 8;
            int *flag_out = sctl + 1 + 4 * role + wave, *flag_in = sctl + 1 + 4 * max(role - 1, 0) + wave;
#pragma unroll 1
            for (int pass = 0; pass < 3; pass++) {
                unsigned long long m = cls_mask;
                while (m) {
                    const int p = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
                    m &= m - 1;
                    const int pst = __builtin_amdgcn_readlane(my_pst, p), pcn_ = __builtin_amdgcn_readlane(my_pcn, p);
                    if (pst + pcn_ <= x0c) continue;                       // finished in an earlier chunk
                    if (pst >= x0c + VB_CLS) break;                        // begins in a later one
                    const bool in = pst < x0c, out = pst + pcn_ > x0c + VB_CLS;
                    if ((in ? 2 : (out ? 0 : 1)) != pass) continue;
                    double4_t acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                    if (in) {
                        // bounded wait for the previous chunk's accumulators (flag = this launch's generation). The writer has the lower workgroup index and is
                        // normally running or done; if it never publishes — not scheduled, dead — the window is marked (dev_error: the host returns VILF_ERR_DEVICE
                        // for it) and the launch ends instead of hanging the device
                        int spins = 0;
                        bool got = false;
                        for (; spins < VB_SPLIT_SPIN_MAX; spins++) {
                            if (__hip_atomic_load(flag_in, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == b.split_gen) { got = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (got) {
                            const double2_t *ci = reinterpret_cast<const double2_t *>(carry_in);
                            const double2_t c0_ = ci[0], c1_ = ci[1], c2_ = ci[2], c3_ = ci[3];
                            acc = double4_t{c0_[0], c0_[1], c1_[0], c1_[1]}; acc1 = double4_t{c2_[0], c2_[1], c3_[0], c3_[1]};
                        } else if (lane == 0) __hip_atomic_store(&st->dev_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const int lo = max(pst, x0c), hi = min(pst + pcn_, x0c + VB_CLS);
                    const int r_lo = 2 * (VB_CLS * wave + lo - x0c), r_hi = 2 * (VB_CLS * wave + hi - x0c);
                    auto ld4 = [&](int r0, double *a) {            // a lane outside the pair's rows / the 13 columns reads the zero behind the chunk: the SELECT is on the address —
#pragma unroll                                                     // a select on the loaded value is turned into a branch around the load, each with its own wait for LDS
                        for (int u = 0; u < 4; u++) {
                            const int row = r0 + 4 * u + (lane >> 4);
                            a[u] = s_X[(row < r_hi && (lane & 15) < VB_XLD) ? row * VB_XLD + (lane & 15) : 2 * VB_CHUNK * VB_XLD];
                        }
                    };
                    double a[4], an[4];
                    ld4(r_lo, a);
                    for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
                        ld4(r0 + 16, an);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], a[1], acc1, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], a[2], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], a[3], acc1, 0, 0, 0);
#pragma unroll
                        for (int u = 0; u < 4; u++) a[u] = an[u];
                    }
                    if (out) {
                        double2_t *co = reinterpret_cast<double2_t *>(carry_out);
                        co[0] = double2_t{acc[0], acc[1]}; co[1] = double2_t{acc[2], acc[3]}; co[2] = double2_t{acc1[0], acc1[1]}; co[3] = double2_t{acc1[2], acc1[3]};
                        __threadfence();
                        if (lane == 0 && !(b.split_fault && role == 0)) __hip_atomic_store(flag_out, b.split_gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
#pragma unroll
                        for (int q4 = 0; q4 < 4; q4++) if (pe[q4] >= 0) pd[p * VB_PAIRD + pe[q4]] = acc[q4] + acc1[q4];
                    }
                }
            }
        } else
        // the pairs of this wave's class, in class-list order = ascending position: a cursor carried across the chunks stops at the first pair that starts in a later
        // chunk (walking the whole list in every chunk and skipping cost three dependent LDS reads per pair and chunk: half of this phase)
        while (mrem) {
            const int p = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mrem));
            const int pst = __builtin_amdgcn_readlane(my_pst, p), pcn_ = __builtin_amdgcn_readlane(my_pcn, p);
            if (pst >= x0c + VB_CLS) break;
            const int lo = max(pst, x0c), hi = min(pst + pcn_, x0c + VB_CLS);
            if (lo >= hi) { mrem &= mrem - 1; continue; }
            const int r_lo = 2 * (VB_CLS * wave + lo - x0c), r_hi = 2 * (VB_CLS * wave + hi - x0c);
            // a pair that began in an earlier chunk continues in the registers it was left in (a class list is walked in order: one open pair per wave at most)
            double4_t acc = (pst < x0c) ? cacc : double4_t{0, 0, 0, 0}, acc1 = (pst < x0c) ? cacc1 : double4_t{0, 0, 0, 0};
            auto ld4 = [&](int r0, double *a) {         // unconditional LDS reads: a lane outside the pair's rows / the 13 columns reads the zero behind the chunk. The SELECT is
#pragma unroll                                          // on the ADDRESS: a select on the loaded value is compiled into a branch around the load — four reads, each followed by its
                for (int u = 0; u < 4; u++) {           // own s_waitcnt lgkmcnt(0), per four MFMAs (measured: 950 cycles per group against 256 of matrix-core time)
                    const int row = r0 + 4 * u + (lane >> 4);
                    a[u] = s_X[(row < r_hi && (lane & 15) < VB_XLD) ? row * VB_XLD + (lane & 15) : 2 * VB_CHUNK * VB_XLD];
                }
            };
            double a[4], an[4];
            ld4(r_lo, a);
            for (int r0 = r_lo; r0 < r_hi; r0 += 16) {
                ld4(r0 + 16, an);                         // prefetch the next 16 rows while the 4 MFMAs below execute
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], a[0], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], a[1], acc1, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[2], a[2], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[3], a[3], acc1, 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 4; u++) a[u] = an[u];
            }
            // the pair's products go to pairD (global memory, L2) ONCE, when its last factor has been seen: a pair running on into the next chunk stays in
            // registers — reading the partial sums back cost a dependent global round trip per pair and chunk (the largest part of this loop)
            if (pst + pcn_ > x0c + VB_CLS) { cacc = acc; cacc1 = acc1; break; }          // continues in the next chunk: the cursor stays on it
            else {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) if (pe[q4] >= 0) pd[p * VB_PAIRD + pe[q4]] = acc[q4] + acc1[q4];
                mrem &= mrem - 1;
            }
        }
        { long long t_b = TICK(); t_mfma += t_b - t_a; t_a = t_b; }
        if (c0 + VB_CHUNK < c_end) fv_n = feat[(int)(__double_as_longlong(rb_n[2]) & 0xffffffffu)];
        lds_barrier();                            // (the next chunk overwrites the rows; the gather above stays in flight across it)
        { long long t_b = TICK(); t_sync2 += t_b - t_a; t_a = t_b; }
    }
    __syncthreads();                              // everything the chunks stored (W rows, facw records, pair products) is visible to the phases below
    }
    if (JAC && b.dbg && blockIdx.x == 0 && tid == 0) { b.dbg[64 + 16] = t_eval; b.dbg[64 + 17] = t_sync1; b.dbg[64 + 18] = t_mfma; b.dbg[64 + 19] = t_sync2; }
    if (SPLIT) {
        if (do_vis) scost[(3 + role) * 256 + tid] = cost_local;
        if (!arrive()) return 2;
        // the last to arrive: the trust-region bookkeeping nobody has done yet, and every thread's cost sum in the single workgroup's order (IMU, LiDAR, prior, chunk by chunk)
        if (!iteration_zero && tid == 0) step_bookkeeping();
        double c = 0;
        c += __builtin_nontemporal_load(scost + tid); c += __builtin_nontemporal_load(scost + 256 + tid); c += __builtin_nontemporal_load(scost + 512 + tid);
        for (int k = 0; k < NCH; k++) c += __builtin_nontemporal_load(scost + (3 + k) * 256 + tid);
        cost_local = c;
        __syncthreads();
        if (JAC) { double pc_unused; prior_products_j<true>(b, w, s_dx, s_pcol, s_red, s_grad, tid, pc_unused, gprior); }      // (t = r0 + J0 dx is formed again here: the gradient needs it)
    }
    double gmax = 0, xsq = 0;
    if (JAC) {
    LSTAMP(5);
    // ---- visual pose-pose blocks (frame-block lower triangle) -> Hpp[66][36] ------------------------------------------
    // a pair without factors was never written: it reads as zero
#define PD(p, e) pd_get(pd, s_pcn, (p), (e))
    double v5[8]; int dst5[8];
    {
        // off-diagonal frame blocks (a > bb): one load each, eight entries of a thread in flight; then the diagonal blocks: the ten pairs of a frame, all loads first
        // (a loop of dependent load -> add trips would wait for every load in turn), same order of additions
        // (the loads of this phase are issued here and their results stored AFTER the per-feature phase below: the two phases read what the chunk loop wrote through L2 and
        // are otherwise independent — one after the other each paid its own round trips)
        static_assert(55 * 36 <= 8 * NT, "one pass of eight entries per thread");
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int t = min(tid + k * NT, 55 * 36 - 1), ob = t / 36, e = t - 36 * ob;       // ob-th off-diagonal block = pair (bb, a), bb < a, in pair_index order
            int a = 1; while (a * (a + 1) / 2 <= ob) a++;
            const int bb = ob - a * (a - 1) / 2;
            v5[k] = PD(pair_index(bb, a), 36 + e);
            dst5[k] = (tid + k * NT < 55 * 36) ? 36 * (a * (a + 1) / 2 + bb) + e : -1;
        }
    }
    LSTAMP(6);
    // ---- per-feature Schur vectors: H_ff, g_f, anchor block of the H_pf row -------------------------------------------
    {
        const int *f_nobs = b.f_nobs + (size_t)w * FM, *f_fac0 = b.f_fac0 + (size_t)w * FM;
        double *hf = b.hf + ww * FM, *gf = b.gf + ww * FM;
        for (int f = tid; f < F; f += NT) {
            if (f_const[f]) { hf[f] = 0; gf[f] = 0; continue; }
            const int n = f_nobs[f] - 1, f0 = f_fac0[f];
            double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int t0 = 0; t0 < n; t0 += 4) {           // four factor records per trip in flight, added in factor order (eight per trip: +0.4 % on the solve, measured in round 5)
                double v[4][8];
#pragma unroll
                for (int u = 0; u < 4; u++) {              // a record = four 16-byte loads
                    const double2_t *rp = reinterpret_cast<const double2_t *>(facw + (size_t)(f0 + min(t0 + u, n - 1)) * VB_FACW);
#pragma unroll
                    for (int c = 0; c < 4; c++) { const double2_t q2 = rp[c]; v[u][2 * c] = q2[0]; v[u][2 * c + 1] = q2[1]; }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) if (t0 + u < n) {
#pragma unroll
                    for (int c = 0; c < 8; c++) acc[c] += v[u][c];
                }
            }
            double *Wr = W + (size_t)f * VB_WLD;
#pragma unroll
            for (int c = 0; c < 6; c++) Wr[6 * f_start[f] + c] = acc[c];
            Wr[VB_NPOSE] = acc[7];                      // column 66 carries g_f through the Schur MFMA
            hf[f] = acc[6]; gf[f] = acc[7];
        }
    }
    {
        double *Hpp = b.Hpp + ww * 66 * 36;
#pragma unroll
        for (int k = 0; k < 8; k++) if (dst5[k] >= 0) Hpp[dst5[k]] = v5[k];
        for (int t = tid; t < VB_NF * 36; t += NT) {
            const int a = t / 36, e = t - 36 * a;
            double v[VB_NF - 1], s = 0;
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) v[k] = (k < a) ? PD(pair_index(min(k, a - 1), a), e) : PD(pair_index(a, k + 1), 72 + e);
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) s += v[k];
            Hpp[36 * (a * (a + 1) / 2 + a) + e] = s;
        }
    }
    lds_barrier();
    LSTAMP(7);
    // ---- gradient g = J^T r over the reduced camera/IMU block (frame-major order) ---------------------------------
    double *gout = b.g + ww * VB_P;
    if (tid < VB_P) {
        const int a = tid / 15, l = tid - 15 * a;
        const double *imug = b.imug + ww * 300, *lidg = b.lidg + ww * 120;
        double s = 0;
        if (l < 6) {
            double v[VB_NF - 1];
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) v[k] = (k < a) ? PD(pair_index(min(k, a - 1), a), 108 + l) : PD(pair_index(a, k + 1), 114 + l);
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) s += v[k];
            if (a >= 1) s += lidg[12 * (a - 1) + 6 + l];
            if (a <= 9) s += lidg[12 * a + l];
        }
        if (a >= 1) s += imug[30 * (a - 1) + 15 + l];
        if (a <= 9) s += imug[30 * a + l];
        const int pc = s_pcol[tid];
        s += gprior;                                   // the prior: g0 + H0 dx (prior_grad_split)
        // diagonal of J^T J for this reduced variable (its loads are issued before the gradient is stored: a store in between would fence them off)
        const double *imuHg = b.imuH + ww * 9000, *lidHg = b.lidH + ww * 1440;
        double dg = 0;
        if (l < 6) {
            double v[VB_NF - 1];
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) v[k] = (k < a) ? PD(pair_index(min(k, a - 1), a), 7 * l) : PD(pair_index(a, k + 1), 72 + 7 * l);
#pragma unroll
            for (int k = 0; k < VB_NF - 1; k++) dg += v[k];
            if (a >= 1) dg += lidHg[144 * (a - 1) + 13 * (6 + l)];
            if (a <= 9) dg += lidHg[144 * a + 13 * l];
        }
        if (a >= 1) dg += imuHg[900 * (a - 1) + 31 * (15 + l)];
        if (a <= 9) dg += imuHg[900 * a + 31 * l];
        if (pc >= 0) dg += b.prior_H[(size_t)w * VB_PRIOR_LD * VB_PRIOR_LD + (size_t)pc * (VB_PRIOR_LD + 1)];
        gout[tid] = s;
        s_grad[tid] = s;
        b.diagH[ww * VB_P + tid] = dg;
    }
    lds_barrier();
    // gradient_max_norm = || x - Plus(x, -g) ||_inf (trust_region_minimizer.cc), x_norm = ||x||
    if (tid < VB_NF) {
        const double *gp = s_grad + 15 * tid;
        double d[6] = {-gp[0], -gp[1], -gp[2], -gp[3], -gp[4], -gp[5]}, xp[7];
        pose_plus(s_pose + 7 * tid, d, xp);
        for (int k = 0; k < 7; k++) { gmax = fmax(gmax, fabs(s_pose[7 * tid + k] - xp[k])); xsq += s_pose[7 * tid + k] * s_pose[7 * tid + k]; }
        for (int k = 0; k < 9; k++) { gmax = fmax(gmax, fabs(gp[6 + k])); xsq += s_sb[9 * tid + k] * s_sb[9 * tid + k]; }
    }
    {
        const double *gf = b.gf + ww * FM;
        for (int f = tid; f < F; f += NT) if (!f_const[f]) { gmax = fmax(gmax, fabs(gf[f])); xsq += feat[f] * feat[f]; }
    }
    }
    LSTAMP(8);
    const double cost = block_sum(cost_local, s_red);
    const double gm = block_max(gmax, s_red);
    const double xs = block_sum(xsq, s_red);
    if (iteration_zero) {
        // (iteration_zero == 2, FUSED: x was linearised AGAIN for another solve — its cost and norms are what they were)
        if (tid == 0 && iteration_zero != 2) { st->x_cost = cost; st->gradient_max_norm = gm; st->x_norm = sqrt(xs); st->need_linearize = 0; st->reuse = 0; st->initial_cost = cost; st->ws = 0; }
        LSTAMP(9);
        return 0;
    }
    // ---- accept / reject (trust_region_minimizer.cc): `cost` is the cost at the candidate ---------------------------------------------------------------
    const double step_norm = sqrt(block_sum(stepsq, s_red));
    int *s_acc = reinterpret_cast<int *>(s_red);
    if (tid == 0) {
        int accept = 0;
        st->cand_cost = cost;
        const double x_cost = st->x_cost;
        if (step_norm <= b.parameter_tolerance * (st->x_norm + b.parameter_tolerance)) { st->done = 1; st->termination = 2; }
        else if (fabs(x_cost - cost) <= b.function_tolerance * x_cost) { st->done = 1; st->termination = 1; }
        else {
            const double rd = (x_cost - cost) / st->model_cost_change;
            st->relative_decrease = rd;
            if (rd > b.min_relative_decrease) {
                accept = 1;
                // DoglegStrategy::StepAccepted; x, its cost, its norms and its workspace become the candidate's
                if (rd < 0.25) st->radius *= 0.5;
                if (rd > 0.75) st->radius = fmax(st->radius, 3.0 * st->dogleg_step_norm);
                st->mu = fmax(1e-8, 2.0 * st->mu / 10.0);
                st->reuse = 0;
                st->num_successful += 1;
                st->x_cost = cost;
                if (JAC) { st->gradient_max_norm = gm; st->x_norm = sqrt(xs); st->ws = wset; }      // the cost-only launch wrote no workspace (and nothing reads one after it)
            } else {
                st->radius *= 0.5;     // StepRejected: the next solve re-uses the factorisation of x's workspace
                st->reuse = 1;
            }
        }
        s_acc[0] = accept;
        still_live();
    }
    lds_barrier();
    if (s_acc[0]) {
        if (tid < 77) pose_g[tid] = s_pose[tid];
        if (tid < 99) sb_g[tid] = s_sb[tid];
        for (int f = tid; f < F; f += NT) feat_x[f] = cfeat[f];
    }
    LSTAMP(9);
    return 0;
}
extern "C" __global__ __launch_bounds__(NT) void k_linearize(VbBatch b, int iteration_zero) { (void)linearize_body<true>(b, iteration_zero); }
extern "C" __global__ __launch_bounds__(NT) void k_linearize_last(VbBatch b) { (void)linearize_body<false>(b, 0); }
extern "C" __global__ __launch_bounds__(NT) void k_linearize_split(VbBatch b, int iteration_zero) { (void)linearize_body<true, true>(b, iteration_zero); }

// ------------------------------------------------------------------------------------------------------------------
// k_solve helpers

// LDS accumulate without the read-modify-write round trip: inside one barrier-separated assembly phase every tile entry receives at most one
// addend, so the hardware ds_add_f64 gives the same sum as `+=` — but several of them are in flight per thread instead of one dependent
// read -> add -> write chain per element (the compiler must order plain `+=` on possibly aliasing offsets)
__device__ __forceinline__ void lds_add(double *p, double v) { __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// 16x16 lower Cholesky of the diagonal tile by ONE wave. Lane r (< 16; the other lanes mirror lane r & 15) holds row r in 16
// registers. Left-looking column sweep: for column j the pivot row L[j][0..j-1] is broadcast through SGPRs (v_readlane with a
// compile-time lane), so there is no LDS / ds_bpermute round trip on the pivot chain; the update terms of column j+1 that do not
// depend on pivot j are issued while the v_rsq_f64 + Newton chain of pivot j is in flight. Returns false if a pivot is not
// positive (Eigen LLT semantics).
__device__ bool potrf_tile_wave(double *T, double *s_invd, int lane) {
    const int r = lane & 15;
    double v[16];
#pragma unroll
    for (int c = 0; c < 16; c++) v[c] = T[TIX(r, c)];
    bool ok = true;
    double myinv = 1.0;
    // s[c] accumulates A[r][c] - sum_{p < j} L[r][p] L[c][p] for the columns still to be factorised; kept in v[c] itself.
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double djj = readlane_f64(v[j], j);
        if (!(djj > 0.0)) ok = false;
        const double inv = rsqrt_h3(djj);                       // the pivot chain: one third-order step after the v_rsq_f64 seed
        // L[r][j] for r >= j; rows r < j keep junk in v[j] (upper triangle, never read)
        const double lrj = v[j] * inv;
        v[j] = lrj;
        if (r == j) myinv = inv;
        // right-looking update of the remaining columns with the broadcast column entries L[c][j], c > j
#pragma unroll
        for (int c = j + 1; c < 16; c++) {
            const double lcj = readlane_f64(lrj, c);
            v[c] -= lrj * lcj;
        }
    }
#pragma unroll
    for (int c = 0; c < 16; c++) if (lane < 16) T[TIX(r, c)] = (c <= r) ? v[c] : 0.0;
    if (lane < 16) s_invd[r] = myinv;
    return ok;
}

// scatter-add one element of a symmetric source block into the tile matrix (lower block triangle, diagonal tiles full)
__device__ __forceinline__ void tile_add(double *s_T, int r, int c, double v) {
    const int tr = r >> 4, tc = c >> 4;
    if (tr >= tc) s_T[tile_index(tr, tc) * 256 + TIX(r & 15, c & 15)] += v;
}

#define SNT 512
#define SNW (SNT / 64)
// wave butterflies, then the eight wave sums added in wave order by every thread: three barriers instead of eleven (k_solve owns its CU:
// nothing hides a barrier chain), fixed summation order
__device__ __forceinline__ double block_sum_s(double v, double *s_red) {
    const int tid = threadIdx.x;
    v = vilf_wave_sum64(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int k = 0; k < SNW; k++) r += s_red[k];
    __syncthreads();
    return r;
}


// MFMA Schur reduce of the eliminated inverse depths:  H~_pp -= U^T U,  rhs_p -= U^T t  over the 5x5 pose tile block.
// U row f: columns 0..65 = c_f * S_p * W_f[p], column 66 = c_f * g_f (the rhs rides along as one more column), 67..79 = 0.
// K (the features) is split over 4 wave groups; the two waves of a group (HALF = 0 / 1) own 8 / 7 of the 15 tile pairs — tile
// indices are compile-time, so operands are plain register picks. Loads are unconditional, coalesced (zero-padded 80-wide rows,
// rows >= F are zero) and prefetched three k-steps ahead with no arithmetic on them until they are consumed. Group results are
// subtracted in fixed group order (bit-reproducible).
#define SOLVE_RED_OFF (66 * 256 + 6 * VB_NPAD)   // LDS layout of k_solve: 66 tiles, then g, diag, scale, y, invd, v (VB_NPAD each), then s_red
template <int HALF>
__device__ __forceinline__ void schur_mfma(const double *W, int F, const double *s_cf, const double *s_scale, double *s_T, double *s_y, int lane, int wave) {
    constexpr int TA[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4};
    constexpr int TB[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4};
    constexpr int NP = HALF ? 7 : 8, PB = HALF ? 8 : 0;
    const int c16 = lane & 15, grp = wave >> 1, g4 = lane >> 4;
    const int nsteps = ((F + 3) & ~3) / 4;
    double *s_t = s_T + SOLVE_RED_OFF;            // s_red of k_solve (free between its block sums): row 67 of the reduce
    double sc5[5];
#pragma unroll
    for (int t5 = 0; t5 < 5; t5++) { const int col = 16 * t5 + c16; sc5[t5] = (col < VB_NPOSE) ? s_scale[col] : (col <= VB_NPOSE + 1 ? 1.0 : 0.0); }       // 66: rhs, 67: the Cauchy-point column (see k_solve)
    double4_t acc[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) acc[i] = double4_t{0, 0, 0, 0};
    // Three k-steps of W in flight per wave. The loads are inline asm with hand-placed s_waitcnt: written as plain C++ the compiler sinks every
    // load to its use (it carries the ADDRESS across the trips instead of the data), and each trip then waits out a full global-load latency —
    // the whole phase was that latency. Three fixed register sets (no rotation: a v_mov from a register whose load is still in flight would not
    // be interlocked), vmcnt counts: a set is ready when at most the two younger sets (10 loads) are outstanding.
    double ra[5], rb[5], rc[5];
#define SCHUR_ISSUE(ST, R)                                                                                                         \
    {                                                                                                                              \
        const int st_ = min((ST), nsteps - 1);                                                                                     \
        const double *Wr_ = W + (size_t)(4 * st_ + g4) * VB_WLD + c16;                                                            \
        asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:128\n\t"                           \
                     "global_load_dwordx2 %2, %5, off offset:256\n\tglobal_load_dwordx2 %3, %5, off offset:384\n\t"                \
                     "global_load_dwordx2 %4, %5, off offset:512"                                                                  \
                     : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]) : "v"(Wr_) : "memory");                      \
    }
#define SCHUR_WAIT(N, R) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]) : : "memory");
#define SCHUR_STEP(ST, R)                                                                                                          \
    {                                                                                                                              \
        const double c_ = s_cf[4 * (ST) + g4];                                                                                     \
        double u[5];                                                                                                               \
        _Pragma("unroll") for (int t5 = 0; t5 < 5; t5++) u[t5] = R[t5] * c_ * sc5[t5];                                             \
        _Pragma("unroll") for (int i = 0; i < NP; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[TA[PB + i]], u[TB[PB + i]], acc[i], 0, 0, 0); \
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // nothing of the compiler's own in flight: the counts below are exact
    SCHUR_ISSUE(grp, ra) SCHUR_ISSUE(grp + 4, rb)
    for (int st = grp; st < nsteps; st += 12) {
        SCHUR_ISSUE(st + 8, rc) SCHUR_WAIT(10, ra) SCHUR_STEP(st, ra)
        if (st + 4 < nsteps) { SCHUR_ISSUE(st + 12, ra) SCHUR_WAIT(10, rb) SCHUR_STEP(st + 4, rb) }
        if (st + 8 < nsteps) { SCHUR_ISSUE(st + 16, rb) SCHUR_WAIT(10, rc) SCHUR_STEP(st + 8, rc) }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]),
                                        "+v"(rc[0]), "+v"(rc[1]), "+v"(rc[2]), "+v"(rc[3]), "+v"(rc[4]) : : "memory");   // the clamped over-reads land before their registers are reused
#undef SCHUR_ISSUE
#undef SCHUR_WAIT
#undef SCHUR_STEP
    for (int gsel = 0; gsel < 4; gsel++) {
        if (grp == gsel) {
#pragma unroll
            for (int i = 0; i < NP; i++) {       // subtract this group's partial with LDS fp64 atomics: no read-back latency; the groups take turns, so the order of the four subtractions is fixed
                const int ta = TA[PB + i], tb = TB[PB + i];
                double *T = s_T + tile_index(ta, tb) * 256;
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int rl = g4 + 4 * q, col = 16 * tb + c16;   // row inside the tile; tile row 4: rows 64, 65 | row 66 = rhs | row 67 = Cauchy column
                    if (col >= VB_NPOSE) continue;
                    if (ta < 4 || rl < 2) lds_add(&T[TIX(rl, c16)], -acc[i][q]);
                    else if (rl == 2) lds_add(&s_y[col], -acc[i][q]);
                    else if (rl == 3) lds_add(&s_t[col], -acc[i][q]);
                }
            }
        }
        __syncthreads();
    }
}

// dot(W~_f[0..65], vec) for every feature, 16 lanes per feature (4 features per wave per round), coalesced 80-wide rows;
// result per feature is written to out[f] (the caller ignores the entries of constant features). vec lives in LDS (>= 80 entries, 66.. = 0).
// Three rounds of rows in flight per wave, inline-asm loads with explicit vmcnt waits like the Schur reduce (plain loads are sunk to their uses
// by the compiler and every round then waits out a global-load latency).
__device__ __forceinline__ void feature_dots(const double *W, int F, const double *s_sv /*scale .* vec, 80 entries*/, double *out, int tid) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c16 = lane & 15, g = lane >> 4;
    double sv[5];
#pragma unroll
    for (int t5 = 0; t5 < 5; t5++) sv[t5] = s_sv[16 * t5 + c16];
    const int Fk4 = (F + 3) & ~3;                       // rows F..Fk4-1 of W are zero (never written)
    double ra[5], rb[5], rc[5];
#define FD_ISSUE(F0, R)                                                                                                            \
    {                                                                                                                              \
        const double *Wr_ = W + (size_t)min((F0) + g, Fk4 - 1) * VB_WLD + c16;                                                    \
        asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:128\n\t"                           \
                     "global_load_dwordx2 %2, %5, off offset:256\n\tglobal_load_dwordx2 %3, %5, off offset:384\n\t"                \
                     "global_load_dwordx2 %4, %5, off offset:512"                                                                  \
                     : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]) : "v"(Wr_) : "memory");                      \
    }
#define FD_WAIT(N, R) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]) : : "memory");
#define FD_STEP(F0, R)                                                                                                             \
    {                                                                                                                              \
        const int f = (F0) + g;                                                                                                    \
        double acc = 0;                                                                                                            \
        _Pragma("unroll") for (int t5 = 0; t5 < 5; t5++) acc += R[t5] * sv[t5];                                                    \
        if (!(f < F)) acc = 0;                                                                                                     \
        _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);                                           \
        if (c16 == 0 && f < F) out[f] = acc;                                                                                       \
    }
    constexpr int RS = 4 * SNW;                          // features per round of the workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    FD_ISSUE(4 * wave, ra) FD_ISSUE(4 * wave + RS, rb)
    for (int f0 = 4 * wave; f0 < F; f0 += 3 * RS) {
        FD_ISSUE(f0 + 2 * RS, rc) FD_WAIT(10, ra) FD_STEP(f0, ra)
        if (f0 + RS < F) { FD_ISSUE(f0 + 3 * RS, ra) FD_WAIT(10, rb) FD_STEP(f0 + RS, rb) }
        if (f0 + 2 * RS < F) { FD_ISSUE(f0 + 4 * RS, rb) FD_WAIT(10, rc) FD_STEP(f0 + 2 * RS, rc) }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]),
                                        "+v"(rc[0]), "+v"(rc[1]), "+v"(rc[2]), "+v"(rc[3]), "+v"(rc[4]) : : "memory");
#undef FD_ISSUE
#undef FD_WAIT
#undef FD_STEP
}

extern "C" __global__ __launch_bounds__(SNT) void k_solve(VbBatch b) {
    const int w = vb_window(b), tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (SGPR)
    if (w < 0) return;
    VbState *st = b.st + w;
    extern __shared__ double s_dyn[];
    double *s_T = s_dyn;                      // 66 tiles * 256
    double *s_g = s_T + 66 * 256;             // g~ (permuted, padded to 176)
    double *s_diag = s_g + VB_NPAD;
    double *s_scale = s_diag + VB_NPAD;
    double *s_y = s_scale + VB_NPAD;          // rhs -> solution
    double *s_invd = s_y + VB_NPAD;           // 1 / L_jj
    double *s_v = s_invd + VB_NPAD;           // v = g~ ./ diagonal_^2
    double *s_red = s_v + VB_NPAD;            // NT
    double *s_cf = s_red + SNT;                // per feature: s_f / sqrt(h~')   (0 for constant features)   [<= 1000]
    int *s_rng = (int *)(s_cf + VILF_MAX_FEATURES_DEV);   // per feature: 6*start | (6*(start+nobs)) << 16
    __shared__ int s_pcol[VB_P], s_pinv[VB_PRIOR_LD];
    __shared__ double s_dx[VB_PRIOR_LD];
    __shared__ int s_flag[4];

    // ---- iteration begin: FinalizeIterationAndCheckIfMinimizerCanContinue() of the previous iteration -------------
    if (tid == 0) {
        // the five state words in one round trip (read one after the other behind the short-circuit tests they were five dependent global loads)
        const int done = st->done, iteration = st->iteration, reuse = st->reuse;
        const double gmn = st->gradient_max_norm, radius = st->radius;
        int go = 1;
        if (done) go = 0;
        else if (iteration >= b.max_iterations) { st->done = 1; st->termination = 0; go = 0; }
        else if (gmn <= b.gradient_tolerance) { st->done = 1; st->termination = 3; go = 0; }
        else if (radius <= b.min_radius) { st->done = 1; st->termination = 4; go = 0; }
        if (go) st->iteration = iteration + 1;
        s_flag[0] = go;
        s_flag[1] = go ? reuse : 1;
    }
    __syncthreads();
    if (!s_flag[0] || s_flag[1]) return;     // done, or the previous Gauss-Newton step is re-used (rejected step)

    STAMP(1, 0);
    const int F = b.n_feat[w];
    const size_t FM = b.Fmax;
    const size_t ww = (size_t)st->ws * b.B + w;                             // the workspace that belongs to the current state x (k_linearize)
    const double *Hpp = b.Hpp + ww * 66 * 36;
    const double *imuH = b.imuH + ww * 9000, *lidH = b.lidH + ww * 1440;
    const double *priorH = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    double *W = b.W + ww * FM * VB_WLD;
    const double *hf = b.hf + ww * FM, *gf = b.gf + ww * FM;
    const uint8_t *f_const = b.f_const + (size_t)w * FM;
    const int *f_start = b.f_start + (size_t)w * FM, *f_nobs = b.f_nobs + (size_t)w * FM;
    double *scale_g = b.scale + (size_t)w * (VB_P + FM), *diag_g = b.diag + (size_t)w * (VB_P + FM);
    double *grad_g = b.grad + (size_t)w * (VB_P + FM), *gn_g = b.gn + (size_t)w * (VB_P + FM);
    const double *g_in = b.g + ww * VB_P;
    const int *phdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    const int pn = phdr[0] ? phdr[1] : 0;

    prior_setup(b, w, b.pose + (size_t)w * 77, b.sb + (size_t)w * 99, s_pcol, s_dx, tid);
    // inverse map: prior column -> permuted reduced index (-1: Ex_Pose columns, constant in the solve)
    if (tid < VB_PRIOR_LD) s_pinv[tid] = -1;
    __syncthreads();
    if (tid < VB_P) { const int pc = s_pcol[tid]; if (pc >= 0) { const int a = tid / 15, l = tid - 15 * a; s_pinv[pc] = perm_index(a, l); } }
    const int scaling_ready = st->scaling_ready;
    double mu = st->mu;
    int tries = 0;
    bool solved = false;
    double Jg2 = 0, G2 = 0;
    // ---- Jacobi scaling (computed once, iteration 0: 1/(1+sqrt(diag J^T J))), dogleg diagonal_, gradient_, v = g~ ./ diagonal_^2 --------
    double g2 = 0;
    if (tid < VB_NPAD) {
        double sc = 1.0, d = 1.0, gs = 0.0, vv = 0.0;
        if (tid < VB_P) {
            int a, l; unperm(tid, a, l);
            const double dh = b.diagH[ww * VB_P + 15 * a + l];
            if (scaling_ready) sc = scale_g[tid]; else { sc = 1.0 / (1.0 + sqrt(dh)); scale_g[tid] = sc; }
            d = sqrt(fmin(fmax(sc * sc * dh, b.min_lm_diagonal), b.max_lm_diagonal));
            gs = g_in[15 * a + l] * sc;
            diag_g[tid] = d;
            const double gr = gs / d;
            grad_g[tid] = gr;
            g2 = gr * gr;
            vv = gr / d;
        }
        s_scale[tid] = sc; s_diag[tid] = d; s_g[tid] = gs; s_v[tid] = vv;
    }
    // per-feature scalars (<= 2 features per thread): scale, diagonal, gradient -> global (re-read where they are needed)
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int f = tid + u * SNT;
        if (f < F) {
            s_rng[f] = (6 * f_start[f]) | ((6 * (f_start[f] + f_nobs[f])) << 16);
            if (!f_const[f]) {
                const double hfv = hf[f], gfv = gf[f];
                double sf;
                if (scaling_ready) sf = scale_g[VB_P + f]; else { sf = 1.0 / (1.0 + sqrt(hfv)); scale_g[VB_P + f] = sf; }
                const double d = sqrt(fmin(fmax(sf * sf * hfv, b.min_lm_diagonal), b.max_lm_diagonal));
                diag_g[VB_P + f] = d;
                const double gr = sf * gfv / d;
                grad_g[VB_P + f] = gr;
                g2 += gr * gr;
            }
        }
    }
    __syncthreads();
    STAMP(1, 1);
    for (;;) {
        // ---- streaming assembly of the SCALED H~ into 16x16 tiles; v^T H~_pp v accumulated on the fly (tries == 0) ------------
        double part = 0;
        for (int i = tid; i < 66 * 256; i += SNT) s_T[i] = 0.0;
        __syncthreads();
        if (tid < VB_NPAD - VB_P) s_T[tile_index(10, 10) * 256 + TIX(5 + tid, 5 + tid)] = 1.0;     // padding rows 165..175
        {                                                                                     // visual pose-pose blocks
            const int2 *lv = (const int2 *)b.lut_vis;
            for (int t0 = tid; t0 < 66 * 36; t0 += 5 * SNT) {
                double v[5]; int2 o[5];
#pragma unroll
                for (int u = 0; u < 5; u++) { const int t = min(t0 + u * SNT, 66 * 36 - 1); v[u] = Hpp[t]; o[u] = lv[t]; }
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    if (t0 + u * SNT >= 66 * 36) continue;
                    const int r = (o[u].x >> 15) & 255, c = (o[u].x >> 23) & 255;
                    const double val = v[u] * s_scale[r] * s_scale[c];
                    const int off = (o[u].x & 0x7fff) - 1;
                    if (off >= 0) lds_add(&s_T[off], val);
                    if (o[u].y > 0) lds_add(&s_T[o[u].y - 1], val);                                   // diagonal tiles hold both triangles
                    part += (((r / 6) != (c / 6)) ? 2.0 : 1.0) * s_v[r] * val * s_v[c];
                }
            }
        }
        __syncthreads();
        for (int par = 0; par < 2; par++) {                                                   // IMU + LiDAR factors k = par, par+2, ...
            for (int t0 = tid; t0 < 5 * 900; t0 += 5 * SNT) {
                double v[5]; int o[5];
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    const int t = min(t0 + u * SNT, 5 * 900 - 1);
                    const int k2 = t / 900, src = 900 * (2 * k2 + par) + (t - 900 * k2);
                    o[u] = b.lut_imu[src]; v[u] = imuH[src];
                }
#pragma unroll
                for (int u = 0; u < 5; u++) {
                    if (t0 + u * SNT >= 5 * 900) continue;
                    const int r = (o[u] >> 15) & 255, c = (o[u] >> 23) & 255;
                    const double val = v[u] * s_scale[r] * s_scale[c];
                    const int off = (o[u] & 0x7fff) - 1;
                    if (off >= 0) lds_add(&s_T[off], val);
                    part += s_v[r] * val * s_v[c];
                }
            }
            __syncthreads();                      // LiDAR factor k adds into the same pose entries as IMU factor k
            for (int t = tid; t < 5 * 144; t += SNT) {
                const int k2 = t / 144, src = 144 * (2 * k2 + par) + (t - 144 * k2);
                const int o0 = b.lut_lid[src];
                const int r = (o0 >> 15) & 255, c = (o0 >> 23) & 255;
                const double val = lidH[src] * s_scale[r] * s_scale[c];
                const int off = (o0 & 0x7fff) - 1;
                if (off >= 0) lds_add(&s_T[off], val);
                part += s_v[r] * val * s_v[c];
            }
            __syncthreads();
        }
        if (pn > 0) for (int t0 = tid; t0 < pn * pn; t0 += 6 * SNT) {                          // marginalization prior J0^T J0
            double v[6]; int rr[6], cc[6];
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int t = min(t0 + u * SNT, pn * pn - 1);
                const int i = t / pn, j = t - pn * i;
                rr[u] = s_pinv[i]; cc[u] = s_pinv[j]; v[u] = priorH[i * VB_PRIOR_LD + j];
            }
#pragma unroll
            for (int u = 0; u < 6; u++) {
                const int r = rr[u], c = cc[u];
                if (t0 + u * SNT >= pn * pn || r < 0 || c < 0) continue;
                const double val = v[u] * s_scale[r] * s_scale[c];
                if ((r >> 4) >= (c >> 4)) lds_add(&s_T[tile_index(r >> 4, c >> 4) * 256 + TIX(r & 15, c & 15)], val);
                part += s_v[r] * val * s_v[c];
            }
        }
        __syncthreads();
        STAMP(1, 2);
        if (tries == 0) {
            // Cauchy point: alpha = ||gradient_||^2 / || J~ (gradient_ ./ diagonal_) ||^2
            // v^T H~ v = v_p^T H~_pp v_p (accumulated above) + 2 sum_f v_f (w~_f . v_p) + sum_f h~_f v_f^2
            // The cross term 2 sum_p v_p S_p sum_f s_f v_f W_f[p] rides through the MFMA Schur reduce below instead of a pass of its own over W:
            // column 67 of U (padding of the 80-wide rows) is set to c_f x_f with c_f^2 x_f = s_f v_f, so (U^T U)[p][67] is exactly that inner sum.
#pragma unroll
            for (int u = 0; u < 2; u++) {          // the per-feature scalars are re-read (this thread wrote them above): nothing stays live across the assembly
                const int f = tid + u * SNT;
                if (f < F) {
                    double xf = 0.0;
                    if (!f_const[f]) {
                        const double sf = scale_g[VB_P + f], df = diag_g[VB_P + f], hfv = hf[f], gfv = gf[f];
                        const double vf = sf * gfv / (df * df);
                        part += sf * sf * hfv * vf * vf;
                        xf = vf * (sf * sf * hfv + mu * df * df) / sf;
                    }
                    W[(size_t)f * VB_WLD + VB_NPOSE + 1] = xf;
                }
            }
            __threadfence_block();                 // the MFMA loader of every wave of this workgroup reads column 67 back from global memory after the barriers below
                                                   // (workgroup scope: one CU, one L1 — an agent-scope fence writes back / invalidates L2 and doubled the kernel time)
            G2 = block_sum_s(g2, s_red);
            Jg2 = block_sum_s(part, s_red);
        }
        STAMP(1, 3);
        // ---- LM regularisation mu * diagonal_^2 on the reduced block; per-feature coefficients ----------------------
        if (tid < VB_P) { const int tt = tid >> 4, e = tid & 15; s_T[tile_index(tt, tt) * 256 + TIX(e, e)] += mu * s_diag[tid] * s_diag[tid]; }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int f = tid + u * SNT;
            if (f < F) { double c = 0; if (!f_const[f]) { const double sf = scale_g[VB_P + f], df = diag_g[VB_P + f]; c = sf * rsqrt_nr(sf * sf * hf[f] + mu * df * df); } s_cf[f] = c; }
        }
        if (tid < VB_NPAD) s_y[tid] = s_g[tid];
        __syncthreads();
        STAMP(1, 4);
        // ---- MFMA Schur reduce: H~_pp -= U^T U and rhs_p -= U^T t over the 5x5 pose tile block -------------------------
        for (int f = F + tid; f < ((F + 3) & ~3); f += SNT) s_cf[f] = 0.0;
        if (tid < 80) s_red[tid] = 0.0;           // row 67 of the reduce: -(U^T U)[p][67] (s_red is free between the block sums)
        __syncthreads();
        __threadfence_block();
        if (wave & 1) schur_mfma<1>(W, F, s_cf, s_scale, s_T, s_y, lane, wave); else schur_mfma<0>(W, F, s_cf, s_scale, s_T, s_y, lane, wave);
        if (tries == 0) {                         // every wave adds the cross term itself (same order in every wave: uniform and identical)
            double cr = ((lane < VB_NPOSE) ? s_v[lane] * s_red[lane] : 0.0) + ((lane + 64 < VB_NPOSE) ? s_v[lane + 64] * s_red[lane + 64] : 0.0);
            cr = vilf_wave_sum64(cr);
            Jg2 -= 2.0 * cr;
            __syncthreads();                      // s_red is reused by later block sums
        }
        STAMP(1, 5);
        // ---- blocked Cholesky (lower), 11 tile steps: TRSM (row per thread) -> MFMA trailing update, POTRF by wave 0 ------------
        // The reduced rhs rides along as row 165 (first padding row) of the lower-triangular storage: after the factorisation
        // L[165][0..164] = L^-1 rhs, i.e. the forward substitution is done by the TRSMs. Its diagonal is set large enough to
        // stay positive (it only has to exceed rhs^T S^-1 rhs).
        if (tid < VB_P) s_T[tile_index(10, tid >> 4) * 256 + TIX(5, tid & 15)] = s_y[tid];
        if (tid == 0) s_T[tile_index(10, 10) * 256 + TIX(5, 5)] = 1e250;
        __syncthreads();
        // Lookahead: after the TRSM of step k, column k+1 of the trailing matrix is updated first; then wave 0 factorises the
        // next diagonal tile while waves 1..7 update the remaining columns.
        bool ok = true;
        long long tc_a = TICK(), tc_trsm = 0, tc_p1 = 0, tc_p2 = 0, tc_potrf0 = 0;
        if (wave == 0) { bool o = potrf_tile_wave(s_T + tile_index(0, 0) * 256, s_invd, lane); if (lane == 0) s_flag[2] = o ? 1 : 0; }
        __syncthreads();
        { long long t = TICK(); tc_potrf0 = t - tc_a; tc_a = t; }
        for (int k = 0; k < VB_NTILE; k++) {
            if (!s_flag[2]) { ok = false; break; }
            const double *Tkk = s_T + tile_index(k, k) * 256;
            const int nrows = (VB_NTILE - 1 - k) * 16;
            if (tid < nrows) {
                const int ti = k + 1 + (tid >> 4), rr = tid & 15;
                double *Tik = s_T + tile_index(ti, k) * 256;
                double x[16];
#pragma unroll
                for (int c = 0; c < 16; c++) x[c] = Tik[TIX(rr, c)];
#pragma unroll
                for (int c = 0; c < 16; c++) {
                    double s = x[c];
#pragma unroll
                    for (int p = 0; p < c; p++) s -= x[p] * Tkk[TIX(c, p)];
                    x[c] = s * s_invd[16 * k + c];
                }
#pragma unroll
                for (int c = 0; c < 16; c++) Tik[TIX(rr, c)] = x[c];
            }
            __syncthreads();
            { long long t = TICK(); tc_trsm += t - tc_a; tc_a = t; }
            if (k == VB_NTILE - 1) break;
            const int nt = VB_NTILE - 1 - k;
            auto update_tile = [&](int ti, int tj) {
                double *C = s_T + tile_index(ti, tj) * 256;
                const double *A = s_T + tile_index(ti, k) * 256, *Bm = s_T + tile_index(tj, k) * 256;
                double4_t acc;
                double av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; q++) acc[q] = C[TIX((lane >> 4) + 4 * q, lane & 15)];
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) { av[s4] = -A[TIX(lane & 15, 4 * s4 + (lane >> 4))]; bv[s4] = Bm[TIX(lane & 15, 4 * s4 + (lane >> 4))]; }
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s4], bv[s4], acc, 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; q++) C[TIX((lane >> 4) + 4 * q, lane & 15)] = acc[q];
            };
            // phase 1: column k+1 (tiles (k+1+ii, k+1), ii = 0..nt-1)
            for (int ii = wave; ii < nt; ii += SNW) update_tile(k + 1 + ii, k + 1);
            __syncthreads();
            { long long t = TICK(); tc_p1 += t - tc_a; tc_a = t; }
            // phase 2: wave 0 factorises tile (k+1, k+1); the other waves update the remaining columns
            if (wave == 0) {
                bool o = potrf_tile_wave(s_T + tile_index(k + 1, k + 1) * 256, s_invd + 16 * (k + 1), lane);
                if (lane == 0) s_flag[2] = o ? 1 : 0;
            } else {
                const int nrest = (nt - 1) * nt / 2;            // tiles (ii, jj) with 1 <= jj <= ii <= nt-1
                for (int tt = wave - 1; tt < nrest; tt += SNW - 1) {
                    int ii = 0; while ((ii + 1) * (ii + 2) / 2 <= tt) ii++;
                    const int jj = tt - ii * (ii + 1) / 2;
                    update_tile(k + 2 + ii, k + 2 + jj);
                }
            }
            __syncthreads();
            { long long t = TICK(); tc_p2 += t - tc_a; tc_a = t; }
        }
        if (b.dbg && blockIdx.x == 0 && tid == 0) { b.dbg[64 + 20] = tc_potrf0; b.dbg[64 + 21] = tc_trsm; b.dbg[64 + 22] = tc_p1; b.dbg[64 + 23] = tc_p2; }
        STAMP(1, 6);
        tries++;
        if (ok) { solved = true; break; }
        mu *= 10.0;                                 // dogleg_strategy.cc: mu_ *= mu_increase_factor_
        if (!(mu < 1.0)) break;                     // max_mu_
        __syncthreads();
    }
    if (!solved) {
        if (tid == 0) { st->solve_failed = 1; st->mu = mu; st->num_linear_solves += tries; st->scaling_ready = 1; st->grad_sqnorm = G2; st->Jg2 = Jg2; }
        return;
    }
    // ---- forward / backward substitution by ONE wave, wave-synchronous (no block barriers): the rhs lives in s_y, each lane owns
    // rows {lane, lane+64, lane+128}; diagonal tiles are solved through SGPR broadcasts (v_readlane) ----------------------------------
    if (tid < VB_P) s_y[tid] = s_T[tile_index(10, tid >> 4) * 256 + TIX(5, tid & 15)];       // z = L^-1 rhs (row 165 of L)
    if (tid >= VB_P && tid < VB_NPAD) s_y[tid] = 0.0;
    __syncthreads();
    if (tid < 16) { if (tid == 5) s_T[tile_index(10, 10) * 256 + TIX(5, 5)] = 1.0; if (tid < 5) s_T[tile_index(10, 10) * 256 + TIX(5, tid)] = 0.0; }   // restore the padding row
    if (tid >= 16 && tid < 16 + 160) { const int c = tid - 16; s_T[tile_index(10, c >> 4) * 256 + TIX(5, c & 15)] = 0.0; }
    if (tid == 0) s_invd[165] = 1.0;
    __syncthreads();
    if (wave == 0) {
        const int l16 = lane & 15;
        for (int k = VB_NTILE - 1; k >= 0; k--) {       // L^T y = z
            const double *Tkk = s_T + tile_index(k, k) * 256;
            double Lc[16];
#pragma unroll
            for (int j = 0; j < 16; j++) Lc[j] = Tkk[TIX(j, l16)];
            double x = s_y[16 * k + l16];
            const double myinv = s_invd[16 * k + l16];
            double xs[16];
#pragma unroll
            for (int j = 15; j >= 0; j--) {
                if (l16 == j) x *= myinv;
                xs[j] = readlane_f64(x, j);
                if (l16 < j) x -= Lc[j] * xs[j];
            }
            if (lane < 16) s_y[16 * k + lane] = x;
            for (int col = lane; col < 16 * k; col += 64) {         // rows above: y_i -= sum_r L[16k+r][i] * y[16k+r]
                const double *T = s_T + tile_index(k, col >> 4) * 256;
                double sacc = 0;
#pragma unroll
                for (int r2 = 0; r2 < 16; r2++) sacc += T[TIX(r2, col & 15)] * xs[r2];
                s_y[col] -= sacc;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    STAMP(1, 7);
    // ---- back-substitute the features, Gauss-Newton step = -diagonal_ .* y, reductions ------------------------------
    double gy = 0, gn2 = 0;
    if (tid < VB_P) {
        const double y = s_y[tid], d = s_diag[tid];
        gn_g[tid] = -d * y;
        gy += s_g[tid] * y;
        gn2 += d * d * y * y;
    }
    if (tid < 80) s_v[tid] = (tid < VB_NPOSE) ? s_scale[tid] * s_y[tid] : 0.0;
    __syncthreads();
    feature_dots(W, F, s_v, s_cf, tid);                                                      // s_cf <- W_f . (S y)_p
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; u++) {
        const int f = tid + u * SNT;
        if (f >= F || f_const[f]) continue;
        const double sf = scale_g[VB_P + f], df = diag_g[VB_P + f];
        const double hp = sf * sf * hf[f] + mu * df * df;
        const double gt = sf * gf[f];
        const double y = (gt - sf * s_cf[f]) / hp;
        gn_g[VB_P + f] = -df * y;
        gy += gt * y;
        gn2 += df * df * y * y;
    }
    gy = block_sum_s(gy, s_red);
    gn2 = block_sum_s(gn2, s_red);
    if (tid == 0) {
        st->grad_sqnorm = G2; st->Jg2 = Jg2; st->alpha = G2 / Jg2;
        st->gy = gy; st->gn_sqnorm = gn2; st->mu = mu; st->mu_used = mu;
        st->num_linear_solves += tries; st->solve_failed = 0; st->reuse = 1; st->scaling_ready = 1;
    }
    STAMP(1, 8);
}

// ------------------------------------------------------------------------------------------------------------------
// k_solve_sb — the same linear solve as k_solve (DoglegStrategy::ComputeStep on the Jacobi-scaled DENSE_SCHUR system), with the
// speed-bias part eliminated FIRST. One 256-thread workgroup per window, 78 KB of LDS => two workgroups per CU.
//
//   H~ = [ dense (poses + SpeedBias[0], 75) | chain (SpeedBias[1..10], 10 x 9) ]: the chain part is block tridiagonal, its coupling to the dense part a
//   band (SpeedBias[a] <-> Pose[a-1], Pose[a], Pose[a+1]; SpeedBias[1] <-> SpeedBias[0]). A single window is latency-bound, so the phases are cut to
//   keep every dependent chain short and to run the two long ones side by side:
//   P1/P2  wave 3   : gathers the chain blocks D_a, E_a, then the block-tridiagonal Cholesky, newest block first:
//                     L_a = chol(D_a), B_a = L_a^-1 E_(a-1), D_(a-1) -= B_a^T B_a                      (the longest dependent chain of the kernel)
//          waves 0-2: gather assembly of the dense block and the band — one thread per DESTINATION entry sums its <= 6 source elements in a fixed order
//                     (the source index of every entry is arithmetic in the thread id; raw buffer loads, out of range = 0), scales, adds the LM term and stores: no atomics, no zero fill; v^T H~ v of the Cauchy point rides along —
//                     then the MFMA Schur reduce of the inverse depths, K split over the three waves (15 lower 16x16 tiles of the 80-wide dense block each)
//   P2b    the three partial U^T U are subtracted in wave order (fixed summation order); M_a = L_a^-1 and N_a = M_a B_(a+1)^T replace L_a / B_(a+1): every
//          later use of the chain is a product, not a substitution
//   P3     for a = 10..1: Y_a = M_a band_a - N_a Y_(a+1), all four waves, each owning a 16-column strip of the 76 band columns: the D layout of the fp64
//          16x16x4 MFMA (element q <-> row (lane>>4)+4q, column lane&15) IS the B-operand layout of the next product, so Y_(a+1) never leaves the
//          registers and the recurrence needs no barrier; Y_a^T Y_a accumulates into the wave's tiles of the dense block (rhs as row 75, Cauchy row 76)
//   P4     dense -= Y^T Y (batched read-subtract-write of the wave's tiles)
//   P5     Cholesky of the 75 + 2 dense rows with the trailing matrix resident in MFMA accumulator tiles: per 4-column panel the panel is extracted
//          to LDS, one thread per row factors the 4 x 4 diagonal block redundantly and solves its own strip, and the rank-4 update is ONE MFMA per
//          tile; the inverted diagonal blocks are kept for P6
//   P6     wave 0: block back substitution of the dense part, then the chain: u_a = M_a r_a - N_a u_(a+1), w_(a+1) = u_(a+1) - N_a^T w_a, x_a = M_a^T w_a;
//          waves 1-3: W_f . (S y)_p of every feature meanwhile
//   P7     feature back-substitution, Gauss-Newton step, dogleg scalars (as k_solve)
// Same arithmetic as a Cholesky of the whole system under another elimination order: results equal k_solve's to rounding.
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
// raw buffer load of one double: ONE address register per load (32-bit byte offset against an SGPR descriptor) instead of a 64-bit pair, and an offset past the
// end of the buffer returns zero — an absent source needs no clamped address and no select. elem < 0 means absent (offset far past num_records, no 32-bit wrap).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sb_rsrc(const double *p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void *)p, 0, (int)bytes, 0x00027000); }
__device__ __forceinline__ double sb_bload(__amdgpu_buffer_rsrc_t r, int elem) {
    const uint2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, elem < 0 ? 0x7ffffff0u : 8u * (unsigned)elem, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
#define SBT 256
#define SBW (SBT / 64)
__device__ __forceinline__ double block_sum_sb(double v, double *s_red) {
    const int tid = threadIdx.x;
    v = vilf_wave_sum64(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int k = 0; k < SBW; k++) r += s_red[k];
    __syncthreads();
    return r;
}
__device__ __forceinline__ void block_sum2_sb(double v0, double v1, double *s_red, double &r0, double &r1) {      // two sums in one pass (same order of additions as block_sum_sb)
    const int tid = threadIdx.x;
    v0 = vilf_wave_sum64(v0); v1 = vilf_wave_sum64(v1);
    __syncthreads();
    if ((tid & 63) == 0) { s_red[tid >> 6] = v0; s_red[8 + (tid >> 6)] = v1; }
    __syncthreads();
    r0 = 0; r1 = 0;
#pragma unroll
    for (int k = 0; k < SBW; k++) { r0 += s_red[k]; r1 += s_red[8 + k]; }
    __syncthreads();
}
__device__ __forceinline__ int sb_prow(int r) { return r * (r + 1) / 2; }
// band position of dense column c in step a (-1: structurally zero)
__device__ __forceinline__ int sb_band_pos(int a, int c) {
    if (c < VB_NPOSE) { const int p = c - 6 * (a - 1); return (p >= 0 && p < 18) ? p : -1; }
    if (c < SB_ND) return (a == 1) ? 18 + (c - VB_NPOSE) : -1;
    return (c == SB_ND) ? SB_BRHS(a) : -1;
}

// the 15 lower 16 x 16 tiles of the 80-wide dense block. Function-local constant tables: a static data member would be a device global, which the host may
// re-initialise (externally_initialized) — its loads are not folded after unrolling and every MFMA operand became a chain of selects.
__device__ __forceinline__ constexpr int sb_t15_a(int i) { constexpr int T[15] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4}; return T[i]; }
__device__ __forceinline__ constexpr int sb_t15_b(int i) { constexpr int T[15] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3, 0, 1, 2, 3, 4}; return T[i]; }
// P3: every wave owns <= 4 of those tiles and the 3 column tiles CT they touch; tile i = (CT[PA[i]], CT[PB[i]]):
//   wave 0: (0,0) (1,0) (1,1) (2,0)   wave 1: (2,1) (2,2) (3,1) (3,2)   wave 2: (3,0) (3,3) (4,0) (4,3)   wave 3: (4,1) (4,2) (4,4)
__device__ __forceinline__ constexpr int sb_y_ntl(int wv) { return wv == 3 ? 3 : 4; }
__device__ __forceinline__ constexpr int sb_y_ct(int wv, int k) { constexpr int T[4][3] = {{0, 1, 2}, {1, 2, 3}, {0, 3, 4}, {1, 2, 4}}; return T[wv][k]; }
__device__ __forceinline__ constexpr int sb_y_pa(int wv, int i) { constexpr int T[4][4] = {{0, 1, 1, 2}, {1, 1, 2, 2}, {1, 1, 2, 2}, {2, 2, 2, 2}}; return T[wv][i]; }
__device__ __forceinline__ constexpr int sb_y_pb(int wv, int i) { constexpr int T[4][4] = {{0, 0, 1, 0}, {0, 1, 0, 1}, {0, 1, 0, 1}, {0, 1, 2, 2}}; return T[wv][i]; }

// waves 0..2: U^T U of the feature rows, K split: wave wv takes the k-steps wv, wv + 3, ...
//   dense columns 0..65 = W columns 0..65 (poses); 66..74 (SpeedBias[0]) = 0; 75 (rhs) = W column 66 (g_f); 76 (Cauchy row) = W column 67
__device__ __forceinline__ void sb_feature_reduce(const double *W, const double *cf, int F, int wv, const double *s_scale, double4_t (&acc)[15], int lane) {
    const int c16 = lane & 15, g4 = lane >> 4;
    const int nsteps = ((F + 3) & ~3) / 4;
    if (wv >= nsteps) return;
    double sc5[5];
#pragma unroll
    for (int t5 = 0; t5 < 4; t5++) sc5[t5] = s_scale[16 * t5 + c16];
    sc5[4] = (c16 < 2) ? s_scale[64 + c16] : ((c16 == 11 || c16 == 12) ? 1.0 : 0.0);
    const int col4 = (c16 < 2) ? 64 + c16 : (c16 == 11 ? VB_NPOSE : (c16 == 12 ? VB_NPOSE + 1 : VB_NPOSE + 2));   // W column 68 is never written: zero
    // five k-steps of W in flight per wave (a global round trip is ~3 us here; the wave alone on its SIMD has nothing else to cover it)
    double r0[5], r1[5], r2[5], r3[5], r4[5], c0, c1, c2, c3, c4;
#define SBF_ISSUE(ST, R, C)                                                                                                        \
    {                                                                                                                              \
        const int row_ = 4 * min((ST), nsteps - 1) + g4;                                                                           \
        const double *p0_ = W + (size_t)row_ * VB_WLD + c16, *p4_ = W + (size_t)row_ * VB_WLD + col4, *pc_ = cf + row_;             \
        asm volatile("global_load_dwordx2 %0, %6, off\n\tglobal_load_dwordx2 %1, %6, off offset:128\n\t"                           \
                     "global_load_dwordx2 %2, %6, off offset:256\n\tglobal_load_dwordx2 %3, %6, off offset:384\n\t"                \
                     "global_load_dwordx2 %4, %7, off\n\tglobal_load_dwordx2 %5, %8, off"                                          \
                     : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(C) : "v"(p0_), "v"(p4_), "v"(pc_) : "memory"); \
    }
#define SBF_WAIT(N, R, C) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(C) : : "memory");
#define SBF_STEP(R, C)                                                                                                             \
    {                                                                                                                              \
        double u[5];                                                                                                               \
        _Pragma("unroll") for (int t5 = 0; t5 < 5; t5++) u[t5] = R[t5] * C * sc5[t5];                                              \
        _Pragma("unroll") for (int i = 0; i < 15; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(u[sb_t15_a(i)], u[sb_t15_b(i)], acc[i], 0, 0, 0); \
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SBF_ISSUE(wv, r0, c0) SBF_ISSUE(wv + 3, r1, c1) SBF_ISSUE(wv + 6, r2, c2) SBF_ISSUE(wv + 9, r3, c3)
    for (int st = wv; st < nsteps; st += 15) {
        SBF_ISSUE(st + 12, r4, c4) SBF_WAIT(24, r0, c0) SBF_STEP(r0, c0)
        if (st + 3 < nsteps) { SBF_ISSUE(st + 15, r0, c0) SBF_WAIT(24, r1, c1) SBF_STEP(r1, c1) }
        if (st + 6 < nsteps) { SBF_ISSUE(st + 18, r1, c1) SBF_WAIT(24, r2, c2) SBF_STEP(r2, c2) }
        if (st + 9 < nsteps) { SBF_ISSUE(st + 21, r2, c2) SBF_WAIT(24, r3, c3) SBF_STEP(r3, c3) }
        if (st + 12 < nsteps) { SBF_ISSUE(st + 24, r3, c3) SBF_WAIT(24, r4, c4) SBF_STEP(r4, c4) }
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(r0[3]), "+v"(r0[4]), "+v"(r1[0]), "+v"(r1[1]), "+v"(r1[2]), "+v"(r1[3]), "+v"(r1[4]),
                                        "+v"(r2[0]), "+v"(r2[1]), "+v"(r2[2]), "+v"(r2[3]), "+v"(r2[4]), "+v"(r3[0]), "+v"(r3[1]), "+v"(r3[2]), "+v"(r3[3]), "+v"(r3[4]),
                                        "+v"(r4[0]), "+v"(r4[1]), "+v"(r4[2]), "+v"(r4[3]), "+v"(r4[4]), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4) : : "memory");
#undef SBF_ISSUE
#undef SBF_WAIT
#undef SBF_STEP
}
// dense -= acc (rows <= 74, lower triangle), rhs row 75, Cauchy row 76 -> s_t, NTL tiles at a time: all reads, then the subtractions, then all writes — the entries of
// one call are distinct (one lane of one wave owns each), so nothing orders the loads behind the stores; written as `x -= a` per entry the compiler must assume aliasing
// and serialises the LDS round trips, and ds_add_f64 costs about as much per wave. An entry outside the stored triangle reads / writes a dump slot of its own.
template <int NTL, int PA, int PB>       // tile i = ((PA >> 4 i) & 15, (PB >> 4 i) & 15)
__device__ __forceinline__ void sb_acc_sub(double *s_P, double *s_t, const double4_t *a, int lane, bool cauchy, double *s_dump) {
    const int c16 = lane & 15, g4 = lane >> 4;
    double v[NTL][4]; int off[NTL][4];
#pragma unroll
    for (int i = 0; i < NTL; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int r = 16 * ((PA >> (4 * i)) & 15) + g4 + 4 * q, c = 16 * ((PB >> (4 * i)) & 15) + c16;
            int o = (int)(s_dump - s_P) + lane;                                          // not stored: a private dump slot (never read back as data)
            if (r < SB_NR && c <= r && c < SB_ND) o = sb_prow(r) + c;
            else if (cauchy && r == SB_NR && c < SB_ND) o = (int)(s_t - s_P) + c;
            off[i][q] = o;
            v[i][q] = s_P[o];
        }
#pragma unroll
    for (int i = 0; i < NTL; i++)
#pragma unroll
        for (int q = 0; q < 4; q++) s_P[off[i][q]] = v[i][q] - a[i][q];
}
// P3, one wave: Y_a = M_a band_a - N_a Y_(a+1) for a = 10..1 on the wave's three column tiles, entirely in registers — the MFMA output layout of
// T = N_a Y_(a+1) (lane (g4, c16), element q: row g4 + 4 q) IS the B-operand layout of the next product (k-step q: row 4 q + g4), so the recurrence needs
// no exchange and no barrier; Y_a^T Y_a of the wave's tiles accumulates alongside. M_a band_a and N_a are read-only in LDS.
template <int WV>
__device__ __forceinline__ void sb_y_chain(const double *s_band, const double *s_E, const double *s_zero, double4_t (&acc)[4], int lane) {
    const int c16 = lane & 15, g4 = lane >> 4;
    double4_t t[3];
#pragma unroll
    for (int ct = 0; ct < 3; ct++) t[ct] = double4_t{0, 0, 0, 0};
#pragma unroll 1
    for (int a = SB_NCH; a >= 1; a--) {
        const double *Ba = s_band + SB_BOFF(a);
        const int str = SB_BSTR(a);
        double yo[3][3];
#pragma unroll
        for (int ct = 0; ct < 3; ct++) {
            const int p = sb_band_pos(a, 16 * sb_y_ct(WV, ct) + c16);
#pragma unroll
            for (int ks = 0; ks < 3; ks++) {
                const int r = 4 * ks + g4;
                // (structural zeros are read from a zero in LDS: the select sits on the ADDRESS — a select on the loaded value is compiled into a branch around the
                //  read, every read followed by its own wait for LDS)
                const double v = *((r < 9 && p >= 0) ? Ba + r * str + p : s_zero);
                yo[ks][ct] = v - t[ct][ks];
            }
        }
        if (a > 1) {                                   // T of the next step first: the chain waits for it, the tile products below do not
            const double *Na = s_E + 81 * (a - 2);     // N_(a-1)
            double av[3];
#pragma unroll
            for (int ks = 0; ks < 3; ks++) { const int k = 4 * ks + g4; av[ks] = *((c16 < 9 && k < 9) ? Na + 9 * c16 + k : s_zero); }
#pragma unroll
            for (int ct = 0; ct < 3; ct++) t[ct] = double4_t{0, 0, 0, 0};
#pragma unroll
            for (int ks = 0; ks < 3; ks++)
#pragma unroll
                for (int ct = 0; ct < 3; ct++) t[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], yo[ks][ct], t[ct], 0, 0, 0);
        }
#pragma unroll
        for (int ks = 0; ks < 3; ks++)
#pragma unroll
            for (int i = 0; i < sb_y_ntl(WV); i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(yo[ks][sb_y_pa(WV, i)], yo[ks][sb_y_pb(WV, i)], acc[i], 0, 0, 0);
    }
}
template <int WV>
__device__ __forceinline__ void sb_y_store(const double4_t (&acc)[4], double *s_P, double *s_t, int lane, double *s_dump) {
    constexpr int PA = sb_y_ct(WV, sb_y_pa(WV, 0)) | (sb_y_ct(WV, sb_y_pa(WV, 1)) << 4) | (sb_y_ct(WV, sb_y_pa(WV, 2)) << 8) | (sb_y_ct(WV, sb_y_pa(WV, 3)) << 12);
    constexpr int PB = sb_y_ct(WV, sb_y_pb(WV, 0)) | (sb_y_ct(WV, sb_y_pb(WV, 1)) << 4) | (sb_y_ct(WV, sb_y_pb(WV, 2)) << 8) | (sb_y_ct(WV, sb_y_pb(WV, 3)) << 12);
    sb_acc_sub<sb_y_ntl(WV), PA, PB>(s_P, s_t, &acc[0], lane, false, s_dump);
}

// wave 3: 9 x 9 lower Cholesky by lanes 0..8 (lane r = row r in registers, pivots broadcast through SGPRs), result written back, 1 / L_ii to linv
__device__ __forceinline__ bool sb_potrf9(double *Dp, double *linv, int lane) {
    const int r = min(lane, 8);
    double v[9];
#pragma unroll
    for (int c = 0; c < 9; c++) v[c] = Dp[9 * r + c];
    bool ok = true;
    double myinv = 1.0;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const double djj = readlane_f64(v[j], j);
        if (!(djj > 0.0)) ok = false;
        const double inv = rsqrt_h3(djj);
        const double lrj = v[j] * inv;
        v[j] = lrj;
        if (r == j) myinv = inv;
#pragma unroll
        for (int c = j + 1; c < 9; c++) v[c] -= lrj * readlane_f64(lrj, c);
    }
    if (lane < 9) {
#pragma unroll
        for (int c = 0; c < 9; c++) if (c <= r) Dp[9 * r + c] = v[c];
        linv[r] = myinv;
    }
    return ok;
}
// one gather entry of the chain / band tables: (meta, src0, src1, -) -> scaled value + LM term, Cauchy-point contribution
// ---- the gather's index table --------------------------------------------------------------------------------------------------------------------------
// Which sources a destination entry of the dense block / band / chain blocks adds up is the same for every window: it is decoded ONCE (k_sb_table, at vilf_create) into
// a table the solve reads with coalesced 16-byte loads — 23 per thread, always L2 hits — instead of ~2000 integer instructions per thread and launch (divisions by
// 36 / 81 / 243 / 27, a triangular root: 35 k of the 53 k cycles of the gather at one VALU instruction per ~12 cycles and wave). Layout: int4 rows [k][256 threads];
// waves 0..2, A part: 4 x (meta, Hpp, imu0, imu1, lid0, lid1), 9 x (meta, Hpp) in rows 0..10; B part: 4 x (meta, imu), 13 x (meta, imu0, imu1) in rows 11..22;
// wave 3 (its 64 lanes at threads 192..255): 13 x (meta, imu0, imu1), 12 x (meta, imu) in rows 0..15. meta = LDS offset | r << 14 | c << 22, bit 31 = "no such entry"
// (r, c stay valid: the prior's column lookup is issued for every entry); a source index of -1 = absent (the buffer load returns zero).
#define SB_TAB_ROWS 23
#define SB_META(off, r, c, on) (((off) & 0x3fff) | ((r) << 14) | ((c) << 22) | ((on) ? 0 : (int)0x80000000))
extern "C" __global__ __launch_bounds__(256) void k_sb_table(int *tab) {
    const int tid = threadIdx.x, td = tid, ln = tid & 63;
    int t[SB_TAB_ROWS * 4];
    for (int k = 0; k < SB_TAB_ROWS * 4; k++) t[k] = -1;
    if (tid >= 192) {
        for (int u = 0; u < 13; u++) {          // D_a[i][j], j <= i: IMU factor a-1 rows 21.. (+ factor a rows 6..)
            const int q = min(ln + 64 * u, 809), a1 = q / 81, rem = q - 81 * a1, i = rem / 9, j = rem - 9 * i;
            const bool on = ln + 64 * u < 810 && j <= i;
            t[3 * u] = SB_META(SB_OFF_D + q, VB_NPOSE + 9 + 9 * a1 + i, VB_NPOSE + 9 + 9 * a1 + j, on);
            t[3 * u + 1] = 900 * a1 + 30 * (21 + i) + 21 + j;
            t[3 * u + 2] = a1 + 1 <= 9 ? 900 * (a1 + 1) + 30 * (6 + i) + 6 + j : -1;
        }
        for (int u = 0; u < 12; u++) {          // E_a[i][j] = H(SpeedBias[a+1], SpeedBias[a]): IMU factor a, rows 21.., columns 6..
            const int q = min(ln + 64 * u, 728), a1 = q / 81, rem = q - 81 * a1, i = rem / 9, j = rem - 9 * i;
            t[39 + 2 * u] = SB_META(SB_OFF_E + q, VB_NPOSE + 18 + 9 * a1 + i, VB_NPOSE + 9 + 9 * a1 + j, ln + 64 * u < 729);
            t[39 + 2 * u + 1] = 900 * (a1 + 1) + 30 * (21 + i) + 6 + j;
        }
    } else {
        for (int u = 0; u < 4; u++) {           // pose blocks on / next to the diagonal (21 blocks): visual + IMU (<= 2) + LiDAR (<= 2) + prior
            const int q = min(td + 192 * u, 755), nb = q / 36, e = q - 36 * nb, l1 = e / 6, l2 = e - 6 * l1;
            const bool dg = nb < 11;
            const int A = dg ? nb : nb - 10, Bf = dg ? nb : nb - 11, r = 6 * A + l1, c = 6 * Bf + l2;
            const int i0 = dg ? 900 * (A - 1) + 30 * (15 + l1) + 15 + l2 : 900 * Bf + 30 * (15 + l1) + l2, i1i = 900 * A + 30 * l1 + l2;
            const int j0 = dg ? 144 * (A - 1) + 12 * (6 + l1) + 6 + l2 : 144 * Bf + 12 * (6 + l1) + l2, j1 = 144 * A + 12 * l1 + l2;
            const bool h0 = !dg || A >= 1, h1 = dg && A <= 9;
            t[6 * u] = SB_META(SB_OFF_P + sb_prow(r) + c, r, c, td + 192 * u < 756 && c <= r);
            t[6 * u + 1] = 36 * (A * (A + 1) / 2 + Bf) + e;
            t[6 * u + 2] = h0 ? i0 : -1; t[6 * u + 3] = h1 ? i1i : -1; t[6 * u + 4] = h0 ? j0 : -1; t[6 * u + 5] = h1 ? j1 : -1;
        }
        for (int u = 0; u < 9; u++) {           // the other 45 pose blocks: visual + prior
            const int q = min(td + 192 * u, 1619), fb = q / 36, e = q - 36 * fb, l1 = e / 6, l2 = e - 6 * l1;
            int A2 = 0; while ((A2 + 1) * (A2 + 2) / 2 <= fb) A2++;
            const int Bf = fb - A2 * (A2 + 1) / 2, A = A2 + 2, r = 6 * A + l1, c = 6 * Bf + l2;
            t[24 + 2 * u] = SB_META(SB_OFF_P + sb_prow(r) + c, r, c, td + 192 * u < 1620);
            t[24 + 2 * u + 1] = 36 * (A * (A + 1) / 2 + Bf) + e;
        }
        for (int u = 0; u < 4; u++) {           // SpeedBias[0] rows of the dense block: IMU factor 0 (rows 6 + i) + prior
            const int q = min(td + 192 * u, 674), i = q / 75, c = q - 75 * i, r = VB_NPOSE + i, Bf = c / 6, l2 = c - 6 * Bf;
            const int src = (c < VB_NPOSE) ? 30 * (6 + i) + (Bf == 0 ? l2 : 15 + l2) : 30 * (6 + i) + 6 + (c - VB_NPOSE);
            const bool hs = c >= VB_NPOSE || Bf <= 1;
            t[44 + 2 * u] = SB_META(SB_OFF_P + sb_prow(r) + c, r, c, td + 192 * u < 675 && c <= r);
            t[44 + 2 * u + 1] = hs ? src : -1;
        }
        for (int u = 0; u < 13; u++) {          // band_a[i][pos]: SpeedBias[a] x (Pose a-1 | Pose a | Pose a+1 | SpeedBias[0] for a = 1)
            const int q = min(td + 192 * u, 2429), a1 = q / 243, rem = q - 243 * a1, i = rem / 27, pos = rem - 27 * i, a = a1 + 1;
            const int seg = pos / 6, m = pos - 6 * seg;            // seg 0: Pose a-1, 1: Pose a, 2: Pose a+1, 3..4: SpeedBias[0]
            const bool on = td + 192 * u < 2430 && (seg < 2 || (seg == 2 && a <= 9) || (seg >= 3 && a == 1));
            int s0, s1 = -1, c165;
            if (seg == 0) { s0 = 900 * a1 + 30 * (21 + i) + m; c165 = 6 * a1 + m; }
            else if (seg == 1) { s0 = 900 * a1 + 30 * (21 + i) + 15 + m; s1 = (a <= 9) ? 900 * a + 30 * (6 + i) + m : -1; c165 = 6 * a + m; }
            else if (seg == 2) { s0 = 900 * min(a, 9) + 30 * (6 + i) + 15 + m; c165 = 6 * (a + 1) + m; }
            else { s0 = 30 * (21 + i) + 6 + (pos - 18); c165 = VB_NPOSE + (pos - 18); }
            t[52 + 3 * u] = SB_META(SB_OFF_BAND + SB_BOFF(a) + i * SB_BSTR(a) + pos, VB_NPOSE + 9 * a + i, c165, on);
            t[52 + 3 * u + 1] = on ? s0 : -1;
            t[52 + 3 * u + 2] = s1;
        }
    }
    for (int k = 0; k < SB_TAB_ROWS; k++) for (int j = 0; j < 4; j++) tab[((size_t)k * 256 + tid) * 4 + j] = t[4 * k + j];
}

template <bool FUSED>
__device__ __forceinline__ void solve_sb_body(const VbBatch &b, int w, size_t ww_in, int tid_in = 0) {
    const int tid = FUSED ? tid_in : (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (w < 0) return;
    VbState *st = b.st + w;
    extern __shared__ double s_dyn[];
    double *s_P = s_dyn + SB_OFF_P, *s_D = s_dyn + SB_OFF_D, *s_E = s_dyn + SB_OFF_E, *s_band = s_dyn + SB_OFF_BAND;
    double *s_g = s_dyn + SB_OFF_VEC, *s_diag = s_g + SB_VLD, *s_scale = s_diag + SB_VLD, *s_y = s_scale + SB_VLD, *s_v = s_y + SB_VLD;
    double *s_t = s_dyn + SB_OFF_T, *s_linv = s_dyn + SB_OFF_LINV, *s_dinv = s_dyn + SB_OFF_DINV, *s_pan = s_dyn + SB_OFF_PAN, *s_u = s_dyn + SB_OFF_U, *s_w = s_dyn + SB_OFF_W;
    double *s_red = s_dyn + SB_OFF_RED;
    __shared__ int s_pcp[VB_P];        // permuted reduced index -> prior column (-1: not in the prior)
    __shared__ int s_flag[4];

    if (tid == 0) {     // FinalizeIterationAndCheckIfMinimizerCanContinue() of the previous iteration (as k_solve)
        const int done = st->done, iteration = st->iteration, reuse = st->reuse;
        const double gmn = st->gradient_max_norm, radius = st->radius;
        int go = 1;
        if (done) go = 0;
        else if (iteration >= b.max_iterations) { st->done = 1; st->termination = 0; go = 0; }
        else if (gmn <= b.gradient_tolerance) { st->done = 1; st->termination = 3; go = 0; }
        else if (radius <= b.min_radius) { st->done = 1; st->termination = 4; go = 0; }
        if (go) st->iteration = iteration + 1;
        s_flag[0] = go;
        s_flag[1] = go ? reuse : 1;
    }
    __syncthreads();
    if (!s_flag[0] || s_flag[1]) return;

    STAMP(1, 0);
    const int F = b.n_feat[w];
    const size_t FM = b.Fmax;
    const size_t ww = FUSED ? ww_in : (size_t)st->ws * b.B + w;            // the workspace that belongs to the current state x (k_linearize; FUSED: the workgroup's slot)
    const double *Hpp = b.Hpp + ww * 66 * 36;
    const double *imuH = b.imuH + ww * 9000, *lidH = b.lidH + ww * 1440;
    const double *priorH = b.prior_H + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD;
    double *W = b.W + ww * FM * VB_WLD;
    double *cf = b.cf + (FUSED ? ww : (size_t)w) * FM;
    const double *hf = b.hf + ww * FM, *gf = b.gf + ww * FM;
    const uint8_t *f_const = b.f_const + (size_t)w * FM;
    double *scale_g = b.scale + (size_t)w * (VB_P + FM), *diag_g = b.diag + (size_t)w * (VB_P + FM);
    double *grad_g = b.grad + (size_t)w * (VB_P + FM), *gn_g = b.gn + (size_t)w * (VB_P + FM);
    const double *g_in = b.g + ww * VB_P;
    const int *phdr = b.prior_hdr + (size_t)w * VB_PRIOR_HDR;
    const bool has_prior = phdr[0] != 0;
    const __amdgpu_buffer_rsrc_t rHpp = sb_rsrc(Hpp, 66 * 36 * 8), rImu = sb_rsrc(imuH, 9000 * 8), rLid = sb_rsrc(lidH, 1440 * 8), rPri = sb_rsrc(priorH, VB_PRIOR_LD * VB_PRIOR_LD * 8);

    // ---- set-up: prior column map, zero fill (padding entries stay zero for the whole launch), scaling vectors --------------------------------
    for (int i = tid; i < SB_NCH * 81; i += SBT) s_D[i] = 0.0;              // upper triangles of D_a (read by the row-wise factorisation), band padding, Cauchy row
    for (int i = tid; i < 252 + 180 * (SB_NCH - 1); i += SBT) s_band[i] = 0.0;
    if (tid < 80) s_t[tid] = 0.0;
    if (tid == 0) s_P[sb_prow(SB_ND) + SB_ND] = 0.0;
    if (tid < VB_P) s_pcp[tid] = -1;
    __syncthreads();
    if (has_prior && tid < phdr[2]) {
        const int id = phdr[3 + tid], idx = phdr[51 + tid];
        if (id < VB_NF) { for (int k = 0; k < 6; k++) s_pcp[6 * id + k] = idx + k; }
        else if (id == VB_NF) { for (int k = 0; k < 9; k++) s_pcp[VB_NPOSE + k] = idx + k; }     // SpeedBias[0]; other speed-bias blocks: k_solve (host-selected)
    }
    const int scaling_ready = st->scaling_ready;
    double mu = st->mu;
    int tries = 0;
    bool solved = false;
    double Jg2 = 0, G2 = 0, g2 = 0;
    if (tid < VB_P) {
        int a, l; unperm(tid, a, l);
        const double dh = b.diagH[ww * VB_P + 15 * a + l];
        double sc;
        if (scaling_ready) sc = scale_g[tid]; else { sc = 1.0 / (1.0 + sqrt(dh)); scale_g[tid] = sc; }
        const double d = sqrt(fmin(fmax(sc * sc * dh, b.min_lm_diagonal), b.max_lm_diagonal));
        const double gs = g_in[15 * a + l] * sc;
        diag_g[tid] = d;
        const double gr = gs / d;
        grad_g[tid] = gr;
        g2 = gr * gr;
        s_scale[tid] = sc; s_diag[tid] = d; s_g[tid] = gs; s_v[tid] = gr / d;
    }
    for (int f = tid; f < F; f += SBT) {
        if (f_const[f]) continue;
        const double hfv = hf[f], gfv = gf[f];
        double sf;
        if (scaling_ready) sf = scale_g[VB_P + f]; else { sf = 1.0 / (1.0 + sqrt(hfv)); scale_g[VB_P + f] = sf; }
        const double d = sqrt(fmin(fmax(sf * sf * hfv, b.min_lm_diagonal), b.max_lm_diagonal));
        diag_g[VB_P + f] = d;
        const double gr = sf * gfv / d;
        grad_g[VB_P + f] = gr;
        g2 += gr * gr;
    }
    __threadfence_block();
    __syncthreads();
    // thread -> 4 x 4 block of the dense lower triangle (19 block rows, 190 blocks)
    int blk_i = 0;
    while ((blk_i + 1) * (blk_i + 2) / 2 <= tid) blk_i++;
    const int blk_j = tid - blk_i * (blk_i + 1) / 2;

    STAMP(1, 1);
    for (;;) {
        // lane-derived predicates and addresses of the phases below are loop invariant; hoisted out of this (rarely repeated) loop they would pin ~100 registers
        int ln = lane, bi = blk_i, bjm = blk_j, td = tid;
        asm volatile("" : "+v"(ln), "+v"(bi), "+v"(bjm), "+v"(td));
        double part = 0;
        if (td == 0) s_flag[2] = 1;
        // ---- P1 / P2 --------------------------------------------------------------------------------------------------------------------------------
        if (wave == 3) {
            // chain blocks D_a (lower) and E_a straight from the IMU blocks: every destination entry decodes its own source indices (no table: a table read
            // would be one more dependent global round trip, ~3 us each here), all loads of the lane in flight at once
            {
                double sv[25][2]; int meta[25];
                int ti[64];
                {
                    const int4 *tp = reinterpret_cast<const int4 *>(b.sb_tab) + td;
#pragma unroll
                    for (int k = 0; k < 16; k++) { const int4 v = tp[k * 256]; ti[4 * k] = v.x; ti[4 * k + 1] = v.y; ti[4 * k + 2] = v.z; ti[4 * k + 3] = v.w; }
                }
#pragma unroll
                for (int u = 0; u < 13; u++) {          // D_a[i][j], j <= i: IMU factor a-1 rows 21.. (+ factor a rows 6..)
                    sv[u][0] = sb_bload(rImu, ti[3 * u + 1]);
                    sv[u][1] = sb_bload(rImu, ti[3 * u + 2]);
                    meta[u] = ti[3 * u];
                }
#pragma unroll
                for (int u = 0; u < 12; u++) {          // E_a[i][j] = H(SpeedBias[a+1], SpeedBias[a]): IMU factor a, rows 21.., columns 6..
                    sv[13 + u][0] = sb_bload(rImu, ti[39 + 2 * u + 1]);
                    sv[13 + u][1] = 0.0;
                    meta[13 + u] = ti[39 + 2 * u];
                }
#pragma unroll
                for (int u = 0; u < 25; u++) {
                    if (meta[u] < 0) continue;
                    const int r = (meta[u] >> 14) & 255, c = (meta[u] >> 22) & 255;
                    const double val = (sv[u][0] + sv[u][1]) * s_scale[r] * s_scale[c];
                    part += ((r != c) ? 2.0 : 1.0) * s_v[r] * val * s_v[c];
                    s_dyn[meta[u] & 0x3fff] = val + ((r == c) ? mu : 0.0) * s_diag[r] * s_diag[r];        // (+ 0 off the diagonal: the same bits, and no branch around the s_diag reads)
                }
            }
            bool ok = true;
#pragma unroll 1
            for (int a = SB_NCH; a >= 1; a--) {
                double *Da = s_D + 81 * (a - 1);
                ok = sb_potrf9(Da, s_linv + 9 * (a - 1), ln) && ok;
                if (a == 1) break;
                // B_a = L_a^-1 E_(a-1): lane c owns column c
                double *Ea = s_E + 81 * (a - 2);
                const int c = min(ln, 8);
                double x[9];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    double s = Ea[9 * i + c];
#pragma unroll
                    for (int j = 0; j < i; j++) s -= Da[9 * i + j] * x[j];
                    x[i] = s * s_linv[9 * (a - 1) + i];
                }
                if (ln < 9) {
#pragma unroll
                    for (int i = 0; i < 9; i++) Ea[9 * i + c] = x[i];
                }
                // D_(a-1) -= B_a^T B_a (lower triangle): lane -> (r, cc), r >= cc
                if (ln < 45) {
                    int r = 0; while ((r + 1) * (r + 2) / 2 <= ln) r++;
                    const int cc = ln - r * (r + 1) / 2;
                    double s = 0;
#pragma unroll
                    for (int i = 0; i < 9; i++) s += Ea[9 * i + r] * Ea[9 * i + cc];
                    s_D[81 * (a - 2) + 9 * r + cc] -= s;
                }
            }
            if (!ok && ln == 0) s_flag[2] = 0;
        } else {
            // dense block and band: gather assembly by the 192 threads of waves 0..2. One thread per DESTINATION entry; which sources it adds up comes from the
            // index table (k_sb_table: coalesced 16-byte loads, L2 hits; the part-B rows are requested before part A's sources so that they arrive under them),
            // every load of the thread in flight at once.
            // (requesting the rows before the set-up in front of the loop, to have them arrive under it, made the compiler hold all 92 registers across the
            //  factorisation: 151 spills)
            int tb[48];
            {
                constexpr int KN = 4, KF = 9, KT = KN + KF;
                double sn[KN][6], sf[KF][2];
                int meta[KT];
                int ti[44];
                {
                    const int4 *tp = reinterpret_cast<const int4 *>(b.sb_tab) + td;
#pragma unroll
                    for (int k = 0; k < 11; k++) { const int4 v = tp[k * 256]; ti[4 * k] = v.x; ti[4 * k + 1] = v.y; ti[4 * k + 2] = v.z; ti[4 * k + 3] = v.w; }
#pragma unroll
                    for (int k = 0; k < 12; k++) { const int4 v = tp[(11 + k) * 256]; tb[4 * k] = v.x; tb[4 * k + 1] = v.y; tb[4 * k + 2] = v.z; tb[4 * k + 3] = v.w; }
                }
#pragma unroll
                for (int u = 0; u < KN; u++) {           // pose blocks on / next to the diagonal (21 blocks): visual + IMU (<= 2) + LiDAR (<= 2) + prior
                    const int m = ti[6 * u], pr = s_pcp[(m >> 14) & 255], pc = s_pcp[(m >> 22) & 255];
                    sn[u][0] = sb_bload(rHpp, ti[6 * u + 1]);
                    sn[u][1] = sb_bload(rImu, ti[6 * u + 2]); sn[u][2] = sb_bload(rImu, ti[6 * u + 3]); sn[u][3] = sb_bload(rLid, ti[6 * u + 4]); sn[u][4] = sb_bload(rLid, ti[6 * u + 5]);
                    sn[u][5] = sb_bload(rPri, (pr >= 0 && pc >= 0) ? pr * VB_PRIOR_LD + pc : -1);
                    meta[u] = m;
                }
#pragma unroll
                for (int u = 0; u < KF; u++) {           // the other 45 pose blocks: visual + prior
                    const int m = ti[24 + 2 * u], pr = s_pcp[(m >> 14) & 255], pc = s_pcp[(m >> 22) & 255];
                    sf[u][0] = sb_bload(rHpp, ti[24 + 2 * u + 1]);
                    sf[u][1] = sb_bload(rPri, (pr >= 0 && pc >= 0) ? pr * VB_PRIOR_LD + pc : -1);
                    meta[KN + u] = m;
                }
                STAMP(1, 20);
#pragma unroll
                for (int u = 0; u < KT; u++) {
                    if (meta[u] < 0) continue;
                    const int r = (meta[u] >> 14) & 255, c = (meta[u] >> 22) & 255;
                    double raw;
                    if (u < KN) raw = ((((sn[u][0] + sn[u][1]) + sn[u][2]) + sn[u][3]) + sn[u][4]) + sn[u][5];
                    else raw = sf[u - KN][0] + sf[u - KN][1];
                    const double val = raw * s_scale[r] * s_scale[c];
                    part += ((r != c) ? 2.0 : 1.0) * s_v[r] * val * s_v[c];
                    s_dyn[meta[u] & 0x3fff] = val + ((r == c) ? mu : 0.0) * s_diag[r] * s_diag[r];        // (+ 0 off the diagonal: the same bits, and no branch around the s_diag reads)
                }
            }
            STAMP(1, 21);
            {
                constexpr int KS = 4, KB2 = 13, KT = KS + KB2;
                double ss[KS][2], sb3[KB2][2];
                int meta[KT];
#pragma unroll
                for (int u = 0; u < KS; u++) {           // SpeedBias[0] rows of the dense block: IMU factor 0 (rows 6 + i) + prior
                    const int m = tb[2 * u], pr = s_pcp[(m >> 14) & 255], pc = s_pcp[(m >> 22) & 255];
                    ss[u][0] = sb_bload(rImu, tb[2 * u + 1]);
                    ss[u][1] = sb_bload(rPri, (pr >= 0 && pc >= 0) ? pr * VB_PRIOR_LD + pc : -1);
                    meta[u] = m;
                }
#pragma unroll
                for (int u = 0; u < KB2; u++) {          // band_a[i][pos]: SpeedBias[a] x (Pose a-1 | Pose a | Pose a+1 | SpeedBias[0] for a = 1)
                    sb3[u][0] = sb_bload(rImu, tb[8 + 3 * u + 1]);
                    sb3[u][1] = sb_bload(rImu, tb[8 + 3 * u + 2]);
                    meta[KS + u] = tb[8 + 3 * u];
                }
                STAMP(1, 22);
#pragma unroll
                for (int u = 0; u < KT; u++) {
                    if (meta[u] < 0) continue;
                    const int r = (meta[u] >> 14) & 255, c = (meta[u] >> 22) & 255;
                    const double raw = (u < KS) ? ss[u < KS ? u : 0][0] + ss[u < KS ? u : 0][1] : sb3[u >= KS ? u - KS : 0][0] + sb3[u >= KS ? u - KS : 0][1];
                    const double val = raw * s_scale[r] * s_scale[c];
                    part += ((r != c) ? 2.0 : 1.0) * s_v[r] * val * s_v[c];
                    s_dyn[meta[u] & 0x3fff] = val + ((r == c) ? mu : 0.0) * s_diag[r] * s_diag[r];        // (+ 0 off the diagonal: the same bits, and no branch around the s_diag reads)
                }
            }
            STAMP(1, 23);
            // right-hand sides: dense row 75, band rhs column
            if (td < SB_ND) s_P[sb_prow(SB_ND) + td] = s_g[td];
            if (td >= 96 && td < 96 + 9 * SB_NCH) { const int q = td - 96, a = q / 9 + 1, i = q - 9 * (a - 1); s_band[SB_BOFF(a) + i * SB_BSTR(a) + SB_BRHS(a)] = s_g[VB_NPOSE + 9 + q]; }
            // per-feature Schur coefficient and (first try) the Cauchy-point terms. Every wave writes ALL features (identical values): each wave then reads
            // back what it wrote itself — no cross-wave synchronisation while wave 3 runs the chain
            for (int f = ln; f < ((F + 3) & ~3); f += 64) {
                double c = 0.0, xf = 0.0;
                if (f < F && !f_const[f]) {
                    const double sf = scale_g[VB_P + f], df = diag_g[VB_P + f], hfv = hf[f], gfv = gf[f];
                    const double hp = sf * sf * hfv + mu * df * df;
                    c = sf * rsqrt_nr(hp);
                    if (tries == 0) {
                        const double vf = sf * gfv / (df * df);
                        if (wave == 0) part += sf * sf * hfv * vf * vf;
                        xf = vf * hp / sf;              // column 67 of W: c_f x_f with c_f^2 x_f = s_f v_f  =>  (U^T U)[p][76] = sum_f s_f v_f W~_f[p]
                    }
                }
                cf[f] = c;
                if (tries == 0 && f < F) W[(size_t)f * VB_WLD + VB_NPOSE + 1] = xf;
            }
            __threadfence_block();
        }
        STAMP(1, 2);
        STAMP(1, 24);
        double4_t acc15[15];
#pragma unroll
        for (int i = 0; i < 15; i++) acc15[i] = double4_t{0, 0, 0, 0};
        if (wave < 3) sb_feature_reduce(W, cf, F, wave, s_scale, acc15, ln);
#ifdef VILF_STAMPS
        if (b.dbg && blockIdx.x == 0 && (tid & 63) == 0) b.dbg[32 + 16 + (tid >> 6)] = __builtin_readcyclecounter();
#endif
        __syncthreads();
        STAMP(1, 3);
        // ---- P2b: dense -= U^T U, the three K partials in wave order (fixed order of the additions) ----------------------------------------------------
        // A wave's fifteen tiles go in five groups of three (tile order of sb_t15_a / sb_t15_b); wave w subtracts group g in phase g + w: every entry still sees wave 0,
        // then 1, then 2, but the waves work at the same time on different groups — seven short phases instead of three turns of a whole wave each (round 5: the turns
        // were 26 k of the kernel's 224 k cycles, one SIMD at work, the read-modify-write of a group being LDS latency + index arithmetic, not LDS bandwidth).
#define SB_P2B_GROUP(G) do { \
            if ((G) == 0) sb_acc_sub<3, 0x110, 0x100>(s_P, s_t, &acc15[0], ln, true, s_pan);       /* (0,0) (1,0) (1,1) */ \
            else if ((G) == 1) sb_acc_sub<3, 0x222, 0x210>(s_P, s_t, &acc15[3], ln, true, s_pan);  /* (2,0) (2,1) (2,2) */ \
            else if ((G) == 2) sb_acc_sub<3, 0x333, 0x210>(s_P, s_t, &acc15[6], ln, true, s_pan);  /* (3,0) (3,1) (3,2) */ \
            else if ((G) == 3) sb_acc_sub<3, 0x443, 0x103>(s_P, s_t, &acc15[9], ln, true, s_pan);  /* (3,3) (4,0) (4,1) */ \
            else sb_acc_sub<3, 0x444, 0x432>(s_P, s_t, &acc15[12], ln, true, s_pan);               /* (4,2) (4,3) (4,4) */ \
        } while (0)
#pragma unroll
        for (int ph = 0; ph < 7; ph++) {
            if (wave == 0) { if (ph < 5) SB_P2B_GROUP(ph); }
            else if (wave == 1) { if (ph >= 1 && ph < 6) SB_P2B_GROUP(ph - 1); }
            else if (wave == 2) { if (ph >= 2) SB_P2B_GROUP(ph - 2); }
            lds_barrier();
        }
#undef SB_P2B_GROUP
        STAMP(1, 26);
        // M_a = L_a^-1 over L_a, band_a <- M_a band_a, N_a = M_a B_(a+1)^T over B_(a+1): every later use of the chain is a product
        {
            double mx[9];
            {                                        // column c of M_a: L x = e_c (threads >= 90 repeat column 89: the values stay in registers across the barrier)
                const int tt = min(td, 9 * SB_NCH - 1), a1 = tt / 9, c = tt - 9 * a1;
                const double *La = s_D + 81 * a1, *li = s_linv + 9 * a1;
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    double sacc = (i == c) ? 1.0 : 0.0;
#pragma unroll
                    for (int j = 0; j < i; j++) sacc -= La[9 * i + j] * mx[j];
                    mx[i] = (i >= c) ? sacc * li[i] : 0.0;
                }
            }
            __syncthreads();
            if (td < 9 * SB_NCH) {
                const int a1 = td / 9, c = td - 9 * a1;
#pragma unroll
                for (int i = 0; i < 9; i++) if (i >= c) s_D[81 * a1 + 9 * i + c] = mx[i];
            }
            __syncthreads();
            if (td < 28 + 19 * (SB_NCH - 1)) {       // band_a <- M_a band_a in place, one thread per column
                const int a = (td < 28) ? 1 : 2 + (td - 28) / 19, p = (td < 28) ? td : (td - 28) % 19;
                double *Bc = s_band + SB_BOFF(a) + p;
                const int str = SB_BSTR(a);
                const double *Ma = s_D + 81 * (a - 1);
                double bc[9], o[9];
#pragma unroll
                for (int kk = 0; kk < 9; kk++) bc[kk] = Bc[kk * str];
#pragma unroll
                for (int i = 0; i < 9; i++) {
                    double sacc = 0;
#pragma unroll
                    for (int kk = 0; kk <= i; kk++) sacc += Ma[9 * i + kk] * bc[kk];
                    o[i] = sacc;
                }
#pragma unroll
                for (int i = 0; i < 9; i++) Bc[i * str] = o[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            double nv[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {            // N_a[i][j] = sum_(k <= i) M_a[i][k] B_(a+1)[j][k], a = 1..9
                const int e = min(td + u * SBT, 81 * (SB_NCH - 1) - 1), a1 = e / 81, ij = e - 81 * a1, i = ij / 9, j = ij - 9 * i;
                const double *Ma = s_D + 81 * a1 + 9 * i, *Bn = s_E + 81 * a1 + 9 * j;
                double sacc = 0;
#pragma unroll
                for (int kk = 0; kk < 9; kk++) sacc += ((kk <= i) ? Ma[kk] : 0.0) * Bn[kk];
                nv[u] = sacc;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 3; u++) { const int e = td + u * SBT; if (e < 81 * (SB_NCH - 1)) s_E[e] = nv[u]; }
        }
        STAMP(1, 27);
        if (tries == 0) block_sum2_sb(g2, part, s_red, G2, Jg2); else __syncthreads();
        if (tries == 0) {                             // cross term of the Cauchy point: 2 sum_p v_p S_p sum_f s_f v_f W_f[p]  (s_t = -that inner sum, scaled)
            double cr = ((ln < VB_NPOSE) ? s_v[ln] * s_t[ln] : 0.0) + ((ln + 64 < VB_NPOSE) ? s_v[ln + 64] * s_t[ln + 64] : 0.0);
            cr = vilf_wave_sum64(cr);
            Jg2 -= 2.0 * cr;
        }
        STAMP(1, 4);
        // ---- P3: Y_a = M_a band_a - N_a Y_(a+1) and Y_a^T Y_a, every wave on its own column tiles (registers only, no barrier) ----------------------------
        double4_t acc[4];
#pragma unroll
        for (int i = 0; i < 4; i++) acc[i] = double4_t{0, 0, 0, 0};
        if (wave == 0) sb_y_chain<0>(s_band, s_E, s_t + 79, acc, ln); else if (wave == 1) sb_y_chain<1>(s_band, s_E, s_t + 79, acc, ln);      // s_t[75..79] stay zero for the whole launch
        else if (wave == 2) sb_y_chain<2>(s_band, s_E, s_t + 79, acc, ln); else sb_y_chain<3>(s_band, s_E, s_t + 79, acc, ln);
        STAMP(1, 5);
        // ---- P4 / P5: Cholesky of the dense block with the trailing matrix in MFMA accumulator registers ---------------------------------------------------
        // Every wave keeps its <= 4 tiles (dense - U^T U from LDS, minus its Y^T Y accumulators) in registers. Per 4-column panel: the lanes holding the
        // panel's columns write them to LDS; one thread per ROW factors the 4 x 4 diagonal block itself and solves its own row strip (the factor row goes to
        // the packed matrix for the back substitution and to the operand buffer); the rank-4 trailing update is ONE MFMA per tile. Right-hand side = row 75.
        {
            int tTA[4], tTB[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int wv = wave;
                tTA[i] = (wv == 0) ? sb_y_ct(0, sb_y_pa(0, i)) : (wv == 1) ? sb_y_ct(1, sb_y_pa(1, i)) : (wv == 2) ? sb_y_ct(2, sb_y_pa(2, i)) : sb_y_ct(3, sb_y_pa(3, i));
                tTB[i] = (wv == 0) ? sb_y_ct(0, sb_y_pb(0, i)) : (wv == 1) ? sb_y_ct(1, sb_y_pb(1, i)) : (wv == 2) ? sb_y_ct(2, sb_y_pb(2, i)) : sb_y_ct(3, sb_y_pb(3, i));
            }
            const int ntl = (wave == 3) ? 3 : 4;
            const int c16 = ln & 15, g4 = ln >> 4;
            double4_t T[4];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int r = 16 * tTA[i] + g4 + 4 * q, c = 16 * tTB[i] + c16;
                    const double v = s_P[sb_prow(min(r, SB_NR - 1)) + min(c, min(r, SB_NR - 1))];
                    T[i][q] = ((r < SB_NR && c <= r) ? v : 0.0) - acc[i][q];
                }
            double *s_lp = s_pan + 304;               // operand buffer of the current panel's factor rows [80][4] (rows 76..79 zero)
            if (td < 16) s_lp[304 + td] = 0.0;
            __syncthreads();                          // every tile is in registers: the packed matrix may be overwritten by the factor
#pragma unroll 1
            for (int bj = 0; bj < 19; bj++) {
                const int j0 = 4 * bj, tc = bj >> 2, sp = bj & 3;
                // 1. the panel's columns, rows >= j0, to LDS
                if ((c16 >> 2) == sp) {
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        if (i < ntl && tTB[i] == tc) {
#pragma unroll
                            for (int q = 0; q < 4; q++) { const int r = 16 * tTA[i] + g4 + 4 * q; if (r >= j0 && r < SB_NR) s_pan[4 * r + (c16 & 3)] = T[i][q]; }
                        }
                }
                __syncthreads();
                // 2. one thread per row: Cholesky of the diagonal block, the row's strip of the factor
                if (td >= j0 && td < SB_NR) {
                    const double *dg = s_pan + 4 * j0, *rp = s_pan + 4 * td;
                    const double d00 = dg[0], d10 = dg[4], d11 = dg[5], d20 = dg[8], d21 = dg[9], d22 = dg[10], d30 = dg[12], d31 = dg[13], d32 = dg[14], d33 = dg[15];
                    const double r0v = rp[0], r1v = rp[1], r2v = rp[2], r3v = rp[3];
                    const double i0 = rsqrt_h3(d00), l10 = d10 * i0, l20 = d20 * i0, l30 = d30 * i0;
                    const double t11 = d11 - l10 * l10, i1 = rsqrt_h3(t11), l21 = (d21 - l20 * l10) * i1, l31 = (d31 - l30 * l10) * i1;
                    const double t22 = d22 - l20 * l20 - l21 * l21, i2 = rsqrt_h3(t22), l32 = (d32 - l30 * l20 - l31 * l21) * i2;
                    const double t33 = d33 - l30 * l30 - l31 * l31 - l32 * l32;
                    const bool last = j0 + 3 >= SB_ND;                // panel 18: column 75 is the right-hand side row's own diagonal — not a pivot
                    const double i3 = last ? 0.0 : rsqrt_h3(t33);
                    const double x0 = r0v * i0, x1 = (r1v - x0 * l10) * i1, x2 = (r2v - x0 * l20 - x1 * l21) * i2, x3 = (r3v - x0 * l30 - x1 * l31 - x2 * l32) * i3;
                    const int k = td - j0;                            // rows of the diagonal block keep their lower part only
                    double *Lr = s_P + sb_prow(td) + j0;
                    Lr[0] = x0;
                    if (k >= 1) Lr[1] = x1;
                    if (k >= 2) Lr[2] = x2;
                    if (k >= 3 && j0 + 3 < SB_ND) Lr[3] = x3;
                    double *lp = s_lp + 4 * td;
                    lp[0] = x0; lp[1] = (k >= 1) ? x1 : 0.0; lp[2] = (k >= 2) ? x2 : 0.0; lp[3] = (k >= 3) ? x3 : 0.0;
                    if (k == 0) {
                        if (!(d00 > 0.0) || !(t11 > 0.0) || !(t22 > 0.0) || (!last && !(t33 > 0.0))) s_flag[2] = 0;
                        const double m10 = -l10 * i0 * i1, m21 = -l21 * i1 * i2, m32 = -l32 * i2 * i3;
                        const double m20 = -(l20 * i0 + l21 * m10) * i2, m31 = -(l31 * i1 + l32 * m21) * i3;
                        const double m30 = -(l30 * i0 + l31 * m10 + l32 * m20) * i3;
                        double *dv = s_dinv + 16 * bj;
                        dv[0] = i0; dv[4] = m10; dv[5] = i1; dv[8] = m20; dv[9] = m21; dv[10] = i2; dv[12] = m30; dv[13] = m31; dv[14] = m32; dv[15] = i3;
                    }
                }
                __syncthreads();
                // 3. rank-4 update of the tiles that reach beyond the panel: T -= L_panel(rows of the tile) L_panel(columns of the tile)^T, one MFMA each.
                //    Rows above the panel's end still hold older strips in the operand buffer: they only touch entries that are never read again.
#pragma unroll
                for (int i = 0; i < 4; i++)
                    if (i < ntl && 16 * tTB[i] + 15 >= j0 + 4) {
                        const double av = -s_lp[4 * min(16 * tTA[i] + c16, 79) + g4], bv = s_lp[4 * min(16 * tTB[i] + c16, 79) + g4];
                        T[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, T[i], 0, 0, 0);
                    }
            }
            __syncthreads();
        }
        STAMP(1, 7);
        tries++;
        if (s_flag[2]) { solved = true; break; }
        mu *= 10.0;                                 // dogleg_strategy.cc: mu_ *= mu_increase_factor_
        if (!(mu < 1.0)) break;                     // max_mu_
            __syncthreads();
    }
    if (!solved) {
        if (tid == 0) { st->solve_failed = 1; st->mu = mu; st->num_linear_solves += tries; st->scaling_ready = 1; st->grad_sqnorm = G2; st->Jg2 = Jg2; }
        return;
    }
    // ---- P6: wave 0 — block back substitution L^T y = z of the dense part (z = row 75 of L); each lane owns entries lane and lane + 64 -----------------
    if (wave == 0) {
        const double *zrow = s_P + sb_prow(SB_ND);
        double z0 = zrow[lane], z1 = (lane + 64 < SB_ND) ? zrow[lane + 64] : 0.0;
        // the factor rows and the inverse diagonal block of the NEXT block are fetched before the current block's dependent chain runs
        double dvn[10], l0n[4], l1n[4];
        auto fetch = [&](int bj) {
            const double *d = s_dinv + 16 * bj;
            dvn[0] = d[0]; dvn[1] = d[4]; dvn[2] = d[5]; dvn[3] = d[8]; dvn[4] = d[9]; dvn[5] = d[10]; dvn[6] = d[12]; dvn[7] = d[13]; dvn[8] = d[14]; dvn[9] = d[15];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) { const int r = 4 * bj + kk; const double *Lr = s_P + sb_prow(r); l0n[kk] = Lr[min(lane, r)]; l1n[kk] = Lr[min(lane + 64, r)]; }
        };
        fetch(18);
#pragma unroll 1
        for (int bj = 18; bj >= 16; bj--) {           // rows 64..75: the second register of lanes 0..10 holds them
            const int r0 = 4 * bj;
            double dv[10], l0[4], l1[4];
#pragma unroll
            for (int q = 0; q < 10; q++) dv[q] = dvn[q];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) { l0[kk] = l0n[kk]; l1[kk] = l1n[kk]; }
            fetch(bj - 1);
            double zb[4], yb[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) zb[kk] = readlane_f64(z1, r0 + kk - 64);
            // y_blk = Ld^-T z_blk (dv = Ld^-1, lower; row / column 3 of block 18 are zero: row 75 is not a variable)
            yb[0] = dv[0] * zb[0] + dv[1] * zb[1] + dv[3] * zb[2] + dv[6] * zb[3];
            yb[1] = dv[2] * zb[1] + dv[4] * zb[2] + dv[7] * zb[3];
            yb[2] = dv[5] * zb[2] + dv[8] * zb[3];
            yb[3] = dv[9] * zb[3];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) {
                z0 -= l0[kk] * yb[kk];
                if (lane + 64 < r0) z1 -= l1[kk] * yb[kk];
                if (lane + 64 == r0 + kk) z1 = yb[kk];
            }
        }
#pragma unroll 1
        for (int bj = 15; bj >= 0; bj--) {            // rows 0..63: one register, one row of the factor per lane
            const int r0 = 4 * bj;
            double dv[10], l0[4];
#pragma unroll
            for (int q = 0; q < 10; q++) dv[q] = dvn[q];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) l0[kk] = l0n[kk];
            if (bj > 0) {
                const double *d = s_dinv + 16 * (bj - 1);
                dvn[0] = d[0]; dvn[1] = d[4]; dvn[2] = d[5]; dvn[3] = d[8]; dvn[4] = d[9]; dvn[5] = d[10]; dvn[6] = d[12]; dvn[7] = d[13]; dvn[8] = d[14]; dvn[9] = d[15];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) { const int r = r0 - 4 + kk; l0n[kk] = s_P[sb_prow(r) + min(lane, r)]; }
            }
            double zb[4], yb[4];
#pragma unroll
            for (int kk = 0; kk < 4; kk++) zb[kk] = readlane_f64(z0, r0 + kk);
            yb[0] = dv[0] * zb[0] + dv[1] * zb[1] + dv[3] * zb[2] + dv[6] * zb[3];
            yb[1] = dv[2] * zb[1] + dv[4] * zb[2] + dv[7] * zb[3];
            yb[2] = dv[5] * zb[2] + dv[8] * zb[3];
            yb[3] = dv[9] * zb[3];
            double upd = l0[0] * yb[0] + l0[1] * yb[1] + l0[2] * yb[2] + l0[3] * yb[3];
            const int kme = lane - r0;
            const double mine = (kme == 0) ? yb[0] : (kme == 1) ? yb[1] : (kme == 2) ? yb[2] : yb[3];
            z0 = (lane < r0) ? z0 - upd : ((kme >= 0 && kme < 4) ? mine : z0);
        }
        s_y[lane] = z0;
        if (lane + 64 < SB_ND) s_y[lane + 64] = z1;
    }
    __syncthreads();
    STAMP(1, 8);
    // chain: c_a = M_a (g~_a - band_a y_dense) from the products M_a band_a (90 threads); (S y)_p for the feature back-substitution
    if (tid < 9 * SB_NCH) {
        const int a = tid / 9 + 1, i = tid - 9 * (a - 1);
        const double *Ba = s_band + SB_BOFF(a) + i * SB_BSTR(a);
        double sacc = Ba[SB_BRHS(a)];
#pragma unroll
        for (int p = 0; p < 18; p++) { const int c = 6 * (a - 1) + p; if (c < VB_NPOSE) sacc -= Ba[p] * s_y[c]; }
        if (a == 1) {
#pragma unroll
            for (int p = 0; p < 9; p++) sacc -= Ba[18 + p] * s_y[VB_NPOSE + p];
        }
        s_w[tid] = sacc;
    }
    if (tid >= 128 && tid < 128 + 80) { const int c = tid - 128; s_v[c] = (c < VB_NPOSE) ? s_scale[c] * s_y[c] : 0.0; }
    __syncthreads();
    if (wave == 0) {
        // forward u_a = c_a - N_a u_(a+1) (a = 10..1), backward w_1 = u_1, w_(a+1) = u_(a+1) - N_a^T w_a; lanes 0..8 = rows
        const int i9 = min(lane, 8);
        double un = 0.0;
#pragma unroll 1
        for (int a = SB_NCH; a >= 1; a--) {
            double t = s_w[9 * (a - 1) + i9];
            if (a < SB_NCH) {
                const double *Na = s_E + 81 * (a - 1) + 9 * i9;
#pragma unroll
                for (int j = 0; j < 9; j++) t -= Na[j] * readlane_f64(un, j);
            }
            un = t;
            if (lane < 9) s_u[9 * (a - 1) + i9] = t;
        }
        double wp = un;                 // w_1 = u_1
        if (lane < 9) s_w[i9] = wp;
#pragma unroll 1
        for (int a = 1; a < SB_NCH; a++) {
            const double *Na = s_E + 81 * (a - 1);
            double t = s_u[9 * a + i9];
#pragma unroll
            for (int j = 0; j < 9; j++) t -= Na[9 * j + i9] * readlane_f64(wp, j);
            wp = t;
            if (lane < 9) s_w[9 * a + i9] = t;
        }
#ifdef VILF_STAMPS
        if (b.dbg && blockIdx.x == 0 && tid == 0) b.dbg[32 + 9] = __builtin_readcyclecounter();
#endif
    } else if (F > 0) {
        // waves 1..3: W~_f[0..65] . (S y)_p for every feature, 16 lanes per feature, three rounds of rows in flight (as feature_dots of k_solve)
        const int c16 = lane & 15, g = lane >> 4, w3 = wave - 1;
        double sv[5];
#pragma unroll
        for (int t5 = 0; t5 < 5; t5++) sv[t5] = s_v[16 * t5 + c16];
        const int Fk4 = (F + 3) & ~3;
        double ra[5], rb[5], rc[5];
#define FD_ISSUE(F0, R)                                                                                                            \
        {                                                                                                                          \
            const double *Wr_ = W + (size_t)min((F0) + g, Fk4 - 1) * VB_WLD + c16;                                                \
            asm volatile("global_load_dwordx2 %0, %5, off\n\tglobal_load_dwordx2 %1, %5, off offset:128\n\t"                       \
                         "global_load_dwordx2 %2, %5, off offset:256\n\tglobal_load_dwordx2 %3, %5, off offset:384\n\t"            \
                         "global_load_dwordx2 %4, %5, off offset:512"                                                              \
                         : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]) : "v"(Wr_) : "memory");                  \
        }
#define FD_WAIT(N, R) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]) : : "memory");
#define FD_STEP(F0, R)                                                                                                             \
        {                                                                                                                          \
            const int f = (F0) + g;                                                                                                \
            double dacc = 0;                                                                                                       \
            _Pragma("unroll") for (int t5 = 0; t5 < 5; t5++) dacc += R[t5] * sv[t5];                                               \
            if (!(f < F)) dacc = 0;                                                                                                \
            _Pragma("unroll") for (int o = 8; o > 0; o >>= 1) dacc += __shfl_xor(dacc, o, 64);                                     \
            if (c16 == 0 && f < F) cf[f] = dacc;                                                                                   \
        }
        constexpr int RS = 4 * 3;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FD_ISSUE(4 * w3, ra) FD_ISSUE(4 * w3 + RS, rb)
        for (int f0 = 4 * w3; f0 < F; f0 += 3 * RS) {
            FD_ISSUE(f0 + 2 * RS, rc) FD_WAIT(10, ra) FD_STEP(f0, ra)
            if (f0 + RS < F) { FD_ISSUE(f0 + 3 * RS, ra) FD_WAIT(10, rb) FD_STEP(f0 + RS, rb) }
            if (f0 + 2 * RS < F) { FD_ISSUE(f0 + 4 * RS, rb) FD_WAIT(10, rc) FD_STEP(f0 + 2 * RS, rc) }
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]),
                                            "+v"(rc[0]), "+v"(rc[1]), "+v"(rc[2]), "+v"(rc[3]), "+v"(rc[4]) : : "memory");
#undef FD_ISSUE
#undef FD_WAIT
#undef FD_STEP
        __threadfence_block();
    }
    __syncthreads();
    if (tid < 9 * SB_NCH) {      // x_a = M_a^T w_a
        const int a1 = tid / 9, i = tid - 9 * a1;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += ((k >= i) ? s_D[81 * a1 + 9 * k + i] : 0.0) * s_w[9 * a1 + k];
        s_y[VB_NPOSE + 9 + tid] = s;
    }
    __syncthreads();
    STAMP(1, 10);
    // ---- P7: Gauss-Newton step = -diagonal_ .* y, feature back-substitution, reductions ---------------------------------------------------------
    double gy = 0, gn2 = 0;
    if (tid < VB_P) {
        const double y = s_y[tid], d = s_diag[tid];
        gn_g[tid] = -d * y;
        gy += s_g[tid] * y;
        gn2 += d * d * y * y;
    }
    for (int f = tid; f < F; f += SBT) {
        if (f_const[f]) continue;
        const double sf = scale_g[VB_P + f], df = diag_g[VB_P + f];
        const double hp = sf * sf * hf[f] + mu * df * df;
        const double gt = sf * gf[f];
        const double y = (gt - sf * cf[f]) / hp;
        gn_g[VB_P + f] = -df * y;
        gy += gt * y;
        gn2 += df * df * y * y;
    }
    block_sum2_sb(gy, gn2, s_red, gy, gn2);
    STAMP(1, 11);
    if (tid == 0) {
        st->grad_sqnorm = G2; st->Jg2 = Jg2; st->alpha = G2 / Jg2;
        st->gy = gy; st->gn_sqnorm = gn2; st->mu = mu; st->mu_used = mu;
        st->num_linear_solves += tries; st->solve_failed = 0; st->reuse = 1; st->scaling_ready = 1;
    }
}
extern "C" __global__ __launch_bounds__(SBT, 2) void k_solve_sb(VbBatch b) { solve_sb_body<false>(b, vb_window(b), 0); }

// ------------------------------------------------------------------------------------------------------------------
// k_iter: ONE launch per iteration, one workgroup per window: the trust-region step + linearisation at the candidate + accept / reject (linearize_body) and — in the
// same workgroup, behind a barrier — the reduce + solve of the new linearisation (solve_sb_body). What the two kernels handed over through per-window double-buffered
// workspaces in HBM (W, facw, pairD, Hpp, imuH / imug, lidH / lidg, g, diagH, hf, gf, cf: ~0.5 MB per window and set, written by one launch, read by the next after 4096
// other windows had gone through the caches) is now the WORKGROUP's scratch, consumed microseconds after it was written. For big batches the scratch is a SLOT, not the
// window's own workspace: a workgroup takes a free slot of its XCD (bitmap in global memory, atomicAnd / atomicOr — a bounded search, nobody waits for anybody) and gives
// it back when it ends, so the batch works in <= 1024 slots (the lowest free ones first: in practice two per CU, close to what the 4 MB L2 of an XCD and the 256 MB
// memory-side cache hold) instead of 2 x B workspaces. A workgroup only ever reads what it has written itself in this launch (the W rows it uses are cleared first), so
// neither another XCD's L2 nor a stale L1 line of the slot's previous user can be observed. A rejected step needs no solve (the old Gauss-Newton step is re-used with a
// smaller radius); an INVALID step asks for another solve from the linearisation at x, which is gone: x is linearised again (same arithmetic, same state: same bits).
// LDS: the larger of the two bodies' plans, overlaid. Launch sequence of a solve: k_iter(iteration_zero = 1), k_iter(0) x (max_iterations - 1), k_linearize_last.
#define VB_SLOTS_PER_XCD 128
#define VB_SLOTS (8 * VB_SLOTS_PER_XCD)
extern "C" __global__ __launch_bounds__(NT, 2) void k_iter(VbBatch b, int iteration_zero, unsigned *slot_bm, int *err) {
    const int w = blockIdx.x + b.w0;
    __shared__ int s_slot;
    if (slot_bm) {
        if (!iteration_zero && b.st[w].done) return;                       // (before a slot is taken: a finished window costs one load)
        if (threadIdx.x == 0) {
            const int xcc = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7);     // HW_REG_XCC_ID[3:0]: slots stay on the XCD whose L2 holds their lines
            int slot = -1;
            for (int attempt = 0; attempt < 8 * 4 * 64 && slot < 0; attempt++) {      // own XCD's four words first, then the others'; bounded
                const int word = ((xcc * 4 + (attempt & 3)) + 4 * ((attempt >> 2) & 7)) & 31;
                const unsigned v = __hip_atomic_load(slot_bm + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!v) continue;
                const int bit = __builtin_ctz(v);
                if (atomicAnd(slot_bm + word, ~(1u << bit)) & (1u << bit)) slot = 32 * word + bit;
            }
            if (slot < 0) atomicExch(err, 1);                              // more workgroups resident than slots: cannot happen at two per CU; reported, not waited for
            s_slot = slot;
        }
        __syncthreads();
        if (s_slot < 0) return;
    }
    const size_t slot = slot_bm ? (size_t)s_slot : (size_t)w;
    const int r = linearize_body<true, false, true>(b, iteration_zero ? 1 : 0, w, slot, (int)threadIdx.x);
    __syncthreads();
    if (r != 2) solve_sb_body<true>(b, w, slot, (int)threadIdx.x);
    if (slot_bm) {
        __syncthreads();                                                   // every load of the slot has returned (the barrier waits for vmcnt(0))
        if (threadIdx.x == 0) atomicOr(slot_bm + (s_slot >> 5), 1u << (s_slot & 31));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// k_step (folded into k_linearize)
// (the trust-region step — dogleg step, candidate, cost, accept / reject — is the prologue and the epilogue of k_linearize)

// ------------------------------------------------------------------------------------------------------------------
// double2vector(): yaw / position gauge fix of the whole window (estimator.cpp:549-596)
extern "C" __global__ void k_finalize(VbBatch b) {
    const int w = blockIdx.x + b.w0, tid = threadIdx.x;
    if (tid >= VB_NF) return;
    const double *pose = b.pose + (size_t)w * 77, *sb = b.sb + (size_t)w * 99;
    const double *R0b = b.gauge_R0 + (size_t)w * 9, *P0b = b.gauge_P0 + (size_t)w * 3;
    double R00[9], y0[3], y00[3], rot[9];
    q_toR(q_load(pose + 3), R00);
    R2ypr(R0b, y0); R2ypr(R00, y00);
    double yd[3] = {y0[0] - y00[0], 0, 0};
    ypr2R(yd, rot);
    if (fabs(fabs(y0[1]) - 90) < 1.0 || fabs(fabs(y00[1]) - 90) < 1.0) {
        // rot_diff = Rs[0] * R00^T
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rot[3 * i + j] = R0b[3 * i] * R00[3 * j] + R0b[3 * i + 1] * R00[3 * j + 1] + R0b[3 * i + 2] * R00[3 * j + 2];
    }
    double Ri[9], Ro[9], d[3], o[3];
    q_toR(q_normalized(q_load(pose + 7 * tid + 3)), Ri);
    m3_mul(rot, Ri, Ro);
    for (int k = 0; k < 9; k++) b.out_Rs[((size_t)w * VB_NF + tid) * 9 + k] = Ro[k];
    d[0] = pose[7 * tid] - pose[0]; d[1] = pose[7 * tid + 1] - pose[1]; d[2] = pose[7 * tid + 2] - pose[2];
    m3_vec(rot, d, o);
    for (int k = 0; k < 3; k++) b.out_Ps[((size_t)w * VB_NF + tid) * 3 + k] = o[k] + P0b[k];
    m3_vec(rot, sb + 9 * tid, o);
    for (int k = 0; k < 3; k++) {
        b.out_Vs[((size_t)w * VB_NF + tid) * 3 + k] = o[k];
        b.out_Bas[((size_t)w * VB_NF + tid) * 3 + k] = sb[9 * tid + 3 + k];
        b.out_Bgs[((size_t)w * VB_NF + tid) * 3 + k] = sb[9 * tid + 6 + k];
    }
}

// options.max_solver_time reached (host clock, vilf_batch_solve): the windows still iterating stop as Ceres does at the top of an iteration —
// termination NO_CONVERGENCE, state = last accepted point. only_margin_old: the 4/5 limit of the windows that marginalize the oldest frame (estimator.cpp:847-850)
extern "C" __global__ void k_time_limit(VbBatch b, const int *mflag, int only_margin_old) {
    const int w = blockIdx.x + b.w0;
    if (threadIdx.x) return;
    if (only_margin_old && mflag[w] != 0) return;
    VbState *st = b.st + w;
    if (!st->done) { st->done = 1; st->termination = 0; }
}

// reset of the per-window solver state (≙ TrustRegionMinimizer::Init + DoglegStrategy ctor) and state rewind
extern "C" __global__ void k_reset(VbBatch b, int rewind_state) {
    const int w = blockIdx.x + b.w0, tid = threadIdx.x;
    if (blockIdx.x == 0 && b.live_ctl && tid < 128) b.live_ctl[tid] = 0;            // the solve's live-list flags and counters (one per iteration)
    if (rewind_state) {
        if (tid < 77) b.pose[(size_t)w * 77 + tid] = b.pose_init[(size_t)w * 77 + tid];
        if (tid < 99) b.sb[(size_t)w * 99 + tid] = b.sb_init[(size_t)w * 99 + tid];
        for (int f = tid; f < b.Fmax; f += blockDim.x) b.feat[(size_t)w * b.Fmax + f] = b.feat_init[(size_t)w * b.Fmax + f];
    }
    if (tid == 0) {
        VbState s;
        s.x_cost = 0; s.cand_cost = 0; s.initial_cost = 0;
        s.radius = b.initial_radius; s.mu = 1e-8; s.alpha = 0; s.dogleg_step_norm = 0;
        s.x_norm = 0; s.gradient_max_norm = 1e300; s.grad_sqnorm = 0; s.Jg2 = 0; s.gy = 0; s.gn_sqnorm = 0; s.mu_used = 1e-8;
        s.model_cost_change = 0; s.relative_decrease = 0;
        s.iteration = 0; s.num_successful = 0; s.num_linear_solves = 0; s.num_consecutive_invalid = 0;
        s.termination = 0; s.done = 0; s.reuse = 0; s.need_linearize = 1; s.solve_failed = 0; s.scaling_ready = 0; s.started = 1; s.ws = 0; s.dev_error = 0;
        b.st[w] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Ceres-layout test hooks: one thread evaluates one factor through the same device functions as above
extern "C" __global__ void k_hook_projection(const double *p0, const double *p1, const double *p2, double lam, const double *pi, const double *pj,
                                             double sqrt_info, double *out /* r[2] Ji[12] Jj[12] Jf[2] */) {
    if (threadIdx.x) return;
    double Ri[9], Rj[9], ric[9], pt[PT_LD];
    q_toR(q_load(p0 + 3), Ri); q_toR(q_load(p1 + 3), Rj); q_toR(q_load(p2 + 3), ric);
    pair_table(p0, Ri, p1, Rj, ric, p2, pt);
    projection_eval_pair<true>(pt, ric, p2, pi, pj, lam, sqrt_info, out, out + 2, out + 14, out + 26);
}
// in: p0 p1 p2 (poses i, j, ex) [21] | pi pj [6] | vel_i vel_j [4] | lam td td_i td_j row_i_c row_j_c tr_over_row sqrt_info [8]; out: r[2] Ji[12] Jj[12] Jex[12] Jf[2] Jtd[2]
extern "C" __global__ void k_hook_projection_td(const double *in, double *out) {
    if (threadIdx.x) return;
    const double *p0 = in, *p1 = in + 7, *p2 = in + 14, *pi = in + 21, *pj = in + 24, *vi = in + 27, *vj = in + 29, *sc = in + 31;
    double Ri[9], Rj[9], ric[9];
    q_toR(q_load(p0 + 3), Ri); q_toR(q_load(p1 + 3), Rj); q_toR(q_load(p2 + 3), ric);
    projection_td_eval<true>(p0, Ri, p1, Rj, ric, p2, pi, pj, vi, vj, sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[0], sc[7], out, out + 2, out + 14, out + 38, out + 26, out + 40);
}
extern "C" __global__ void k_hook_imu(const double *p0, const double *p1, const double *p2, const double *p3, const double *rec, const double *G,
                                      double *out /* r[15] (whitened) J[15*30] (whitened) */, double *scratch /* 450 + 16 */) {
    if (threadIdx.x) return;
    double *Jr = scratch, *rr = scratch + 450;
    imu_raw_eval<true>(p0, p1, p2, p3, rec, G, rr, Jr);
    const double *S = rec + IMU_SQRT;
    for (int i = 0; i < 15; i++) { double s = 0; for (int m = 0; m < 15; m++) s += S[15 * i + m] * rr[m]; out[i] = s; }
    for (int i = 0; i < 15; i++) for (int c = 0; c < 30; c++) { double s = 0; for (int m = 0; m < 15; m++) s += S[15 * i + m] * Jr[30 * m + c]; out[15 + 30 * i + c] = s; }
}
extern "C" __global__ void k_hook_lidar(const double *p0, const double *p1, const double *qil, const double *til, const double *lc, double *out /* r[6] Ji[36] Jj[36] */) {
    if (threadIdx.x) return;
    lidar_between_eval<true>(p0, p1, q_load(qil), til, q_load(lc), lc + 4, out, out + 6, out + 42);
}
// MarginalizationFactor::Evaluate (marginalization_factor.cpp:333-381): dx per kept block (pose: dp, 2 vec(q0^-1 q) with the
// w-sign flip), r = r0 + J0 dx. in: [n, nb][sizes 24][idx 24] as ints; x0[24][9]; x[24][9]; J0 (n x n); r0. out: r[n]
extern "C" __global__ void k_hook_prior(const int *hdr, const double *x0, const double *x, const double *J0, const double *r0, double *out) {
    __shared__ double s_dx[VB_PRIOR_LD];
    const int tid = threadIdx.x, n = hdr[0], nb = hdr[1];
    if (tid < VB_PRIOR_LD) s_dx[tid] = 0.0;
    __syncthreads();
    if (tid < nb) {
        const int size = hdr[2 + tid], idx = hdr[26 + tid];
        const double *a0 = x0 + 9 * tid, *a = x + 9 * tid;
        if (size == 7) {
            for (int k = 0; k < 3; k++) s_dx[idx + k] = a[k] - a0[k];
            Q dq = q_mul(q_inv(q_load(a0 + 3)), q_load(a + 3));
            const double sgn = (dq.w >= 0) ? 2.0 : -2.0;
            s_dx[idx + 3] = sgn * dq.x; s_dx[idx + 4] = sgn * dq.y; s_dx[idx + 5] = sgn * dq.z;
        } else {
            for (int k = 0; k < size; k++) s_dx[idx + k] = a[k] - a0[k];
        }
    }
    __syncthreads();
    for (int r = tid; r < n; r += blockDim.x) {
        double acc = r0[r];
        for (int c = 0; c < n; c++) acc += J0[(size_t)r * n + c] * s_dx[c];
        out[r] = acc;
    }
}
extern "C" __global__ void k_hook_edge(const double *pose, const double *cp, const double *pa, const double *pb, double *out /* r[3] J[18] */) {
    if (threadIdx.x) return;
    edge_eval<true>(pose, cp, pa, pb, out, out + 3);
}
extern "C" __global__ void k_hook_surf(const double *pose, const double *cp, const double *n, double d, double *out /* r[1] J[6] */) {
    if (threadIdx.x) return;
    surf_eval<true>(pose, cp, n, d, out, out + 1);
}
extern "C" __global__ void k_hook_plus(const double *x, const double *d, int kind, double *out) {
    if (threadIdx.x) return;
    if (kind == 0) pose_plus(x, d, out); else se3_plus(x, d, out);
}

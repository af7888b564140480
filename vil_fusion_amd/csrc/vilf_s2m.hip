// vilf_s2m.hip — scan-to-local-map on the MI355X ≙ EstimationMapping::optimation_processing
// (feature_tracker/include/EstimationMapping.hpp:235-296) and its callees.
//
//   pcl::VoxelGrid           -> leaf key per point, stable radix sort (rocPRIM), one thread per occupied leaf sums its points in the
//                               sorted (= input) order: same float sums as a stable CPU sort + sequential centroid
//   pcl::KdTreeFLANN k = 5   -> RADIX-HASHED VOXEL NEIGHBOUR SEARCH: map points radix-sorted by their 1 m cell key, an open-
//                               addressing hash table cell -> [start, end), a query probes the 27 cells around it and keeps the
//                               5 best (squared float distance, ties by map index). Exact for the reference's gate: it only uses
//                               neighbour sets whose 5th squared distance is < 1.0 (EstimationMapping.hpp:129,189), and every
//                               point closer than 1 m lies in the 27-cell block.
//   EdgeCostFactor / SurfCostFactor (:117-232) -> one thread per query: 5-NN, PCA line fit (3x3 Jacobi) / 5x3 column-pivoted QR
//                               plane fit, validity tests
//   ceres::Solve (DENSE_QR, default Levenberg-Marquardt, HuberLoss(0.1), <= 4 iterations, one SE(3) block)
//                            -> ONE persistent 1024-thread workgroup: residual + jacobian of every factor, fixed-tree reduction of
//                               the 6x6 normal equations, Cholesky, accept / reject, radius update (Ceres 2.0 LM semantics)
//   createSubMap (:298-352)  -> transform + append, crop-box compaction (order preserving), voxel grid
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <cmath>
#include <rocprim/rocprim.hpp>
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

using namespace vd;

// ---------------------------------------------------------------------------------------------------------------------
// kernels
struct MinMax { float mn[3], mx[3]; int minb[3]; long long mul1, mul2; };

__global__ void s2m_minmax(const float4 *p, int n, float inv, MinMax *out) {
    __shared__ float s[6][1024];
    const int tid = threadIdx.x;
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < n; i += blockDim.x) {
        const float4 q = p[i];
        mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
        mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
    }
    for (int k = 0; k < 3; k++) { s[k][tid] = mn[k]; s[3 + k][tid] = mx[k]; }
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if (tid < st) for (int k = 0; k < 3; k++) { s[k][tid] = fminf(s[k][tid], s[k][tid + st]); s[3 + k][tid] = fmaxf(s[3 + k][tid], s[3 + k][tid + st]); }
        __syncthreads();
    }
    if (tid == 0) {
        int divb[3];
        for (int k = 0; k < 3; k++) {
            out->mn[k] = s[k][0]; out->mx[k] = s[3 + k][0];
            out->minb[k] = (int)floorf(__fmul_rn(s[k][0], inv));
            divb[k] = (int)floorf(__fmul_rn(s[3 + k][0], inv)) - out->minb[k] + 1;
        }
        out->mul1 = divb[0]; out->mul2 = (long long)divb[0] * divb[1];
    }
}
__global__ void s2m_voxel_keys(const float4 *p, int n, float inv, const MinMax *mm, unsigned long long *keys, int *vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    const long long a = (long long)floorf(__fmul_rn(q.x, inv)) - mm->minb[0], b = (long long)floorf(__fmul_rn(q.y, inv)) - mm->minb[1], c = (long long)floorf(__fmul_rn(q.z, inv)) - mm->minb[2];
    keys[i] = (unsigned long long)(a + b * mm->mul1 + c * mm->mul2);
    vals[i] = i;
}
__global__ void s2m_heads(const unsigned long long *keys, int n, int *head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}
__global__ void s2m_centroids(const float4 *p, const unsigned long long *keys, const int *vals, const int *head, const int *seg, int n, float4 *out, int *n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) *n_out = seg[i] + head[i];
    if (!head[i]) return;
    float cx = 0, cy = 0, cz = 0, ci = 0; int cnt = 0;
    const unsigned long long k = keys[i];
    for (int j = i; j < n && keys[j] == k; j++) { const float4 q = p[vals[j]]; cx = __fadd_rn(cx, q.x); cy = __fadd_rn(cy, q.y); cz = __fadd_rn(cz, q.z); ci = __fadd_rn(ci, q.w); cnt++; }
    const float nn = (float)cnt;
    out[seg[i]] = make_float4(cx / nn, cy / nn, cz / nn, ci / nn);
}
// 1 m cell key: 21 bits per axis
__device__ __forceinline__ unsigned long long cell_key(int ix, int iy, int iz) {
    return ((unsigned long long)(ix + (1 << 20)) << 42) | ((unsigned long long)(iy + (1 << 20)) << 21) | (unsigned long long)(iz + (1 << 20));
}
__global__ void s2m_cell_keys(const float4 *p, int n, unsigned long long *keys, int *vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    keys[i] = cell_key((int)floorf(q.x), (int)floorf(q.y), (int)floorf(q.z));
    vals[i] = i;
}
struct HashEntry { unsigned long long key; int start, end; };
__device__ __forceinline__ unsigned int hash64(unsigned long long k) { k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33; return (unsigned int)k; }
__global__ void s2m_gather_hash(const float4 *p, const unsigned long long *keys, const int *vals, int n, float4 *sorted, HashEntry *table, unsigned int mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 q = p[vals[i]];
    q.w = __int_as_float(vals[i]);               // original map index (tie-break like a linear scan)
    sorted[i] = q;
    if (i == 0 || keys[i] != keys[i - 1]) {
        const unsigned long long k = keys[i];
        int e = i + 1;
        while (e < n && keys[e] == k) e++;
        unsigned int s = hash64(k) & mask;
        for (;;) {
            const unsigned long long prev = atomicCAS(&table[s].key, ~0ULL, k);
            if (prev == ~0ULL) { table[s].start = i; table[s].end = e; break; }
            s = (s + 1) & mask;
        }
    }
}
// exact 5-NN within the 27-cell block: pos[] = positions in the cell-sorted array, ordered by (squared distance, original index)
__device__ void knn5_cells(const float4 *sorted, const HashEntry *table, unsigned int mask, float qx, float qy, float qz, int pos[5], float d2[5]) {
    int oid[5];
#pragma unroll
    for (int k = 0; k < 5; k++) { pos[k] = -1; oid[k] = 0x7fffffff; d2[k] = 3.0e38f; }
    const int cx = (int)floorf(qx), cy = (int)floorf(qy), cz = (int)floorf(qz);
    for (int dz = -1; dz <= 1; dz++) for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) {
        const unsigned long long k = cell_key(cx + dx, cy + dy, cz + dz);
        unsigned int s = hash64(k) & mask;
        int st = 0, en = 0;
        for (;;) {
            const unsigned long long tk = table[s].key;
            if (tk == k) { st = table[s].start; en = table[s].end; break; }
            if (tk == ~0ULL) break;
            s = (s + 1) & mask;
        }
        for (int j = st; j < en; j++) {
            const float4 m = sorted[j];
            const float ex = m.x - qx, ey = m.y - qy, ez = m.z - qz;
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
            const int oi = __float_as_int(m.w);
            if (d < d2[4] || (d == d2[4] && oi < oid[4])) {
                int kk = 4;
                while (kk > 0 && (d < d2[kk - 1] || (d == d2[kk - 1] && oi < oid[kk - 1]))) { d2[kk] = d2[kk - 1]; oid[kk] = oid[kk - 1]; pos[kk] = pos[kk - 1]; kk--; }
                d2[kk] = d; oid[kk] = oi; pos[kk] = j;
            }
        }
    }
}
// 3x3 symmetric eigen-decomposition by cyclic Jacobi: eigenvalues ascending, V columns
__device__ void eig3(const double *Ain, double *w, double *V) {
    double A[9];
    for (int k = 0; k < 9; k++) { A[k] = Ain[k]; V[k] = (k % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off == 0.0) break;
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double app = A[4 * p], aqq = A[4 * q];
            const double g = 100.0 * fabs(apq);
            if (sweep > 3 && fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) { A[3 * p + q] = 0; A[3 * q + p] = 0; continue; }
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int j = 0; j < 3; j++) { const double a = A[3 * p + j], b = A[3 * q + j]; A[3 * p + j] = c * a - s * b; A[3 * q + j] = s * a + c * b; }
            for (int i = 0; i < 3; i++) { const double a = A[3 * i + p], b = A[3 * i + q]; A[3 * i + p] = c * a - s * b; A[3 * i + q] = s * a + c * b; const double va = V[3 * i + p], vb = V[3 * i + q]; V[3 * i + p] = c * va - s * vb; V[3 * i + q] = s * va + c * vb; }
        }
    }
    int o[3] = {0, 1, 2};
    double d[3] = {A[0], A[4], A[8]};
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2 - i; j++) if (d[o[j]] > d[o[j + 1]]) { int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    double Vt[9];
    for (int k = 0; k < 9; k++) Vt[k] = V[k];
    for (int c = 0; c < 3; c++) { w[c] = d[o[c]]; for (int r = 0; r < 3; r++) V[3 * r + c] = Vt[3 * r + o[c]]; }
}
// 5x3 least squares by column-pivoted Householder QR (Eigen colPivHouseholderQr().solve)
__device__ void qr_solve_5x3(const double *Ain, const double *bin, double *x) {
    double a[5][3], b[5];
    for (int i = 0; i < 5; i++) { for (int j = 0; j < 3; j++) a[i][j] = Ain[3 * i + j]; b[i] = bin[i]; }
    int perm[3] = {0, 1, 2};
    double rdiag[3] = {0, 0, 0}, maxpivot = 0;
    for (int k = 0; k < 3; k++) {
        int best = k; double bn = -1, cn[3] = {0, 0, 0};
        for (int j = k; j < 3; j++) { double s = 0; for (int i = k; i < 5; i++) s += a[i][j] * a[i][j]; cn[j] = s; if (s > bn) { bn = s; best = j; } }
        if (best != k) { for (int i = 0; i < 5; i++) { double t = a[i][k]; a[i][k] = a[i][best]; a[i][best] = t; } int t = perm[k]; perm[k] = perm[best]; perm[best] = t; cn[best] = cn[k]; cn[k] = bn; }
        const double nrm = sqrt(cn[k]);
        if (nrm == 0.0) { rdiag[k] = 0; continue; }
        const double alpha = a[k][k] > 0 ? -nrm : nrm;
        double v[5] = {0, 0, 0, 0, 0};
        v[k] = a[k][k] - alpha;
        for (int i = k + 1; i < 5; i++) v[i] = a[i][k];
        double vtv = 0; for (int i = k; i < 5; i++) vtv += v[i] * v[i];
        if (vtv > 0) {
            for (int j = k; j < 3; j++) { double s = 0; for (int i = k; i < 5; i++) s += v[i] * a[i][j]; s = 2 * s / vtv; for (int i = k; i < 5; i++) a[i][j] -= s * v[i]; }
            double s = 0; for (int i = k; i < 5; i++) s += v[i] * b[i]; s = 2 * s / vtv; for (int i = k; i < 5; i++) b[i] -= s * v[i];
        }
        rdiag[k] = a[k][k];
        if (fabs(rdiag[k]) > maxpivot) maxpivot = fabs(rdiag[k]);
    }
    const double thresh = 2.220446049250313e-16 * 3.0 * maxpivot;
    int rank = 0;
    for (int k = 0; k < 3; k++) if (fabs(rdiag[k]) > thresh) rank++;
    double z[3] = {0, 0, 0};
    for (int k = rank - 1; k >= 0; k--) { double s = b[k]; for (int j = k + 1; j < rank; j++) s -= a[k][j] * z[j]; z[k] = s / a[k][k]; }
    x[0] = x[1] = x[2] = 0;
    for (int k = 0; k < 3; k++) x[perm[k]] = z[k];
}

// factor record: [kind (0 invalid, 1 edge, 2 surf)] cp[3] then edge: pa[3] pb[3] / surf: n[3] d  -> 10 doubles + kind
#define S2M_FREC 10
__global__ void s2m_associate(const float4 *pts, int n, int is_surf, const double *pose, const float4 *sorted, const HashEntry *table, unsigned int mask, int nmap,
                              double *frec, int *fkind) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const double cp[3] = {p.x, p.y, p.z};
    double pw[3];
    q_rot(q_load(pose), cp, pw);
    const float qx = (float)(pw[0] + pose[4]), qy = (float)(pw[1] + pose[5]), qz = (float)(pw[2] + pose[6]);
    int kind = 0;
    double rec[S2M_FREC] = {cp[0], cp[1], cp[2], 0, 0, 0, 0, 0, 0, 0};
    int idx[5]; float d2[5];
    if (nmap >= 5) {
        knn5_cells(sorted, table, mask, qx, qy, qz, idx, d2);
        if (d2[4] < 1.0f) {
            double nb[5][3];
            for (int t = 0; t < 5; t++) { const float4 m = sorted[idx[t]]; nb[t][0] = m.x; nb[t][1] = m.y; nb[t][2] = m.z; }
            if (!is_surf) {
                double c[3] = {0, 0, 0};
                for (int j = 0; j < 5; j++) for (int a = 0; a < 3; a++) c[a] += nb[j][a];
                for (int a = 0; a < 3; a++) c[a] /= 5.0;
                double cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                for (int j = 0; j < 5; j++) { const double d[3] = {nb[j][0] - c[0], nb[j][1] - c[1], nb[j][2] - c[2]}; for (int a = 0; a < 3; a++) for (int b2 = 0; b2 < 3; b2++) cov[3 * a + b2] += d[a] * d[b2]; }
                double w[3], V[9];
                eig3(cov, w, V);
                if (w[2] > 3 * w[1]) {
                    kind = 1;
                    for (int a = 0; a < 3; a++) { rec[3 + a] = 0.1 * V[3 * a + 2] + c[a]; rec[6 + a] = -0.1 * V[3 * a + 2] + c[a]; }
                }
            } else {
                double A[15], B[5] = {-1, -1, -1, -1, -1}, nn[3];
                for (int j = 0; j < 5; j++) for (int a = 0; a < 3; a++) A[3 * j + a] = nb[j][a];
                qr_solve_5x3(A, B, nn);
                const double nrm = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                const double d = 1.0 / nrm;
                nn[0] /= nrm; nn[1] /= nrm; nn[2] /= nrm;
                bool ok = true;
                for (int j = 0; j < 5; j++) if (fabs(nn[0] * nb[j][0] + nn[1] * nb[j][1] + nn[2] * nb[j][2] + d) > 0.2) ok = false;
                if (ok) { kind = 2; rec[3] = nn[0]; rec[4] = nn[1]; rec[5] = nn[2]; rec[6] = d; }
            }
        }
    }
    fkind[i] = kind;
    for (int k = 0; k < S2M_FREC; k++) frec[(size_t)i * S2M_FREC + k] = rec[k];
}

// ---- the persistent LM solve -------------------------------------------------------------------------------------------
struct S2MSolveOut { double pose[7]; double final_cost; int iterations; int n_edge, n_surf; int pad; };

#define S2M_NT 1024
__device__ double s2m_block_sum(double v, double *s_red) {
    const int tid = threadIdx.x;
    __syncthreads();
    s_red[tid] = v;
    __syncthreads();
    for (int s = S2M_NT / 2; s > 0; s >>= 1) { if (tid < s) s_red[tid] += s_red[tid + s]; __syncthreads(); }
    const double r = s_red[0];
    __syncthreads();
    return r;
}
// cost (and, JAC: gradient g[6], hessian H[21] lower-packed) at pose x over all valid factors
template <bool JAC>
__device__ void s2m_evaluate(const double *x, const double *frec, const int *fkind, int nfac, double huber_a, double *s_red, double *s_out /*28*/) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0;
    for (int i = threadIdx.x; i < nfac; i += S2M_NT) {
        const int kind = fkind[i];
        if (!kind) continue;
        const double *rec = frec + (size_t)i * S2M_FREC;
        double r[3], J[18];
        int nr;
        if (kind == 1) { edge_eval<JAC>(x, rec, rec + 3, rec + 6, r, J); nr = 3; }
        else { surf_eval<JAC>(x, rec, rec + 3, rec[6], r, J); nr = 1; }
        double s = 0;
        for (int k = 0; k < nr; k++) s += r[k] * r[k];
        double rho0, sw;
        huber(s, huber_a, rho0, sw);
        acc[27] += 0.5 * rho0;
        if (JAC) {
            for (int k = 0; k < nr; k++) {
                const double rk = sw * r[k];
                double jr[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { jr[c] = sw * J[6 * k + c]; acc[21 + c] += jr[c] * rk; }
                int e = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int b2 = 0; b2 <= a; b2++) acc[e++] += jr[a] * jr[b2];
            }
        }
    }
    for (int k = JAC ? 0 : 27; k < 28; k++) { const double v = s2m_block_sum(acc[k], s_red); if (threadIdx.x == 0) s_out[k] = v; }
    __syncthreads();
}

__global__ __launch_bounds__(S2M_NT) void s2m_solve(const double *pose_in, const double *frec, const int *fkind, int n_edge_q, int n_surf_q, double huber_a, int max_it, S2MSolveOut *out) {
    __shared__ double s_red[S2M_NT], s_ev[28], s_cand[28], s_x[7], s_c[7], s_scale[6], s_diag[6], s_step[6];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x, nfac = n_edge_q + n_surf_q;
    if (tid < 7) s_x[tid] = pose_in[tid];
    int ne = 0, ns = 0;
    for (int i = tid; i < nfac; i += S2M_NT) { const int k = fkind[i]; if (k == 1) ne++; else if (k == 2) ns++; }
    const int tne = (int)(s2m_block_sum((double)ne, s_red) + 0.5), tns = (int)(s2m_block_sum((double)ns, s_red) + 0.5);
    if (tne + tns == 0) { if (tid == 0) { for (int k = 0; k < 7; k++) out->pose[k] = s_x[k]; out->final_cost = 0; out->iterations = 0; out->n_edge = 0; out->n_surf = 0; } return; }
    __syncthreads();
    s2m_evaluate<true>(s_x, frec, fkind, nfac, huber_a, s_red, s_ev);
    // thread-0 scalars of the trust-region loop (trust_region_minimizer.cc + levenberg_marquardt_strategy.cc)
    double x_cost = s_ev[27], radius = 1e4, decrease_factor = 2.0, x_norm = 0, mcc = 0;
    bool reuse_diagonal = false;
    int iteration = 0, invalid = 0;
    if (tid == 0) {
        for (int c = 0; c < 6; c++) { const int dd = c * (c + 1) / 2 + c; s_scale[c] = 1.0 / (1.0 + sqrt(s_ev[dd])); }
        for (int k = 0; k < 7; k++) x_norm += s_x[k] * s_x[k];
        x_norm = sqrt(x_norm);
    }
    __syncthreads();
    for (;;) {
        if (tid == 0) {
            int go = 1;
            if (iteration >= max_it) go = 0;
            else {
                double d[6], xp[7], gm = 0;
                for (int c = 0; c < 6; c++) d[c] = -s_ev[21 + c];
                se3_plus(s_x, d, xp);
                for (int k = 0; k < 7; k++) gm = fmax(gm, fabs(s_x[k] - xp[k]));
                if (gm <= 1e-10) go = 0;
                if (radius <= 1e-32) go = 0;
            }
            int valid = 0;
            if (go) {
                iteration++;
                // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system
                double Hs[36], gs[6];
                for (int a = 0; a < 6; a++) { gs[a] = s_ev[21 + a] * s_scale[a]; for (int b2 = 0; b2 <= a; b2++) { const double v = s_ev[a * (a + 1) / 2 + b2] * s_scale[a] * s_scale[b2]; Hs[6 * a + b2] = v; Hs[6 * b2 + a] = v; } }
                if (!reuse_diagonal) for (int c = 0; c < 6; c++) s_diag[c] = fmin(fmax(Hs[7 * c], 1e-6), 1e32);
                double L[36], y[6];
                for (int k = 0; k < 36; k++) L[k] = Hs[k];
                for (int c = 0; c < 6; c++) L[7 * c] += s_diag[c] / radius;
                bool ok = true;
                for (int j = 0; j < 6 && ok; j++) {
                    double sd = L[7 * j];
                    for (int k = 0; k < j; k++) sd -= L[6 * j + k] * L[6 * j + k];
                    if (!(sd > 0)) { ok = false; break; }
                    const double l = sqrt(sd);
                    L[7 * j] = l;
                    for (int i = j + 1; i < 6; i++) { double t = L[6 * i + j]; for (int k = 0; k < j; k++) t -= L[6 * i + k] * L[6 * j + k]; L[6 * i + j] = t / l; }
                }
                reuse_diagonal = true;
                if (ok) {
                    for (int i = 0; i < 6; i++) { double sd = gs[i]; for (int k = 0; k < i; k++) sd -= L[6 * i + k] * y[k]; y[i] = sd / L[7 * i]; }
                    for (int i = 5; i >= 0; i--) { double sd = y[i]; for (int k = i + 1; k < 6; k++) sd -= L[6 * k + i] * y[k]; y[i] = sd / L[7 * i]; }
                    double sg = 0, sHs = 0;
                    for (int a = 0; a < 6; a++) s_step[a] = -y[a];
                    for (int a = 0; a < 6; a++) { sg += s_step[a] * gs[a]; double t = 0; for (int b2 = 0; b2 < 6; b2++) t += Hs[6 * a + b2] * s_step[b2]; sHs += s_step[a] * t; }
                    mcc = -sg - 0.5 * sHs;
                    if (mcc > 0) valid = 1;
                }
                if (!valid) {   // StepIsInvalid -> StepRejected(0)
                    invalid++;
                    radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
                    if (invalid >= 5) go = 0;
                } else {
                    invalid = 0;
                    double d[6];
                    for (int c = 0; c < 6; c++) d[c] = s_step[c] * s_scale[c];
                    se3_plus(s_x, d, s_c);
                }
            }
            s_ctl[0] = go; s_ctl[1] = valid;
        }
        __syncthreads();
        const int go = s_ctl[0], valid = s_ctl[1];
        __syncthreads();
        if (!go) break;
        if (!valid) continue;
        s2m_evaluate<false>(s_c, frec, fkind, nfac, huber_a, s_red, s_cand);
        if (tid == 0) {
            const double cand = s_cand[27];
            double sn = 0;
            for (int k = 0; k < 7; k++) sn += (s_x[k] - s_c[k]) * (s_x[k] - s_c[k]);
            int stop = 0, accept = 0;
            if (sqrt(sn) <= 1e-8 * (x_norm + 1e-8)) stop = 1;
            else if (fabs(x_cost - cand) <= 1e-6 * x_cost) stop = 1;
            else {
                const double rd = (x_cost - cand) / mcc;
                if (rd > 1e-3) {
                    accept = 1;
                    radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * rd - 1.0, 3));
                    radius = fmin(1e16, radius);
                    decrease_factor = 2.0; reuse_diagonal = false;
                    for (int k = 0; k < 7; k++) s_x[k] = s_c[k];
                    x_norm = 0; for (int k = 0; k < 7; k++) x_norm += s_x[k] * s_x[k];
                    x_norm = sqrt(x_norm);
                } else { radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; }
            }
            s_ctl[2] = stop; s_ctl[3] = accept;
        }
        __syncthreads();
        const int stop = s_ctl[2], accept = s_ctl[3];
        __syncthreads();
        if (stop) break;
        if (accept) {     // a rejected step keeps the linearisation at x (s_ev) for the next ComputeStep
            s2m_evaluate<true>(s_x, frec, fkind, nfac, huber_a, s_red, s_ev);
            if (tid == 0) x_cost = s_ev[27];
            __syncthreads();
        }
    }
    if (tid == 0) {
        for (int k = 0; k < 7; k++) out->pose[k] = s_x[k];
        out->final_cost = x_cost; out->iterations = iteration; out->n_edge = tne; out->n_surf = tns;
    }
}

__global__ void s2m_transform_append(const float4 *pts, int n, const double *pose, float4 *dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts[i];
    const double cp[3] = {p.x, p.y, p.z};
    double pw[3];
    q_rot(q_load(pose), cp, pw);
    dst[i] = make_float4((float)(pw[0] + pose[4]), (float)(pw[1] + pose[5]), (float)(pw[2] + pose[6]), p.w);
}
__global__ void s2m_crop_flags(const float4 *p, int n, const double *pose, double half, int *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    const float mnx = (float)(pose[4] - half), mny = (float)(pose[5] - half), mnz = (float)(pose[6] - half);
    const float mxx = (float)(pose[4] + half), mxy = (float)(pose[5] + half), mxz = (float)(pose[6] + half);
    flag[i] = !(q.x < mnx || q.y < mny || q.z < mnz || q.x > mxx || q.y > mxy || q.z > mxz) ? 1 : 0;
}
__global__ void s2m_compact(const float4 *p, const int *flag, const int *pos, int n, float4 *out, int *n_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) out[pos[i]] = p[i];
    if (i == n - 1) *n_out = pos[i] + flag[i];
}
__global__ void s2m_predict(double *pose, double *pose_last, double *prev) {
    // globalOdom_est = globalOdom * (globalOdom_last^-1 * globalOdom) (EstimationMapping.hpp:238-243), rotation via matrices
    if (threadIdx.x) return;
    double R[9], Rl[9], Rrel[9], Re[9], d[3], trel[3], te[3];
    q_toR(q_load(pose), R); q_toR(q_load(pose_last), Rl);
    m3_mulT(Rl, R, Rrel);
    for (int k = 0; k < 3; k++) d[k] = pose[4 + k] - pose_last[4 + k];
    m3T_vec(Rl, d, trel);
    m3_mul(R, Rrel, Re);
    m3_vec(R, trel, te);
    for (int k = 0; k < 7; k++) { prev[k] = pose[k]; pose_last[k] = pose[k]; }
    q_store(pose, q_fromR(Re));
    for (int k = 0; k < 3; k++) pose[4 + k] = te[k] + prev[4 + k];
}

// ---------------------------------------------------------------------------------------------------------------------
// host
struct DCloud { DBuf buf; int n = 0; float4 *p() { return buf.as<float4>(); } };
struct KnnIndex { DBuf sorted, table; unsigned int mask = 0; int n = 0; };

struct S2MCtx {
    DCloud mapEdge, mapSurf, tmpA, tmpB, dsEdge, dsSurf, inE, inS;
    KnnIndex idxEdge, idxSurf;
    DBuf keys, keys2, vals, vals2, head, seg, temp, mm, counter, frec, fkind, pose, solve_out, flag;
    size_t temp_bytes = 0;
    double h_pose[7] = {0, 0, 0, 1, 0, 0, 0}, h_last[7] = {0, 0, 0, 1, 0, 0, 0};
};

static S2MCtx *ctx(vilf_handle *h) {
    if (!h->s2m) {
        h->s2m = new S2MCtx();
        h->s2m->pose.ensure(32 * 8); h->s2m->mm.ensure(sizeof(MinMax)); h->s2m->counter.ensure(64); h->s2m->solve_out.ensure(sizeof(S2MSolveOut));
        hipMemcpy(h->s2m->pose.p, h->s2m->h_pose, 56, hipMemcpyHostToDevice);
        hipMemcpy(h->s2m->pose.as<double>() + 8, h->s2m->h_last, 56, hipMemcpyHostToDevice);
    }
    return h->s2m;
}
void vilf_s2m_release(vilf_handle *h) {
    if (!h->s2m) return;
    S2MCtx *c = h->s2m;
    DBuf *all[] = {&c->mapEdge.buf, &c->mapSurf.buf, &c->tmpA.buf, &c->tmpB.buf, &c->dsEdge.buf, &c->dsSurf.buf, &c->inE.buf, &c->inS.buf, &c->idxEdge.sorted, &c->idxEdge.table,
                   &c->idxSurf.sorted, &c->idxSurf.table, &c->keys, &c->keys2, &c->vals, &c->vals2, &c->head, &c->seg, &c->temp, &c->mm, &c->counter, &c->frec, &c->fkind, &c->pose, &c->solve_out, &c->flag};
    for (DBuf *b : all) b->release();
    delete c;
    h->s2m = nullptr;
}

#define GRID(n) dim3(((n) + 255) / 256), dim3(256)

static int ensure_sort(vilf_handle *h, S2MCtx *c, int n) {
    if (!c->keys.ensure((size_t)n * 8) || !c->keys2.ensure((size_t)n * 8) || !c->vals.ensure((size_t)n * 4) || !c->vals2.ensure((size_t)n * 4) || !c->head.ensure((size_t)n * 4) || !c->seg.ensure((size_t)n * 4)) return VILF_ERR_DEVICE;
    size_t need = 0, need2 = 0;
    rocprim::radix_sort_pairs(nullptr, need, c->keys.as<unsigned long long>(), c->keys2.as<unsigned long long>(), c->vals.as<int>(), c->vals2.as<int>(), (size_t)n, 0, 64, h->stream);
    rocprim::exclusive_scan(nullptr, need2, c->head.as<int>(), c->seg.as<int>(), 0, (size_t)n, rocprim::plus<int>(), h->stream);
    need = std::max(need, need2) + 256;
    if (!c->temp.ensure(need)) return VILF_ERR_DEVICE;
    c->temp_bytes = c->temp.cap;
    return VILF_OK;
}

// pcl::VoxelGrid: in -> out (device), returns out.n
static int voxel_grid(vilf_handle *h, S2MCtx *c, DCloud &in, float leaf, DCloud &out) {
    out.n = 0;
    if (in.n == 0) return VILF_OK;
    const int n = in.n;
    int rc = ensure_sort(h, c, n);
    if (rc != VILF_OK) return rc;
    if (!out.buf.ensure((size_t)n * 16)) return VILF_ERR_DEVICE;
    const float inv = 1.0f / leaf;
    hipLaunchKernelGGL(s2m_minmax, dim3(1), dim3(1024), 0, h->stream, in.p(), n, inv, c->mm.as<MinMax>());
    hipLaunchKernelGGL(s2m_voxel_keys, GRID(n), 0, h->stream, in.p(), n, inv, c->mm.as<MinMax>(), c->keys.as<unsigned long long>(), c->vals.as<int>());
    size_t tb = c->temp_bytes;
    HIPCHECK(h, rocprim::radix_sort_pairs(c->temp.p, tb, c->keys.as<unsigned long long>(), c->keys2.as<unsigned long long>(), c->vals.as<int>(), c->vals2.as<int>(), (size_t)n, 0, 64, h->stream));
    hipLaunchKernelGGL(s2m_heads, GRID(n), 0, h->stream, c->keys2.as<unsigned long long>(), n, c->head.as<int>());
    tb = c->temp_bytes;
    HIPCHECK(h, rocprim::exclusive_scan(c->temp.p, tb, c->head.as<int>(), c->seg.as<int>(), 0, (size_t)n, rocprim::plus<int>(), h->stream));
    hipLaunchKernelGGL(s2m_centroids, GRID(n), 0, h->stream, in.p(), c->keys2.as<unsigned long long>(), c->vals2.as<int>(), c->head.as<int>(), c->seg.as<int>(), n, out.p(), c->counter.as<int>());
    HIPCHECK(h, hipMemcpyAsync(&out.n, c->counter.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}

static int build_index(vilf_handle *h, S2MCtx *c, DCloud &map, KnnIndex &ix) {
    ix.n = map.n;
    if (map.n == 0) return VILF_OK;
    const int n = map.n;
    int rc = ensure_sort(h, c, n);
    if (rc != VILF_OK) return rc;
    unsigned int T = 64;
    while (T < 2u * (unsigned int)n) T <<= 1;
    ix.mask = T - 1;
    if (!ix.sorted.ensure((size_t)n * 16) || !ix.table.ensure((size_t)T * sizeof(HashEntry))) return VILF_ERR_DEVICE;
    HIPCHECK(h, hipMemsetAsync(ix.table.p, 0xff, (size_t)T * sizeof(HashEntry), h->stream));
    hipLaunchKernelGGL(s2m_cell_keys, GRID(n), 0, h->stream, map.p(), n, c->keys.as<unsigned long long>(), c->vals.as<int>());
    size_t tb = c->temp_bytes;
    HIPCHECK(h, rocprim::radix_sort_pairs(c->temp.p, tb, c->keys.as<unsigned long long>(), c->keys2.as<unsigned long long>(), c->vals.as<int>(), c->vals2.as<int>(), (size_t)n, 0, 64, h->stream));
    hipLaunchKernelGGL(s2m_gather_hash, GRID(n), 0, h->stream, map.p(), c->keys2.as<unsigned long long>(), c->vals2.as<int>(), n, ix.sorted.as<float4>(), ix.table.as<HashEntry>(), ix.mask);
    HIPCHECK(h, hipGetLastError());
    return VILF_OK;
}

static int upload_cloud(vilf_handle *h, DCloud &c, const float *xyzi, int n) {
    c.n = n;
    if (n == 0) return VILF_OK;
    if (!c.buf.ensure((size_t)n * 16)) return VILF_ERR_DEVICE;
    HIPCHECK(h, hipMemcpyAsync(c.buf.p, xyzi, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    return VILF_OK;
}
static int append_cloud(vilf_handle *h, DCloud &dst, const DCloud &src) {    // dst += src (device copy)
    if (src.n == 0) return VILF_OK;
    if ((size_t)(dst.n + src.n) * 16 > dst.buf.cap) {
        DBuf nb;
        if (!nb.ensure((size_t)(dst.n + src.n) * 16 * 2)) return VILF_ERR_DEVICE;
        if (dst.n) HIPCHECK(h, hipMemcpyAsync(nb.p, dst.buf.p, (size_t)dst.n * 16, hipMemcpyDeviceToDevice, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        dst.buf.release();
        dst.buf = nb;
    }
    HIPCHECK(h, hipMemcpyAsync(dst.p() + dst.n, src.buf.p, (size_t)src.n * 16, hipMemcpyDeviceToDevice, h->stream));
    dst.n += src.n;
    return VILF_OK;
}

extern "C" int vilf_scan2map_init(vilf_handle *h, const float *e, int ne, const float *s, int ns) {
    if (!h || ne < 0 || ns < 0 || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    S2MCtx *c = ctx(h);
    int rc;
    if ((rc = upload_cloud(h, c->inE, e, ne)) != VILF_OK || (rc = upload_cloud(h, c->inS, s, ns)) != VILF_OK) return rc;
    if ((rc = append_cloud(h, c->mapEdge, c->inE)) != VILF_OK || (rc = append_cloud(h, c->mapSurf, c->inS)) != VILF_OK) return rc;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}

extern "C" int vilf_scan2map_set_pose(vilf_handle *h, const double p[7], const double pl[7]) {
    if (!h || !p || !pl) return VILF_ERR_INVALID_ARGUMENT;
    S2MCtx *c = ctx(h);
    std::memcpy(c->h_pose, p, 56); std::memcpy(c->h_last, pl, 56);
    HIPCHECK(h, hipMemcpy(c->pose.p, p, 56, hipMemcpyHostToDevice));
    HIPCHECK(h, hipMemcpy(c->pose.as<double>() + 8, pl, 56, hipMemcpyHostToDevice));
    return VILF_OK;
}

extern "C" int vilf_scan2map_get_map(vilf_handle *h, int which, float *out, int cap, int *n_out) {
    if (!h || !n_out) return VILF_ERR_INVALID_ARGUMENT;
    S2MCtx *c = ctx(h);
    DCloud &m = which == 0 ? c->mapEdge : c->mapSurf;
    *n_out = m.n;
    const int k = std::min(cap, m.n);
    if (k > 0 && out) HIPCHECK(h, hipMemcpy(out, m.buf.p, (size_t)k * 16, hipMemcpyDeviceToHost));
    return VILF_OK;
}

extern "C" int vilf_scan2map_step(vilf_handle *h, const float *e, int ne, const float *s, int ns, vilf_scan2map_result *res) {
    if (!h || !res || ne < 0 || ns < 0 || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    S2MCtx *c = ctx(h);
    std::memset(res, 0, sizeof(*res));
    double *d_pose = c->pose.as<double>(), *d_last = d_pose + 8, *d_prev = d_pose + 16;
    hipLaunchKernelGGL(s2m_predict, dim3(1), dim3(64), 0, h->stream, d_pose, d_last, d_prev);
    int rc;
    if ((rc = upload_cloud(h, c->inE, e, ne)) != VILF_OK || (rc = upload_cloud(h, c->inS, s, ns)) != VILF_OK) return rc;
    if ((rc = voxel_grid(h, c, c->inE, (float)h->opts.edge_leaf_size, c->dsEdge)) != VILF_OK) return rc;
    if ((rc = voxel_grid(h, c, c->inS, (float)h->opts.surf_leaf_size, c->dsSurf)) != VILF_OK) return rc;
    res->n_edge_ds = c->dsEdge.n; res->n_surf_ds = c->dsSurf.n;
    const int nq = c->dsEdge.n + c->dsSurf.n;
    if (c->mapEdge.n > 10 && c->mapSurf.n > 50 && nq > 0) {
        if ((rc = build_index(h, c, c->mapEdge, c->idxEdge)) != VILF_OK || (rc = build_index(h, c, c->mapSurf, c->idxSurf)) != VILF_OK) return rc;
        if (!c->frec.ensure((size_t)nq * S2M_FREC * 8) || !c->fkind.ensure((size_t)nq * 4)) return VILF_ERR_DEVICE;
        for (int iter = 0; iter < h->opts.s2m_outer_iterations && iter < 2; iter++) {
            if (c->dsEdge.n) hipLaunchKernelGGL(s2m_associate, GRID(c->dsEdge.n), 0, h->stream, c->dsEdge.p(), c->dsEdge.n, 0, d_pose, c->idxEdge.sorted.as<float4>(), c->idxEdge.table.as<HashEntry>(), c->idxEdge.mask, c->mapEdge.n, c->frec.as<double>(), c->fkind.as<int>());
            if (c->dsSurf.n) hipLaunchKernelGGL(s2m_associate, GRID(c->dsSurf.n), 0, h->stream, c->dsSurf.p(), c->dsSurf.n, 1, d_pose, c->idxSurf.sorted.as<float4>(), c->idxSurf.table.as<HashEntry>(), c->idxSurf.mask, c->mapSurf.n, c->frec.as<double>() + (size_t)c->dsEdge.n * S2M_FREC, c->fkind.as<int>() + c->dsEdge.n);
            hipLaunchKernelGGL(s2m_solve, dim3(1), dim3(S2M_NT), 0, h->stream, d_pose, c->frec.as<double>(), c->fkind.as<int>(), c->dsEdge.n, c->dsSurf.n, h->opts.huber_a, h->opts.s2m_max_iterations, c->solve_out.as<S2MSolveOut>());
            S2MSolveOut so;
            HIPCHECK(h, hipMemcpyAsync(&so, c->solve_out.p, sizeof(so), hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(h, hipMemcpyAsync(d_pose, c->solve_out.p, 56, hipMemcpyDeviceToDevice, h->stream));
            HIPCHECK(h, hipStreamSynchronize(h->stream));
            res->n_edge_factors[iter] = so.n_edge; res->n_surf_factors[iter] = so.n_surf; res->iterations[iter] = so.iterations; res->final_cost[iter] = so.final_cost;
        }
    }
    // createSubMap: append registered points, crop, voxel grid
    for (int which = 0; which < 2; which++) {
        DCloud &ds = which ? c->dsSurf : c->dsEdge, &map = which ? c->mapSurf : c->mapEdge;
        const float leaf = (float)(which ? h->opts.surf_leaf_size : h->opts.edge_leaf_size);
        if (ds.n) {
            if (!c->tmpA.buf.ensure((size_t)ds.n * 16)) return VILF_ERR_DEVICE;
            hipLaunchKernelGGL(s2m_transform_append, GRID(ds.n), 0, h->stream, ds.p(), ds.n, d_pose, c->tmpA.p());
            c->tmpA.n = ds.n;
            if ((rc = append_cloud(h, map, c->tmpA)) != VILF_OK) return rc;
        }
        if (map.n == 0) continue;
        const int n = map.n;
        if ((rc = ensure_sort(h, c, n)) != VILF_OK) return rc;
        if (!c->flag.ensure((size_t)n * 4) || !c->tmpB.buf.ensure((size_t)n * 16)) return VILF_ERR_DEVICE;
        hipLaunchKernelGGL(s2m_crop_flags, GRID(n), 0, h->stream, map.p(), n, d_pose, h->opts.s2m_crop_half, c->flag.as<int>());
        size_t tb = c->temp_bytes;
        HIPCHECK(h, rocprim::exclusive_scan(c->temp.p, tb, c->flag.as<int>(), c->seg.as<int>(), 0, (size_t)n, rocprim::plus<int>(), h->stream));
        hipLaunchKernelGGL(s2m_compact, GRID(n), 0, h->stream, map.p(), c->flag.as<int>(), c->seg.as<int>(), n, c->tmpB.p(), c->counter.as<int>());
        HIPCHECK(h, hipMemcpyAsync(&c->tmpB.n, c->counter.p, 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        DCloud out;
        if ((rc = voxel_grid(h, c, c->tmpB, leaf, out)) != VILF_OK) return rc;
        map.buf.release();
        map.buf = out.buf; map.n = out.n;
    }
    double hp[24];
    HIPCHECK(h, hipMemcpy(hp, d_pose, sizeof(hp), hipMemcpyDeviceToHost));
    std::memcpy(c->h_pose, hp, 56); std::memcpy(c->h_last, hp + 8, 56);
    std::memcpy(res->pose_qt, hp, 56);
    {   // /Odometry relative pose: q_last^-1 * q, q_last^-1 * (t - t_last) (feature_tracker_node.cpp:392-394)
        const double *pv = hp + 16;
        const double n2 = pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2] + pv[3] * pv[3];
        const double qi[4] = {-pv[0] / n2, -pv[1] / n2, -pv[2] / n2, pv[3] / n2};
        const double *q = hp;
        res->rel_q[0] = qi[3] * q[0] + qi[0] * q[3] + qi[1] * q[2] - qi[2] * q[1];
        res->rel_q[1] = qi[3] * q[1] + qi[1] * q[3] + qi[2] * q[0] - qi[0] * q[2];
        res->rel_q[2] = qi[3] * q[2] + qi[2] * q[3] + qi[0] * q[1] - qi[1] * q[0];
        res->rel_q[3] = qi[3] * q[3] - qi[0] * q[0] - qi[1] * q[1] - qi[2] * q[2];
        const double v[3] = {hp[4] - pv[4], hp[5] - pv[5], hp[6] - pv[6]};
        const double ux = 2 * (qi[1] * v[2] - qi[2] * v[1]), uy = 2 * (qi[2] * v[0] - qi[0] * v[2]), uz = 2 * (qi[0] * v[1] - qi[1] * v[0]);
        res->rel_t[0] = v[0] + qi[3] * ux + (qi[1] * uz - qi[2] * uy);
        res->rel_t[1] = v[1] + qi[3] * uy + (qi[2] * ux - qi[0] * uz);
        res->rel_t[2] = v[2] + qi[3] * uz + (qi[0] * uy - qi[1] * ux);
    }
    res->map_edge_size = c->mapEdge.n; res->map_surf_size = c->mapSurf.n;
    return VILF_OK;
}

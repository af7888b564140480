// vilf_feat.hip — LOAM-style LiDAR feature extraction on the MI355X ≙ featureExtraction::extractFeature
// (feature_tracker/include/featureExtraction.hpp:54-232), the stage that turns a raw scan into the edge / surf clouds
// EstimationMapping consumes.
//
//   getLaserCloud (:54-112)            -> one lane per point: vertical angle -> ring id (the reference's float / double mix), then ONE
//                                         stable radix sort by ring id = the per-ring clouds in firing order
//   curvature (:179-194)               -> one lane per ring point: 10-neighbour second difference, summed in FLOAT left to right
//   featureExtractionFromSector (:114-165), six sectors per ring (:196-211)
//                                      -> one workgroup per (ring, sector): the sector's (curvature, index) pairs and its points (+-5
//                                         halo) live in LDS; bitonic sort ascending by (curvature, index); lane 0 walks the sorted list
//                                         from the top (<= 20 edge picks, +-5 neighbour suppression while consecutive points are closer
//                                         than sqrt(0.05) m); all lanes then compact the un-picked points in sorted order (surf)
//   output order                       -> rings ascending, sectors ascending, edge picks in pick order, surf in ascending curvature:
//                                         a scan over the 6 * N_SCANS counts, then one gather
// Kept quirks: see oracle/lidar_features.cpp (dropped last element per sector, the 21st pick, sector-local suppression).
// Exactly equal curvatures are ordered by point index (std::sort leaves that order unspecified in the reference).
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <cmath>
#include "vilf_sort.hpp"
#include "vilf_internal.hpp"

#define FE_SEC 6
#define FE_MAXSEC 1024          // elements of one sector after padding (ring size <= ~6150 points)
#define FE_HALO 5
#define FE_NT 256

struct FeatCtx {
    DBuf pts, keys, keys2, vals, vals2, temp, rpts, curv, rstart, etmp, stmp, ecnt, scnt, eoff, soff, oute, outs, tot;
    DBuf dcloud, dunit, dfeat, dout;          // getFeatureDepth
    size_t temp_bytes = 0;
    int cap = 0, scans = 0;
};

__global__ void fe_ring(const float4 *p, int n, int n_scans, double min_r, double max_r, unsigned int *keys, int *vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    int id = -1;
    const float dxy = sqrtf(__fadd_rn(__fmul_rn(q.x, q.x), __fmul_rn(q.y, q.y)));      // DistanceXY (common.h:59-62): float
    const double distance = dxy;
    if (!(distance < min_r || distance > max_r)) {
        const double angle = atan((double)q.z / distance) * 180 / M_PI;
        if (angle == angle) {
            // explicit roundings: a contracted multiply-add could move a point across a ring boundary
            if (n_scans == 16) { id = int(__dadd_rn(__ddiv_rn(__dadd_rn(angle, 15.0), 2.0), 0.5)); if (id > n_scans - 1 || id < 0) id = -1; }
            else if (n_scans == 32) { id = int(__ddiv_rn(__dmul_rn(__dadd_rn(angle, 92.0 / 3.0), 3.0), 4.0)); if (id > n_scans - 1 || id < 0) id = -1; }
            else {
                if (angle >= -8.83) id = int(__dadd_rn(__dmul_rn(__dsub_rn(2.0, angle), 3.0), 0.5));
                else id = n_scans / 2 + int(__dadd_rn(__dmul_rn(__dsub_rn(-8.83, angle), 2.0), 0.5));
                if (angle > 2 || angle < -24.33 || id > 63 || id < 0) id = -1;
            }
        }
    }
    keys[i] = id < 0 ? (unsigned int)n_scans : (unsigned int)id;     // rejected points sort behind the last ring
    vals[i] = i;
}
// ring r occupies [rstart[r], rstart[r + 1]) of the sorted arrays
__global__ void fe_ring_bounds(const unsigned int *keys, int n, int n_scans, int *rstart) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    const unsigned int k = i < n ? keys[i] : (unsigned int)n_scans + 1, kp = i > 0 ? keys[i - 1] : 0xffffffffu;
    if (i == 0) { for (unsigned int r = 0; r <= k && r <= (unsigned int)n_scans; r++) rstart[r] = 0; }
    else if (k != kp) { for (unsigned int r = kp + 1; r <= k && r <= (unsigned int)n_scans; r++) rstart[r] = i; }
}
__global__ void fe_gather_curv(const float4 *p, const int *vals, const int *rstart, int n, int n_scans, float4 *rp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rp[i] = p[vals[i]];
}
__global__ void fe_curv(const float4 *rp, const unsigned int *keys, const int *rstart, int n, int n_scans, double *curv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned int r = keys[i];
    if (r >= (unsigned int)n_scans) return;
    const int s0 = rstart[r], cnt = rstart[r + 1] - s0, j = i - s0;
    if (cnt < 131 || j < 5 || j >= cnt - 5) return;
    const float4 *c = rp + i;
    // float arithmetic, left to right, no contraction (featureExtraction.hpp:181-189)
#define FE_SUM(m) __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fsub_rn(__fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(c[-5].m, c[-4].m), c[-3].m), c[-2].m), c[-1].m), __fmul_rn(10.0f, c[0].m)), c[1].m), c[2].m), c[3].m), c[4].m), c[5].m)
    const double X = FE_SUM(x), Y = FE_SUM(y), Z = FE_SUM(z);
#undef FE_SUM
    curv[i] = __dadd_rn(__dadd_rn(__dmul_rn(X, X), __dmul_rn(Y, Y)), __dmul_rn(Z, Z));
}

__device__ __forceinline__ bool fe_less(double va, int ia, double vb, int ib) { return va < vb || (va == vb && ia < ib); }

__global__ __launch_bounds__(FE_NT) void fe_sector(const float4 *rp, const double *curv, const int *rstart, int n_scans, double edge_thr,
                                                   float4 *etmp, float4 *stmp, int *ecnt, int *scnt, int *err) {
    __shared__ double s_val[FE_MAXSEC];
    __shared__ int s_ind[FE_MAXSEC];
    __shared__ float4 s_pt[FE_MAXSEC + 2 * FE_HALO + 2];
    __shared__ unsigned char s_pick[FE_MAXSEC + 2 * FE_HALO + 2];
    __shared__ int s_w[FE_NT / 64], s_ne;
    const int tid = threadIdx.x, s = blockIdx.x, r = blockIdx.y, sec = r * FE_SEC + s;
    const int s0 = rstart[r], cnt = rstart[r + 1] - s0;
    if (cnt < 131) { if (tid == 0) { ecnt[sec] = 0; scnt[sec] = 0; } return; }
    const int cloud_size = cnt - 10, len = cloud_size / 6, start = len * s;
    const int end = (s == 5) ? cloud_size - 1 : len * (s + 1) - 1;
    const int m = end - start;                         // the reference's iterator range drops the element at sector_end (:208)
    if (m <= 0) { if (tid == 0) { ecnt[sec] = 0; scnt[sec] = 0; } return; }
    if (m > FE_MAXSEC) { if (tid == 0) { ecnt[sec] = 0; scnt[sec] = 0; atomicOr(err, 1); } return; }
    const int j0 = 5 + start;                          // ring-local index of the sector's first element
    for (int e = tid; e < FE_MAXSEC; e += FE_NT) {
        if (e < m) { s_val[e] = curv[s0 + j0 + e]; s_ind[e] = j0 + e; }
        else { s_val[e] = __longlong_as_double(0x7ff0000000000000LL); s_ind[e] = 0x7fffffff; }
    }
    for (int e = tid; e < m + 2 * FE_HALO; e += FE_NT) { s_pt[e] = rp[s0 + j0 - FE_HALO + e]; s_pick[e] = 0; }      // j0 - 5 >= 0 and j0 + m + 4 <= cnt - 1
    __syncthreads();
    // bitonic sort, ascending by (curvature, index); only the first power of two >= m takes part
    int N = 1; while (N < m) N <<= 1;
    for (int k = 2; k <= N; k <<= 1)
        for (int jj = k >> 1; jj > 0; jj >>= 1) {
            for (int e = tid; e < N; e += FE_NT) {
                const int x = e ^ jj;
                if (x > e) {
                    const double va = s_val[e], vb = s_val[x];
                    const int ia = s_ind[e], ib = s_ind[x];
                    const bool up = (e & k) == 0;
                    if (fe_less(vb, ib, va, ia) == up) { s_val[e] = vb; s_val[x] = va; s_ind[e] = ib; s_ind[x] = ia; }
                }
            }
            __syncthreads();
        }
    // lane 0: greedy picks from the largest curvature down (:120-157)
    if (tid == 0) {
        int largest = 0, ne = 0;
        auto far = [&](int a, int b) {     // squared distance of ring points a, b (ring-local indices) > 0.05, in the reference's float / double mix
            const float4 pa = s_pt[a - j0 + FE_HALO], pb = s_pt[b - j0 + FE_HALO];
            const double ex = __fsub_rn(pa.x, pb.x), ey = __fsub_rn(pa.y, pb.y), ez = __fsub_rn(pa.z, pb.z);
            return __dadd_rn(__dadd_rn(__dmul_rn(ex, ex), __dmul_rn(ey, ey)), __dmul_rn(ez, ez)) > 0.05;
        };
        for (int i = m - 1; i >= 0; i--) {
            const int ind = s_ind[i];
            if (s_pick[ind - j0 + FE_HALO]) continue;
            if (s_val[i] <= edge_thr) break;
            largest++;
            s_pick[ind - j0 + FE_HALO] = 1;
            if (largest <= 20) etmp[(size_t)sec * 20 + ne++] = s_pt[ind - j0 + FE_HALO]; else break;
            for (int k = 1; k <= 5; k++) { if (far(ind + k, ind + k - 1)) break; s_pick[ind + k - j0 + FE_HALO] = 1; }
            for (int l = -1; l >= -5; l--) { if (far(ind + l, ind + l + 1)) break; s_pick[ind + l - j0 + FE_HALO] = 1; }
        }
        s_ne = ne;
    }
    __syncthreads();
    // surf: the un-picked points in ascending (curvature, index) order (:159-165)
    int carry = 0;
    for (int t0 = 0; t0 < m; t0 += FE_NT) {
        const int e = t0 + tid;
        const int keep = (e < m && !s_pick[s_ind[min(e, m - 1)] - j0 + FE_HALO]) ? 1 : 0;
        int incl = keep;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += u; }
        __syncthreads();
        if ((tid & 63) == 63) s_w[tid >> 6] = incl;
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int k = 0; k < FE_NT / 64; k++) { const int x = s_w[k]; if (k < (tid >> 6)) off += x; tot += x; }
        if (keep) stmp[(size_t)sec * FE_MAXSEC + carry + off + incl - 1] = s_pt[s_ind[e] - j0 + FE_HALO];
        carry += tot;
    }
    if (tid == 0) { ecnt[sec] = s_ne; scnt[sec] = carry; }
}
__global__ void fe_offsets(const int *ecnt, const int *scnt, int nsec, int *eoff, int *soff, int *tot) {
    if (threadIdx.x || blockIdx.x) return;
    int a = 0, b = 0;
    for (int i = 0; i < nsec; i++) { eoff[i] = a; soff[i] = b; a += ecnt[i]; b += scnt[i]; }
    tot[0] = a; tot[1] = b;
}
__global__ void fe_emit(const float4 *etmp, const float4 *stmp, const int *ecnt, const int *scnt, const int *eoff, const int *soff, float4 *oute, float4 *outs) {
    const int sec = blockIdx.x, tid = threadIdx.x;
    for (int k = tid; k < ecnt[sec]; k += blockDim.x) oute[eoff[sec] + k] = etmp[(size_t)sec * 20 + k];
    for (int k = tid; k < scnt[sec]; k += blockDim.x) outs[soff[sec] + k] = stmp[(size_t)sec * FE_MAXSEC + k];
}

// ---- getFeatureDepth (feature_tracker/feature_tracker_node.cpp:54-163) -------------------------------------------------------
// unit-sphere projection of the depth cloud (one lane per point), then ONE workgroup per visual feature: every lane keeps the 3
// nearest cloud points of its strided share (float squared distance, ties by index), the per-lane triples are merged through LDS,
// lane 0 intersects the feature ray with the plane of the 3 winners. FLOAT arithmetic with explicit roundings, as the reference.
__global__ void fd_unit(const float4 *p, int n, float4 *u) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 q = p[i];
    const float range = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(q.x, q.x), __fmul_rn(q.y, q.y)), __fmul_rn(q.z, q.z)));
    u[i] = make_float4(__fdiv_rn(q.x, range), __fdiv_rn(q.y, range), __fdiv_rn(q.z, range), range);
}
__device__ __forceinline__ void fd_insert(float d, int i, float d2[3], int idx[3]) {
    if (d < d2[2] || (d == d2[2] && i < idx[2])) {
        int k = 2;
        while (k > 0 && (d < d2[k - 1] || (d == d2[k - 1] && i < idx[k - 1]))) { d2[k] = d2[k - 1]; idx[k] = idx[k - 1]; k--; }
        d2[k] = d; idx[k] = i;
    }
}
__global__ __launch_bounds__(256) void fd_depth(const float4 *u, int n, const float *feat, int m, float thr, float *out) {
    __shared__ float s_d[256 * 3];
    __shared__ int s_i[256 * 3];
    const int f = blockIdx.x, tid = threadIdx.x;
    float vx = feat[3 * f], vy = feat[3 * f + 1], vz = feat[3 * f + 2];
    const float nrm = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(vx, vx), __fmul_rn(vy, vy)), __fmul_rn(vz, vz)));
    vx = __fdiv_rn(vx, nrm); vy = __fdiv_rn(vy, nrm); vz = __fdiv_rn(vz, nrm);
    float d2[3] = {3.0e38f, 3.0e38f, 3.0e38f}; int idx[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff};
    for (int i = tid; i < n; i += 256) {
        const float4 q = u[i];
        const float ex = __fsub_rn(q.x, vx), ey = __fsub_rn(q.y, vy), ez = __fsub_rn(q.z, vz);
        fd_insert(__fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)), i, d2, idx);
    }
    for (int k = 0; k < 3; k++) { s_d[3 * tid + k] = d2[k]; s_i[3 * tid + k] = idx[k]; }
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {          // tree merge of the per-lane triples
        if (tid < st) {
            for (int k = 0; k < 3; k++) { d2[k] = s_d[3 * tid + k]; idx[k] = s_i[3 * tid + k]; }
            for (int k = 0; k < 3; k++) if (s_i[3 * (tid + st) + k] != 0x7fffffff) fd_insert(s_d[3 * (tid + st) + k], s_i[3 * (tid + st) + k], d2, idx);
            for (int k = 0; k < 3; k++) { s_d[3 * tid + k] = d2[k]; s_i[3 * tid + k] = idx[k]; }
        }
        __syncthreads();
    }
    if (tid) return;
    float res = -1.0f;
    if (idx[2] != 0x7fffffff && d2[2] < thr) {
        const float4 p1 = u[idx[0]], p2 = u[idx[1]], p3 = u[idx[2]];
        const float r1 = p1.w, r2 = p2.w, r3 = p3.w;
        const float A[3] = {__fmul_rn(p1.x, r1), __fmul_rn(p1.y, r1), __fmul_rn(p1.z, r1)};
        const float B[3] = {__fmul_rn(p2.x, r2), __fmul_rn(p2.y, r2), __fmul_rn(p2.z, r2)};
        const float Cc[3] = {__fmul_rn(p3.x, r3), __fmul_rn(p3.y, r3), __fmul_rn(p3.z, r3)};
        const float a[3] = {__fsub_rn(A[0], B[0]), __fsub_rn(A[1], B[1]), __fsub_rn(A[2], B[2])}, b[3] = {__fsub_rn(B[0], Cc[0]), __fsub_rn(B[1], Cc[1]), __fsub_rn(B[2], Cc[2])};
        const float N[3] = {__fsub_rn(__fmul_rn(a[1], b[2]), __fmul_rn(a[2], b[1])), __fsub_rn(__fmul_rn(a[2], b[0]), __fmul_rn(a[0], b[2])), __fsub_rn(__fmul_rn(a[0], b[1]), __fmul_rn(a[1], b[0]))};
        float sc = __fdiv_rn(__fadd_rn(__fadd_rn(__fmul_rn(N[0], A[0]), __fmul_rn(N[1], A[1])), __fmul_rn(N[2], A[2])),
                             __fadd_rn(__fadd_rn(__fmul_rn(N[0], vx), __fmul_rn(N[1], vy)), __fmul_rn(N[2], vz)));
        const float mn = fminf(r1, fminf(r2, r3)), mx = fmaxf(r1, fmaxf(r2, r3));
        if (!(__fsub_rn(mx, mn) > 2 || sc <= 0.5)) {
            if (__fsub_rn(sc, mx) > 0) sc = mx;
            else if (__fsub_rn(sc, mn) < 0) sc = mn;
            const float inten = __fmul_rn(vz, sc);
            if (inten > 2.0) res = inten;
        }
    }
    out[f] = res;
}

void vilf_feat_release(vilf_handle *h) {
    if (!h->feat) return;
    FeatCtx *c = h->feat;
    DBuf *all[] = {&c->pts, &c->keys, &c->keys2, &c->vals, &c->vals2, &c->temp, &c->rpts, &c->curv, &c->rstart, &c->etmp, &c->stmp, &c->ecnt, &c->scnt, &c->eoff, &c->soff, &c->oute, &c->outs, &c->tot, &c->dcloud, &c->dunit, &c->dfeat, &c->dout};
    for (DBuf *b : all) b->release();
    delete c;
    h->feat = nullptr;
}

extern "C" int vilf_lidar_extract_features(vilf_handle *h, const float *xyzi, int n, int n_scans, double min_range, double max_range, double edge_threshold,
                                           float *edge_out, int cap_edge, int *n_edge, float *surf_out, int cap_surf, int *n_surf) {
    if (!h || n < 0 || (n && !xyzi) || !n_edge || !n_surf || (n_scans != 16 && n_scans != 32 && n_scans != 64)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    *n_edge = 0; *n_surf = 0;
    if (n == 0) return VILF_OK;
    if (!h->feat) h->feat = new FeatCtx();
    FeatCtx *c = h->feat;
    const int nsec = n_scans * FE_SEC;
    const size_t sn = (size_t)n;
    if (n > c->cap || n_scans != c->scans) {
        if (!c->pts.ensure(sn * 16) || !c->keys.ensure(sn * 4) || !c->keys2.ensure(sn * 4) || !c->vals.ensure(sn * 4) || !c->vals2.ensure(sn * 4) || !c->rpts.ensure(sn * 16 + 256) ||
            !c->curv.ensure(sn * 8) || !c->rstart.ensure((n_scans + 2) * 4) || !c->etmp.ensure((size_t)nsec * 20 * 16) || !c->stmp.ensure((size_t)nsec * FE_MAXSEC * 16) ||
            !c->ecnt.ensure(nsec * 4) || !c->scnt.ensure(nsec * 4) || !c->eoff.ensure(nsec * 4) || !c->soff.ensure(nsec * 4) || !c->oute.ensure((size_t)nsec * 20 * 16) ||
            !c->outs.ensure(sn * 16) || !c->tot.ensure(64)) { h->err = "hipMalloc failed (feature extraction)"; return VILF_ERR_DEVICE; }
        const size_t need = vilf_sort_temp_bytes(sn, 4);
        if (!c->temp.ensure(need + 256)) return VILF_ERR_DEVICE;
        c->temp_bytes = c->temp.cap;
        c->cap = n; c->scans = n_scans;
    }
    const dim3 blk(256), grd((n + 255) / 256), grd1((n + 256) / 256);
    HIPCHECK(h, hipMemcpyAsync(c->pts.p, xyzi, sn * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemsetAsync(c->tot.p, 0, 64, h->stream));
    hipLaunchKernelGGL(fe_ring, grd, blk, 0, h->stream, c->pts.as<float4>(), n, n_scans, min_range, max_range, c->keys.as<unsigned int>(), c->vals.as<int>());
    size_t tb = c->temp_bytes;
    if (vilf_sort_pairs_u32(h->stream, c->temp.p, tb, c->keys.as<unsigned int>(), c->keys2.as<unsigned int>(), c->vals.as<int>(), c->vals2.as<int>(), sn, 7) != 0) { h->err = "feature extraction: radix sort failed"; return VILF_ERR_DEVICE; }
    hipLaunchKernelGGL(fe_ring_bounds, grd1, blk, 0, h->stream, c->keys2.as<unsigned int>(), n, n_scans, c->rstart.as<int>());
    hipLaunchKernelGGL(fe_gather_curv, grd, blk, 0, h->stream, c->pts.as<float4>(), c->vals2.as<int>(), c->rstart.as<int>(), n, n_scans, c->rpts.as<float4>());
    hipLaunchKernelGGL(fe_curv, grd, blk, 0, h->stream, c->rpts.as<float4>(), c->keys2.as<unsigned int>(), c->rstart.as<int>(), n, n_scans, c->curv.as<double>());
    hipLaunchKernelGGL(fe_sector, dim3(FE_SEC, n_scans), dim3(FE_NT), 0, h->stream, c->rpts.as<float4>(), c->curv.as<double>(), c->rstart.as<int>(), n_scans, edge_threshold,
                       c->etmp.as<float4>(), c->stmp.as<float4>(), c->ecnt.as<int>(), c->scnt.as<int>(), c->tot.as<int>() + 4);
    hipLaunchKernelGGL(fe_offsets, dim3(1), dim3(64), 0, h->stream, c->ecnt.as<int>(), c->scnt.as<int>(), nsec, c->eoff.as<int>(), c->soff.as<int>(), c->tot.as<int>());
    hipLaunchKernelGGL(fe_emit, dim3(nsec), dim3(256), 0, h->stream, c->etmp.as<float4>(), c->stmp.as<float4>(), c->ecnt.as<int>(), c->scnt.as<int>(), c->eoff.as<int>(), c->soff.as<int>(),
                       c->oute.as<float4>(), c->outs.as<float4>());
    HIPCHECK(h, hipGetLastError());
    int tot[8];
    HIPCHECK(h, hipMemcpyAsync(tot, c->tot.p, sizeof(tot), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (tot[4]) { h->err = "feature extraction: a sector holds more than 1024 points (ring with more than ~6150 returns)"; return VILF_ERR_UNSUPPORTED; }
    *n_edge = tot[0]; *n_surf = tot[1];
    if (edge_out && tot[0] > 0) HIPCHECK(h, hipMemcpyAsync(edge_out, c->oute.p, (size_t)std::min(tot[0], cap_edge) * 16, hipMemcpyDeviceToHost, h->stream));
    if (surf_out && tot[1] > 0) HIPCHECK(h, hipMemcpyAsync(surf_out, c->outs.p, (size_t)std::min(tot[1], cap_surf) * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}

extern "C" int vilf_feature_depth(vilf_handle *h, const float *cloud_xyzi, int n, const float *feat_xyz, int m, float *depth_out) {
    if (!h || n < 0 || m < 0 || (n && !cloud_xyzi) || (m && (!feat_xyz || !depth_out))) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    for (int i = 0; i < m; i++) depth_out[i] = -1.0f;
    if (m == 0 || n < 10) return VILF_OK;                         // "depth cloud is too few" (:97-101)
    if (!h->feat) h->feat = new FeatCtx();
    FeatCtx *c = h->feat;
    if (!c->dcloud.ensure((size_t)n * 16) || !c->dunit.ensure((size_t)n * 16) || !c->dfeat.ensure((size_t)m * 12) || !c->dout.ensure((size_t)m * 4)) { h->err = "hipMalloc failed (feature depth)"; return VILF_ERR_DEVICE; }
    const float bin_res = 180.0 / (float)360;
    const float thr = (float)std::pow(std::sin(bin_res / 180.0 * M_PI) * 5.0, 2);
    HIPCHECK(h, hipMemcpyAsync(c->dcloud.p, cloud_xyzi, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipMemcpyAsync(c->dfeat.p, feat_xyz, (size_t)m * 12, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fd_unit, dim3((n + 255) / 256), dim3(256), 0, h->stream, c->dcloud.as<float4>(), n, c->dunit.as<float4>());
    hipLaunchKernelGGL(fd_depth, dim3(m), dim3(256), 0, h->stream, c->dunit.as<float4>(), n, c->dfeat.as<float>(), m, thr, c->dout.as<float>());
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipMemcpyAsync(depth_out, c->dout.p, (size_t)m * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}

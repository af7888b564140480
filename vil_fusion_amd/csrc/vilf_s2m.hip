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
//                            -> ONE persistent 256-thread workgroup per stream: residual + jacobian of every factor, fixed-order reduction of
//                               the 6x6 normal equations, Cholesky, accept / reject, radius update (Ceres 2.0 LM semantics)
//   createSubMap (:298-352)  -> transform + append, crop-box compaction (order preserving), voxel grid
//
// The implementation is batched over S independent LiDAR streams (fixed-capacity segments, device-side counters, one global
// radix sort with the stream id in the high key bits); the reference's single EstimationMapping object is the S = 1 case.
#include <hip/hip_runtime.h>
#include <cstring>
#include <string.h>
#include <cmath>
#include <climits>
#include <algorithm>
#include <utility>
#include "vilf_sort.hpp"
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

// In-kernel phase stamps: diagnostic build only (make DEFS=-DVILF_STAMPS); workgroup S2M_STAMP_WG of a launch writes them, and only when
// its stream is a big one (the short launches of a size class would overwrite the interesting ones otherwise).
#ifdef VILF_STAMPS
__device__ long long s2m_dbg[9 * 32];
#define S2M_STAMP_WG 1500
#define S2M_STAMP(kid, i, cond) do { if (blockIdx.x == S2M_STAMP_WG && threadIdx.x == 0 && (cond)) s2m_dbg[(kid) * 32 + (i)] = __builtin_readcyclecounter(); } while (0)
extern "C" int vilf_debug_stamps_s2m(long long *out288) { return hipMemcpyFromSymbol(out288, HIP_SYMBOL(s2m_dbg), sizeof(long long) * 9 * 32) == hipSuccess ? 0 : -1; }
#define S2M_ACC_DECL long long acc_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}; long long acc_last = __builtin_readcyclecounter();
#define S2M_ACC(i) do { const long long now_ = __builtin_readcyclecounter(); acc_t[i] += now_ - acc_last; acc_last = now_; } while (0)
#define S2M_ACC_OUT(kid) do { if (blockIdx.x == S2M_STAMP_WG && threadIdx.x == 0) for (int i_ = 0; i_ < 8; i_++) s2m_dbg[(kid) * 32 + 16 + i_] = acc_t[i_]; } while (0)
#else
#define S2M_STAMP(kid, i, cond) do { } while (0)
#define S2M_ACC_DECL
#define S2M_ACC(i) do { } while (0)
#define S2M_ACC_OUT(kid) do { } while (0)
#endif

using namespace vd;

// ---------------------------------------------------------------------------------------------------------------------
// Batched layout. S independent LiDAR streams (S = 1 for the single-stream ABI) live side by side in fixed-capacity segments:
//   cloud set   pts[S][cap] (float4 xyzi), n[S] (device counters)
// Every kernel is launched over (ceil(cap / 256), S) or S workgroups and reads its stream's count from device memory, so a whole
// step (voxel grids, index build, 2 x (associate + LM solve), sub-map maintenance) is enqueued without a host round trip.
// Sorts are ONE rocPRIM radix sort over all segments with the stream id in the high key bits; padding entries carry the largest
// key of their stream and stay at the end of their own segment.
struct CSet { float4 *p; int *n; int cap; const int *smap = nullptr; };      // smap (general voxel path only): grid row -> stream, for a sub-batch of streams
__device__ __forceinline__ int cset_sid(const CSet &c, int ls) { return c.smap ? c.smap[ls] : ls; }
struct MinMax { float mn[3], mx[3]; int minb[3]; int pad_; long long mul1, mul2; int divb[3]; int pad2_; };
struct S2BRes {                       // per-stream result of one step (device)
    double pose[7], prev[7];
    double cost[2];
    int n_ds[2], nfe[2], nfs[2], its[2], map_n[2];
    int err, do_opt;
};
// Sort keys are as narrow as the data allows: the leaf-index width / the per-axis cell widths are reduced over all streams on the
// device (b_minmax) and read back (16 bytes) before the keys are built, so a pass of the radix sort is not spent on zero bits.
#define S2B_ERR_MAPCAP 1
#define S2B_ERR_EXTENT 2
#define S2B_ERR_VOXEL 4

__device__ __forceinline__ int bits_of(long long v) { return v <= 0 ? 0 : 64 - __clzll(v); }
// bits[0]: width of the largest leaf index (+1) over all streams; bits[1..3]: widths of the per-axis leaf extents; bits[4..6]: the largest per-axis leaf extents
__global__ void b_minmax(CSet in, float inv, MinMax *mm, int *bits) {
    __shared__ float s[6][1024];
    const int tid = threadIdx.x, ls = blockIdx.x, sid = cset_sid(in, ls), n = in.n[sid];
    const float4 *p = in.p + (size_t)sid * in.cap;
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
        MinMax *out = mm + ls;
        int divb[3];
        for (int k = 0; k < 3; k++) {
            out->mn[k] = s[k][0]; out->mx[k] = s[3 + k][0];
            out->minb[k] = (int)floorf(__fmul_rn(s[k][0], inv));
            divb[k] = (int)floorf(__fmul_rn(s[3 + k][0], inv)) - out->minb[k] + 1;
            out->divb[k] = divb[k];
        }
        out->mul1 = divb[0]; out->mul2 = (long long)divb[0] * divb[1];
        if (n > 0) {
            atomicMax(bits, bits_of((long long)divb[0] * divb[1] * divb[2]));
            for (int k = 0; k < 3; k++) { atomicMax(bits + 1 + k, bits_of(divb[k])); atomicMax(bits + 4 + k, divb[k]); }     // [4..6]: the largest leaf extents (cell-major keys, cell size of an unordered map)
        }
    }
}
// cs < 0: pcl::VoxelGrid's leaf index x + y dx + z dx dy relative to the cloud's min corner; cs >= 0: the cell-major order of the local maps (mu_leaf: cy | cx | z |
// y_low | x_low with cells of 2^cs x 2^cs leaf columns on the ABSOLUTE leaf grid), made narrow the same way: cell coordinates relative to the cloud's first cell
template <typename KeyT>
__global__ void b_voxel_keys(CSet in, float inv, const MinMax *mm, KeyT *keys, int *vals, int *err, int vbits, int cs) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, ls = blockIdx.y, sid = cset_sid(in, ls);
    if (i >= in.cap) return;
    const size_t g = (size_t)ls * in.cap + i;                   // keys / indices are laid out by grid row, the points by stream
    unsigned long long k = (1ULL << vbits) - 1;                 // padding: larger than every leaf index of the stream
    if (i < in.n[sid]) {
        const float4 q = in.p[(size_t)sid * in.cap + i];
        const MinMax *m = mm + ls;
        const long long a = (long long)floorf(__fmul_rn(q.x, inv)) - m->minb[0], b = (long long)floorf(__fmul_rn(q.y, inv)) - m->minb[1], c = (long long)floorf(__fmul_rn(q.z, inv)) - m->minb[2];
        if (cs < 0) k = (unsigned long long)(a + b * m->mul1 + c * m->mul2);
        else {
            const long long ixa = a + m->minb[0], iya = b + m->minb[1], lm = (1LL << cs) - 1;
            const long long cminx = (long long)m->minb[0] >> cs, cminy = (long long)m->minb[1] >> cs;
            const long long ncx = (((long long)m->minb[0] + m->divb[0] - 1) >> cs) - cminx + 1;
            const long long cell = ((iya >> cs) - cminy) * ncx + ((ixa >> cs) - cminx);
            k = (unsigned long long)((((cell * m->divb[2] + c) << cs | (iya & lm)) << cs) | (ixa & lm));
        }
    }
    keys[g] = (KeyT)(((unsigned long long)ls << vbits) | k);
    vals[g] = i;
}
// Fused heads + scan + centroids: ONE workgroup per stream walks its sorted (leaf, index) pairs in tiles of 1024, ranks the run
// heads with a block scan and lets each head thread sum its run in sorted (= input) order — same float sums as pcl::VoxelGrid.
#define S2B_VT 1024
template <int NW>            // workgroup of NW waves
__device__ __forceinline__ int block_excl_scan_nw(int v, int *s_w, int &total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    __syncthreads();
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { const int x = s_w[k]; if (k < wave) off += x; tot += x; }
    total = tot;
    return off + incl - v;
}
__device__ __forceinline__ int block_excl_scan_1024(int v, int *s_w, int &total) { return block_excl_scan_nw<16>(v, s_w, total); }
// Two launches over (tile, stream): b_voxel_heads counts the run heads of every 1024-pair tile, b_voxel_reduce turns the counts of the
// stream's earlier tiles into its output offset (a handful of ints) and then works on its tile alone — every tile of every stream is
// an independent workgroup, instead of one workgroup walking a stream's tiles one after the other.
template <typename KeyT>
__global__ __launch_bounds__(S2B_VT) void b_voxel_heads(CSet in, const KeyT *keys_all, int *tile_heads, int ntiles) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, ls = blockIdx.y, sid = cset_sid(in, ls), t = blockIdx.x, n = in.n[sid];
    const int i = t * S2B_VT + tid;
    int head = 0;
    if (t * S2B_VT < n) {
        const KeyT *keys = keys_all + (size_t)ls * in.cap;
        const int ic = min(i, n - 1);
        const KeyT k = keys[ic], kprev = keys[max(ic - 1, 0)];
        head = (i < n && (i == 0 || kprev != k)) ? 1 : 0;
    }
    int total;
    block_excl_scan_1024(head, s_w, total);
    if (tid == 0) tile_heads[(size_t)ls * ntiles + t] = total;
}
template <typename KeyT>
__global__ __launch_bounds__(S2B_VT) void b_voxel_reduce(CSet in, const KeyT *keys_all, const int *vals_all, CSet out, const int *tile_heads, int ntiles) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, ls = blockIdx.y, sid = cset_sid(in, ls), t = blockIdx.x, n = in.n[sid];
    const int t0 = t * S2B_VT;
    if (t0 >= n) { if (t == 0 && tid == 0) out.n[sid] = 0; return; }
    const size_t base = (size_t)ls * in.cap;
    const KeyT *keys = keys_all + base;
    const int *vals = vals_all + base;
    const float4 *p = in.p + (size_t)sid * in.cap;
    float4 *o = out.p + (size_t)sid * out.cap;
    const int i = t0 + tid, ic = min(i, n - 1);
    // everything a lane needs for its own element is requested up front (clamped, unconditional loads): the keys before and
    // after, the point index, the point. A leaf with one point then needs no further memory round trip after the scan.
    const KeyT k = keys[ic], kprev = keys[max(ic - 1, 0)], knext = keys[min(ic + 1, n - 1)];
    const float4 q0 = p[vals[ic]];
    int carry = 0;
    {
        const int *th = tile_heads + (size_t)ls * ntiles;
        for (int u = tid; u < t; u += S2B_VT) carry += th[u];
        if (t > 1) { int tot; block_excl_scan_1024(carry, s_w, tot); carry = tot; __syncthreads(); }   // t <= 1: only lane 0 holds a value ...
        else carry = t == 1 ? th[0] : 0;                                                                // ... which every lane can read itself
    }
    const int head = (i < n && (i == 0 || kprev != k)) ? 1 : 0;
    int total;
    const int pos = carry + block_excl_scan_1024(head, s_w, total);
    if (head) {
        float cx = __fadd_rn(0.0f, q0.x), cy = __fadd_rn(0.0f, q0.y), cz = __fadd_rn(0.0f, q0.z), ci = __fadd_rn(0.0f, q0.w); int cnt = 1;   // accumulate from +0 like the reference (keeps the sign of a zero sum)
        if (i + 1 < n && knext == k)
            for (int j = i + 1; j < n && keys[j] == k; j++) { const float4 q = p[vals[j]]; cx = __fadd_rn(cx, q.x); cy = __fadd_rn(cy, q.y); cz = __fadd_rn(cz, q.z); ci = __fadd_rn(ci, q.w); cnt++; }
        const float nn = (float)cnt;
        o[pos] = make_float4(cx / nn, cy / nn, cz / nn, ci / nn);
    }
    if (t0 + S2B_VT >= n && tid == 0) out.n[sid] = carry + total;
}
// Fused crop flags + scan + compaction (pcl::CropBox, order preserving): one workgroup per stream
// m_out[s] = how many of the first n_old[s] points (the old map, before the append) survive the crop
__global__ __launch_bounds__(S2B_VT) void b_crop_compact(CSet map, const double *pose_all, double half, CSet out, const int *n_old, int *m_out) {
    __shared__ int s_w[16];
    const int tid = threadIdx.x, sid = blockIdx.x, n = map.n[sid], nold = n_old[sid];
    if (tid == 0 && nold <= 0) m_out[sid] = 0;
    const double *pose = pose_all + 24 * sid;
    const float4 *p = map.p + (size_t)sid * map.cap;
    float4 *o = out.p + (size_t)sid * out.cap;
    const float mnx = (float)(pose[4] - half), mny = (float)(pose[5] - half), mnz = (float)(pose[6] - half);
    const float mxx = (float)(pose[4] + half), mxy = (float)(pose[5] + half), mxz = (float)(pose[6] + half);
    int carry = 0;
    for (int t0 = 0; t0 < n; t0 += S2B_VT) {
        const int i = t0 + tid;
        float4 q = make_float4(0, 0, 0, 0);
        int f = 0;
        if (i < n) { q = p[i]; f = !(q.x < mnx || q.y < mny || q.z < mnz || q.x > mxx || q.y > mxy || q.z > mxz) ? 1 : 0; }
        int total;
        const int pos = carry + block_excl_scan_1024(f, s_w, total);
        if (f) o[pos] = q;
        if (i == nold - 1) m_out[sid] = pos + f;
        carry += total;
    }
    if (tid == 0) out.n[sid] = carry;
}

// ---- pcl::VoxelGrid of one scan cloud entirely inside one workgroup -------------------------------------------------------------
// A raw scan cloud (<= ~19 k points) fits the LDS as 32-bit leaf keys + two 16-bit index arrays, so its voxel grid needs no global
// sort, no key / value arrays in HBM and no key-width agreement between streams (no host round trip): the points are read twice from
// HBM / L2 (bounding box, keys), the indices are radix-sorted in LDS (8-bit digits, only as many passes as this cloud's key has bits;
// stable: every wave owns a contiguous segment and ranks a 64-lane strip by ballot matching), and each leaf is summed in index order.
#define SV_T 1024
#define SV_MAXPTS32 19200              // 32-bit keys: 8 bytes of LDS per point
#define SV_MAXPTS24 22112              // 24 stored key bits: 7 bytes per point (a wider key's top byte is recomputed from the point when needed): 7 x 22112 + 8192 + 736 B of static LDS = 163 712 of the CU's 163 840 bytes
                                       // (22000 until round 5: the bench's largest surf cloud has 22015 points and took the global-sort path — the vendor radix sort — in every step)
__device__ __forceinline__ int sv_bits(int v) { return v <= 1 ? 0 : 32 - __clz(v - 1); }     // bits for values 0 .. v-1
// c += s_tq[b], s_tq[b + 1], ... (cnt points, in this order: the serial float chain pcl's accumulation defines). The next eight points are already on their way from
// LDS while eight are added, so a leaf of a few hundred near-range points costs a few cycles per point instead of an LDS round trip.
#define SV_ADD4(X) _Pragma("unroll") for (int v = 0; v < 4; v++) { cx = __fadd_rn(cx, X[v].x); cy = __fadd_rn(cy, X[v].y); cz = __fadd_rn(cz, X[v].z); ci = __fadd_rn(ci, X[v].w); }
#define SV_LD4(X, i) _Pragma("unroll") for (int v = 0; v < 4; v++) X[v] = q[4 * (i) + v];
__device__ __forceinline__ void sv_run_sum(const float4 *s_tq, int b, int cnt, int tile_n, float &cx, float &cy, float &cz, float &ci) {
    const int nfull = cnt >> 2;
    if (nfull > 0) {            // full batches of four: no predication on the add chain, two batches in flight (three register sets taking turns)
        const float4 *q = s_tq + b;
        float4 S0[4], S1[4], S2[4];
        SV_LD4(S0, 0)
        if (nfull > 1) { SV_LD4(S1, 1) }
        for (int i = 0;;) {
            if (i + 2 < nfull) { SV_LD4(S2, i + 2) }
            SV_ADD4(S0)
            if (++i >= nfull) break;
            if (i + 2 < nfull) { SV_LD4(S0, i + 2) }
            SV_ADD4(S1)
            if (++i >= nfull) break;
            if (i + 2 < nfull) { SV_LD4(S1, i + 2) }
            SV_ADD4(S2)
            if (++i >= nfull) break;
        }
    }
    const int k = nfull << 2;
    if (k < cnt) {              // the last 1..3 points
        float4 R[3];
#pragma unroll
        for (int v = 0; v < 3; v++) R[v] = s_tq[min(b + k + v, tile_n - 1)];
#pragma unroll
        for (int v = 0; v < 3; v++) if (k + v < cnt) { cx = __fadd_rn(cx, R[v].x); cy = __fadd_rn(cy, R[v].y); cz = __fadd_rn(cz, R[v].z); ci = __fadd_rn(ci, R[v].w); }
    }
}
template <bool K24>
__device__ __forceinline__ void sv_body(const int sid, const CSet &in, float inv, const CSet &out, int cap, int *err) {
    extern __shared__ unsigned int sv_lds[];
    // K24: lo16[cap] a[cap] b[cap] (u16) hi8[cap] (u8) cnt;   else: key32[cap] a[cap] b[cap] cnt      (cap is a multiple of 16)
    unsigned int *s_key = sv_lds;
    unsigned short *s_lo = reinterpret_cast<unsigned short *>(sv_lds);
    unsigned short *s_a = K24 ? s_lo + cap : reinterpret_cast<unsigned short *>(s_key + cap), *s_b = s_a + cap;
    unsigned char *s_hi = reinterpret_cast<unsigned char *>(s_b + cap);
    unsigned short *s_cnt = K24 ? reinterpret_cast<unsigned short *>(s_hi + cap) : s_b + cap;     // [256 digits][16 waves]
    __shared__ float s_mm[6][16];
    __shared__ int s_w2[2][16], s_geo[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = in.n[sid];
    if (n > cap) return;                                   // a larger cloud: the global-sort path takes this stream (s2b_step)
    const float4 *p = in.p + (size_t)sid * in.cap;
    float4 *o = out.p + (size_t)sid * out.cap;
    if (n <= 0) { if (tid == 0) out.n[sid] = 0; return; }
    S2M_STAMP(K24 ? 0 : 1, 0, true);
    // ---- A. bounding box
    float mn[3] = {3.0e38f, 3.0e38f, 3.0e38f}, mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int i = tid; i < n; i += SV_T) {
        const float4 q = p[i];
        mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
        mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], of, 64)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], of, 64)); }
        if (lane == 0) { s_mm[k][wave] = mn[k]; s_mm[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (tid == 0) {
        int vb = 0;
        for (int k = 0; k < 3; k++) {
            float a = s_mm[k][0], b = s_mm[3 + k][0];
            for (int w2 = 1; w2 < 16; w2++) { a = fminf(a, s_mm[k][w2]); b = fmaxf(b, s_mm[3 + k][w2]); }
            const int minb = (int)floorf(__fmul_rn(a, inv)), divb = (int)floorf(__fmul_rn(b, inv)) - minb + 1;
            s_geo[k] = minb; s_geo[3 + k] = sv_bits(divb);
            vb += s_geo[3 + k];
        }
        s_geo[6] = vb;
        if (vb > 32) atomicOr(err + sid, S2B_ERR_VOXEL);
    }
    __syncthreads();
    const int mb0 = s_geo[0], mb1 = s_geo[1], mb2 = s_geo[2], bx = s_geo[3], by = s_geo[4], vbits = min(s_geo[6], 32);
    // leaf coordinates packed z | y | x (the order of pcl's leaf index)
    auto key_of = [&](const float4 q) -> unsigned int {
        const unsigned int a = (unsigned int)((int)floorf(__fmul_rn(q.x, inv)) - mb0), b = (unsigned int)((int)floorf(__fmul_rn(q.y, inv)) - mb1), c = (unsigned int)((int)floorf(__fmul_rn(q.z, inv)) - mb2);
        return (bx + by < 32 ? (c << (bx + by)) : 0u) | (b << bx) | a;
    };
    auto key_at = [&](int idx) -> unsigned int {                   // the full key of point idx
        if (!K24) return s_key[idx];
        unsigned int k = (unsigned int)s_lo[idx] | ((unsigned int)s_hi[idx] << 16);
        if (vbits > 24) k |= key_of(p[idx]) & 0xff000000u;         // rare: the top byte is not stored
        return k;
    };
    auto digit_at = [&](int idx, int shift) -> int {
        if (!K24) return (int)((s_key[idx] >> shift) & 255);
        if (shift == 0) return s_lo[idx] & 255;
        if (shift == 8) return s_lo[idx] >> 8;
        if (shift == 16) return s_hi[idx];
        return (int)(key_of(p[idx]) >> 24);
    };
    S2M_STAMP(K24 ? 0 : 1, 1, true);
    // ---- B. keys, index array = identity
    for (int i = tid; i < n; i += SV_T) {
        const unsigned int k = key_of(p[i]);
        if (K24) { s_lo[i] = (unsigned short)(k & 0xffffu); s_hi[i] = (unsigned char)((k >> 16) & 0xffu); } else s_key[i] = k;
        s_a[i] = (unsigned short)i;
    }
    __syncthreads();
    S2M_STAMP(K24 ? 0 : 1, 2, true);
    // ---- C. stable LSD radix sort of the indices by key, 8 bits per pass
    unsigned short *src = s_a, *dst = s_b;
    const int seg = (n + 15) >> 4, lo = wave * seg, hi = min(n, lo + seg);
    unsigned int *cnt32 = reinterpret_cast<unsigned int *>(s_cnt);
    for (int shift = 0; shift < vbits; shift += 8) {
        for (int t = tid; t < 2048; t += SV_T) cnt32[t] = 0;
        __syncthreads();
        for (int base = lo; base < hi; base += 64) {       // per-wave digit counts: lanes with the same digit elect a leader (ballot matching) — neighbouring
            const int j = base + lane;                     // points share leaves, so per-lane LDS atomics on one counter would serialise
            const bool valid = j < hi;
            const int d = valid ? digit_at(src[j], shift) : 0;
            unsigned long long mask = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1); mask &= ((d >> b) & 1) ? bal : ~bal; }
            if (valid && (mask & ((1ULL << lane) - 1ULL)) == 0) s_cnt[d * 16 + wave] += (unsigned short)__popcll(mask);
        }
        __syncthreads();
        {   // exclusive scan in (digit, wave) order: thread t owns entries 4 t .. 4 t + 3 of [digit][wave]
            int c4[4], loc = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) { c4[u] = s_cnt[4 * tid + u]; loc += c4[u]; }
            int tot;
            int run = block_excl_scan_1024(loc, s_w2[0], tot);
#pragma unroll
            for (int u = 0; u < 4; u++) { s_cnt[4 * tid + u] = (unsigned short)run; run += c4[u]; }
        }
        __syncthreads();
        for (int base = lo; base < hi; base += 64) {       // a wave walks its own segment in order: its counter column needs no synchronisation
            const int j = base + lane;
            const bool valid = j < hi;
            const unsigned short idx = valid ? src[j] : (unsigned short)0;
            const int d = valid ? digit_at(idx, shift) : 0;
            unsigned long long mask = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1); mask &= ((d >> b) & 1) ? bal : ~bal; }
            if (valid) {
                const int rank = __popcll(mask & ((1ULL << lane) - 1ULL));
                const int off = s_cnt[d * 16 + wave];
                dst[off + rank] = idx;
                if (rank == 0) s_cnt[d * 16 + wave] = (unsigned short)(off + __popcll(mask));
            }
        }
        __syncthreads();
        unsigned short *t = src; src = dst; dst = t;
    }
    S2M_STAMP(K24 ? 0 : 1, 3, true);
    // ---- D. run heads, output ranks (two sorted positions per lane and tile), centroids in index order. The tile's points and keys are staged
    // in LDS (the index buffer the sort no longer needs, the counter array): a leaf with dozens of points — dense near-range ground — is then
    // summed from LDS by its head lane instead of through a chain of dependent global gathers that the whole workgroup would wait for.
    float4 *s_tq = (cap * 2 >= 2 * SV_T * 16) ? reinterpret_cast<float4 *>(dst) : reinterpret_cast<float4 *>(reinterpret_cast<unsigned char *>(s_cnt) + 8192);
    __shared__ float s_open_f[2][4];
    __shared__ int s_open_i[2][4], s_fh[32];       // open leaf: valid, points so far, -, output slot; first head position of every 64-element strip of the tile
    if (tid == 0) s_open_i[0][0] = 0;
    int carry = 0, par = 0;
    float4 qn[2];
#pragma unroll
    for (int u = 0; u < 2; u++) qn[u] = p[src[min(u * SV_T + tid, n - 1)]];
    S2M_ACC_DECL
    for (int t0 = 0; t0 < n; t0 += 2 * SV_T) {
        S2M_ACC(0);
        int head[2], incl[2];
        unsigned int key[2];
        unsigned long long hb[2];
        float4 q0[2];
        const int tile_n = min(2 * SV_T, n - t0);
#pragma unroll
        for (int u = 0; u < 2; u++) { q0[u] = qn[u]; qn[u] = p[src[min(t0 + 2 * SV_T + u * SV_T + tid, n - 1)]]; }     // the next tile's gathers are in flight during this tile
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int j = t0 + u * SV_T + tid, jc = min(j, n - 1);
            const unsigned short idx = src[jc];
            key[u] = key_at(idx);
            head[u] = (j < n && (j == 0 || key_at(src[max(jc - 1, 0)]) != key[u])) ? 1 : 0;
            hb[u] = __ballot(head[u]);                                          // the heads of this 64-element strip: ranks and run ends come from the mask
            incl[u] = __popcll(hb[u] & ((2ULL << lane) - 1ULL));
        }
#pragma unroll
        for (int u = 0; u < 2; u++) s_tq[u * SV_T + tid] = q0[u];
        if (lane == 0) {
#pragma unroll
            for (int u = 0; u < 2; u++) { s_w2[u][wave] = __popcll(hb[u]); s_fh[u * 16 + wave] = hb[u] ? (u * 16 + wave) * 64 + __ffsll((long long)hb[u]) - 1 : 0x7fffffff; }
        }
        if (tid == 0) s_open_i[par ^ 1][0] = 0;
        S2M_ACC(1);
        __syncthreads();
        S2M_ACC(2);
        // the leaf left open by the previous tile (its run reached the tile end): thread 0 adds this tile's leading points of the same leaf — everything before the
        // tile's first head — in index order and closes it, or hands it on when the whole tile belongs to it. No dependent global gathers, no key compares.
        if (tid == 0 && s_open_i[par][0]) {
            float cx = s_open_f[par][0], cy = s_open_f[par][1], cz = s_open_f[par][2], ci = s_open_f[par][3];
            int m0 = 0x7fffffff;
            for (int k = 0; k < 32 && m0 == 0x7fffffff; k++) m0 = s_fh[k];
            m0 = min(m0, tile_n);
            sv_run_sum(s_tq, 0, m0, tile_n, cx, cy, cz, ci);
            const int len = s_open_i[par][1] + m0;
            if (m0 == tile_n && t0 + tile_n < n) {
                s_open_f[par ^ 1][0] = cx; s_open_f[par ^ 1][1] = cy; s_open_f[par ^ 1][2] = cz; s_open_f[par ^ 1][3] = ci;
                s_open_i[par ^ 1][1] = len; s_open_i[par ^ 1][3] = s_open_i[par][3]; s_open_i[par ^ 1][0] = 1;
            } else {
                const float nn = (float)len;
                o[s_open_i[par][3]] = make_float4(cx / nn, cy / nn, cz / nn, ci / nn);
            }
        }
        int base = carry;
#pragma unroll
        for (int u = 0; u < 2; u++) {
            int off = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < 16; k++) { const int x = s_w2[u][k]; if (k < wave) off += x; tot += x; }
            if (head[u]) {
                const int e = u * SV_T + tid, j = t0 + e, strip = u * 16 + wave;
                // the run ends at the next head: in this strip (mask), else the first head of a later strip, else the tile end
                const unsigned long long rest = lane < 63 ? hb[u] & ~((2ULL << lane) - 1ULL) : 0ULL;
                int re = rest ? strip * 64 + __ffsll((long long)rest) - 1 : 0x7fffffff;
                for (int k = strip + 1; k < 32 && re == 0x7fffffff; k++) re = s_fh[k];
                const int len = min(re, tile_n) - e;
                float cx = __fadd_rn(0.0f, q0[u].x), cy = __fadd_rn(0.0f, q0[u].y), cz = __fadd_rn(0.0f, q0[u].z), ci = __fadd_rn(0.0f, q0[u].w);
                sv_run_sum(s_tq, e + 1, len - 1, tile_n, cx, cy, cz, ci);
                const int slot = base + off + incl[u] - 1;
                if (e + len == tile_n && j + len < n) {      // the run reaches the tile end and points remain: the leaf may continue in the next tile -> left open
                    s_open_f[par ^ 1][0] = cx; s_open_f[par ^ 1][1] = cy; s_open_f[par ^ 1][2] = cz; s_open_f[par ^ 1][3] = ci;
                    s_open_i[par ^ 1][1] = len; s_open_i[par ^ 1][3] = slot; s_open_i[par ^ 1][0] = 1;
                } else {
                    const float nn = (float)len;
                    o[slot] = make_float4(cx / nn, cy / nn, cz / nn, ci / nn);
                }
            }
            base += tot;
        }
        par ^= 1;
        carry = base;
        S2M_ACC(3);
        __syncthreads();                         // the staging array, s_w2 and s_fh are rewritten by the next tile
        S2M_ACC(4);
    }
    S2M_ACC_OUT(K24 ? 0 : 1);
    S2M_STAMP(K24 ? 0 : 1, 4, true);
#ifdef VILF_STAMPS
    if (blockIdx.x == S2M_STAMP_WG && tid == 0) { s2m_dbg[(K24 ? 0 : 1) * 32 + 30] = n; s2m_dbg[(K24 ? 0 : 1) * 32 + 31] = vbits; }
#endif
    if (tid == 0) out.n[sid] = carry;
}
// list == nullptr: one workgroup per stream. Otherwise the streams b_scan_voxel_runs (below) handed back — list[0] of them, list[1..] — are shared out over the grid.
template <bool K24>
__global__ __launch_bounds__(SV_T) void b_scan_voxel(CSet in, float inv, CSet out, int cap, int *err, const int *list) {
    if (!list) { sv_body<K24>(blockIdx.x, in, inv, out, cap, err); return; }
    const int cnt = list[0];
    for (int li = blockIdx.x; li < cnt; li += gridDim.x) {
        sv_body<K24>(list[1 + li], in, inv, out, cap, err);
        __syncthreads();                                   // the next stream rewrites the LDS state
    }
}

// ---- pcl::VoxelGrid of a scan cloud from its RUNS --------------------------------------------------------------------------------
// A scan arrives ring after ring, azimuth ascending: a point mostly falls into the leaf of its predecessor (KITTI-like surf clouds: ~4 points per run, ~1.5 runs
// per leaf). A RUN — consecutive points with one leaf — is what gets sorted here, not the point: the stable sort by key of (key, first index) keeps every leaf's
// runs in scan order, and the points of a run are consecutive in memory, so walking a leaf's runs adds its points in index order: the same serial float chain as sv_body
// (and as pcl's accumulation), from a quarter of the sort and from coalesced reads instead of per-point gathers. The LDS holds 10 bytes per RUN (32-bit key, 16-bit
// first index, two 16-bit index arrays) instead of 7-8 per POINT: two workgroups of eight waves per CU, one's point passes (HBM) beside the other's sort (LDS).
// The points are read twice (runs, sums), not three times: a run is detected on ABSOLUTE leaf coordinates (kept as 16-bit offsets from the first point's leaf), the
// bounding box is the minimum / maximum of those (floor(x * inv) is monotonic: the leaf of the smallest x is the smallest leaf), and the keys follow from both in LDS.
// A cloud with more than rc runs (no scan order: random test clouds), a key wider than 32 bits or leaves further than 2^15 from the first point's is handed back:
// its stream goes on `list`, which b_scan_voxel takes.
#define SR_T 512
#define SR_NW (SR_T / 64)
#define SR_E 4
#define SR_RC 7168
#define SR_RPT (SR_RC / SR_T)              // runs per thread in the coordinate -> key pass
#define SR_SPW (SR_RC / SR_NW / 64)        // 64-run strips of a wave's segment in the sort
__device__ float4 sr_neutral = {-0.0f, -0.0f, -0.0f, -0.0f};      // what an empty slot of a batch loads: the adds of the leaf sums are unconditional
static size_t sr_lds_bytes(int rc) { return (size_t)10 * rc + 4 + 2 * 256 * SR_NW; }
__global__ __launch_bounds__(SR_T, 4) void b_scan_voxel_runs(CSet in, float inv, CSet out, int maxn, int rc, int *list, int kid) {
    extern __shared__ unsigned int sr_lds[];
    unsigned int *s_rkey = sr_lds;                                                          // [rc]        leaf key of run r (runs in scan order); before the keys exist: the z offsets (16-bit)
    unsigned short *s_rstart = reinterpret_cast<unsigned short *>(s_rkey + rc);             // [rc + 2]    index of its first point; [nruns] = n
    unsigned short *s_a = s_rstart + rc + 2, *s_b = s_a + rc;                               // [rc] each   run ids, ping-pong of the sort (before: the x and y offsets; afterwards: first sorted position of every leaf)
    unsigned short *s_cnt = s_b + rc;                                                       // [256 digits][SR_NW waves]
    unsigned short *s_z16 = reinterpret_cast<unsigned short *>(s_rkey);
    __shared__ int s_mm[6][SR_NW];
    __shared__ int s_w2[2][SR_E][SR_NW], s_geo[8], s_sc[SR_NW], s_bad;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sid = blockIdx.x;
    const int n = in.n[sid];
    if (n > maxn) return;                                   // a larger cloud: the global-sort path takes this stream (s2b_step)
    const float4 *p = in.p + (size_t)sid * in.cap;
    float4 *o = out.p + (size_t)sid * out.cap;
    if (n <= 0) { if (tid == 0) out.n[sid] = 0; return; }
    (void)kid;
    S2M_STAMP(kid, 0, true);
    if (tid == 0) s_bad = 0;
    // ---- A. one pass over the points: leaf coordinates, runs (a point whose leaf differs from its predecessor's opens one; run ids in scan order from a running count
    // of the heads), bounding box of the leaves. The next tile's points are in flight.
    int o0, o1, o2;
    { const float4 q0 = p[0]; o0 = (int)floorf(__fmul_rn(q0.x, inv)) - 32768; o1 = (int)floorf(__fmul_rn(q0.y, inv)) - 32768; o2 = (int)floorf(__fmul_rn(q0.z, inv)) - 32768; }
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    int carry = 0, par = 0;
    {
        // (lane 0's predecessor sits in another wave: its point is requested with the tile, one lane's load of a line that is on its way anyway. A load at the point of
        //  use would be the newest one in flight: waiting for it waits for the whole prefetch — memory returns in order — a round trip per strip, as first written)
        float4 qn[SR_E], qpn[SR_E], qn2[SR_E], qpn2[SR_E];              // two tiles in flight: one ahead left every tile waiting a memory round trip
#pragma unroll
        for (int u = 0; u < SR_E; u++) {
            const int i0 = u * SR_T + tid, i1 = SR_E * SR_T + i0;
            qn[u] = p[min(i0, n - 1)]; qpn[u] = qn[u]; if (lane == 0) qpn[u] = p[min(max(i0 - 1, 0), n - 1)];
            qn2[u] = p[min(i1, n - 1)]; qpn2[u] = qn2[u]; if (lane == 0) qpn2[u] = p[min(i1 - 1, n - 1)];
        }
        for (int t0 = 0; t0 < n; t0 += SR_E * SR_T) {
            int lx[SR_E], ly[SR_E], lz[SR_E], head[SR_E];
            unsigned long long hb[SR_E];
            float4 q[SR_E], qp[SR_E];
#pragma unroll
            for (int u = 0; u < SR_E; u++) {
                q[u] = qn[u]; qp[u] = qpn[u]; qn[u] = qn2[u]; qpn[u] = qpn2[u];
                const int inx = t0 + 2 * SR_E * SR_T + u * SR_T + tid;
                qn2[u] = p[min(inx, n - 1)];
                if (lane == 0) qpn2[u] = p[min(inx - 1, n - 1)];
            }
#pragma unroll
            for (int u = 0; u < SR_E; u++) {
                const int i = t0 + u * SR_T + tid;
                lx[u] = (int)floorf(__fmul_rn(q[u].x, inv)) - o0; ly[u] = (int)floorf(__fmul_rn(q[u].y, inv)) - o1; lz[u] = (int)floorf(__fmul_rn(q[u].z, inv)) - o2;
                // (x and y offsets as one word: exact while they fit 16 bits, and a cloud where they do not is handed back below)
                const int xy = (lx[u] & 0xffff) | (ly[u] << 16);
                int pxy = __shfl_up(xy, 1, 64), pz = __shfl_up(lz[u], 1, 64);
                if (lane == 0) { pxy = (((int)floorf(__fmul_rn(qp[u].x, inv)) - o0) & 0xffff) | (((int)floorf(__fmul_rn(qp[u].y, inv)) - o1) << 16); pz = (int)floorf(__fmul_rn(qp[u].z, inv)) - o2; }
                head[u] = (i < n && (i == 0 || pxy != xy || pz != lz[u])) ? 1 : 0;
                hb[u] = __ballot(head[u]);
                if (i < n) {
                    mn[0] = min(mn[0], lx[u]); mn[1] = min(mn[1], ly[u]); mn[2] = min(mn[2], lz[u]);
                    mx[0] = max(mx[0], lx[u]); mx[1] = max(mx[1], ly[u]); mx[2] = max(mx[2], lz[u]);
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int u = 0; u < SR_E; u++) s_w2[par][u][wave] = __popcll(hb[u]);
            }
            __syncthreads();
            int base = carry;
#pragma unroll
            for (int u = 0; u < SR_E; u++) {
                int off = 0, tot = 0;
#pragma unroll
                for (int k = 0; k < SR_NW; k++) { const int x = s_w2[par][u][k]; if (k < wave) off += x; tot += x; }
                if (head[u]) {
                    const int r = base + off + __popcll(hb[u] & ((1ULL << lane) - 1ULL));
                    if (r < rc) { s_a[r] = (unsigned short)lx[u]; s_b[r] = (unsigned short)ly[u]; s_z16[r] = (unsigned short)lz[u]; s_rstart[r] = (unsigned short)(t0 + u * SR_T + tid); }
                }
                base += tot;
            }
            carry = base; par ^= 1;                        // (s_w2 is double-buffered: one barrier per tile)
        }
    }
    const int nruns = carry;
    S2M_STAMP(kid, 1, true);
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], of, 64)); mx[k] = max(mx[k], __shfl_xor(mx[k], of, 64)); }
        if (lane == 0) { s_mm[k][wave] = mn[k]; s_mm[3 + k][wave] = mx[k]; }
    }
    __syncthreads();
    if (tid == 0) {
        int vb = 0, far = 0;
        for (int k = 0; k < 3; k++) {
            int a = s_mm[k][0], b = s_mm[3 + k][0];
            for (int w2 = 1; w2 < SR_NW; w2++) { a = min(a, s_mm[k][w2]); b = max(b, s_mm[3 + k][w2]); }
            if (a < 0 || b > 65535) far = 1;                // a leaf outside the 16-bit window around the first point's
            s_geo[k] = a; s_geo[3 + k] = sv_bits(b - a + 1);
            vb += s_geo[3 + k];
        }
        s_geo[6] = vb;
        s_bad = (far || vb > 32 || nruns > rc) ? 1 : 0;     // (a key wider than 32 bits: sv_body reports it, S2B_ERR_VOXEL)
    }
    __syncthreads();
    if (s_bad) { if (tid == 0) list[1 + atomicAdd(list, 1)] = sid; return; }
    const int mb0 = s_geo[0], mb1 = s_geo[1], mb2 = s_geo[2], bx = s_geo[3], by = s_geo[4], vbits = s_geo[6];
    {   // leaf coordinates packed z | y | x (the order of pcl's leaf index), in place of the offsets: every thread's runs through registers
        unsigned int xy[SR_RPT], zz[SR_RPT];
#pragma unroll
        for (int k = 0; k < SR_RPT; k++) { const int r = min(tid + SR_T * k, nruns - 1); xy[k] = (unsigned int)s_a[r] | ((unsigned int)s_b[r] << 16); zz[k] = s_z16[r]; }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < SR_RPT; k++) {
            const int r = tid + SR_T * k;
            if (r < nruns) {
                const unsigned int a = (xy[k] & 0xffffu) - (unsigned int)mb0, b = (xy[k] >> 16) - (unsigned int)mb1, c = zz[k] - (unsigned int)mb2;
                s_rkey[r] = (bx + by < 32 ? (c << (bx + by)) : 0u) | (b << bx) | a;
                s_a[r] = (unsigned short)r;
            }
        }
        if (tid == 0) s_rstart[nruns] = (unsigned short)n;
    }
    __syncthreads();
    S2M_STAMP(kid, 2, true);
    // ---- C. stable LSD radix sort of the run ids by key, 8 bits per pass (sv_body's: every wave owns a contiguous segment and ranks a 64-lane strip by ballot matching).
    // What the counting loop learns about a run — its id and digit (two dependent LDS reads), its rank among the strip's runs with the same digit and their number (eight
    // ballots) — stays in registers for the scatter loop of the pass (two per strip, at most SR_SPW strips per wave): the phase is bound by instructions issued.
    unsigned short *src = s_a, *dst = s_b;
    {
        const int seg = (nruns + SR_NW - 1) / SR_NW, lo = wave * seg, hi = min(nruns, lo + seg);
        unsigned int *cnt32 = reinterpret_cast<unsigned int *>(s_cnt);
        for (int shift = 0; shift < vbits; shift += 8) {
            for (int t = tid; t < 128 * SR_NW; t += SR_T) cnt32[t] = 0;
            __syncthreads();
            unsigned int pk[SR_SPW], rk[SR_SPW];                 // id | digit << 16 | valid << 24;  rank | same-digit count << 8
            const int nbit = min(8, vbits - shift);
#pragma unroll
            for (int st = 0; st < SR_SPW; st++) {
                pk[st] = 0; rk[st] = 0;
                if (lo + 64 * st < hi) {
                    const int j = lo + 64 * st + lane;
                    const bool valid = j < hi;
                    const unsigned int idx = valid ? src[j] : 0u;
                    const int d = valid ? (int)((s_rkey[idx] >> shift) & 255u) : 0;
                    unsigned long long mask = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; b++) if (b < nbit) { const unsigned long long bal = __ballot((d >> b) & 1); mask &= ((d >> b) & 1) ? bal : ~bal; }     // (the last pass of a 19-bit key has three bits)
                    const int rank = __popcll(mask & ((1ULL << lane) - 1ULL)), same = __popcll(mask);
                    if (valid && rank == 0) s_cnt[d * SR_NW + wave] += (unsigned short)same;
                    pk[st] = idx | ((unsigned int)d << 16) | (valid ? 1u << 24 : 0u); rk[st] = (unsigned int)rank | ((unsigned int)same << 8);
                }
            }
            __syncthreads();
            {   // exclusive scan in (digit, wave) order: thread t owns entries 4 t .. 4 t + 3 of [digit][wave]
                int c4[4], loc = 0;
#pragma unroll
                for (int u = 0; u < 4; u++) { c4[u] = s_cnt[4 * tid + u]; loc += c4[u]; }
                int tot;
                int run = block_excl_scan_nw<SR_NW>(loc, s_sc, tot);
#pragma unroll
                for (int u = 0; u < 4; u++) { s_cnt[4 * tid + u] = (unsigned short)run; run += c4[u]; }
            }
            __syncthreads();
#pragma unroll
            for (int st = 0; st < SR_SPW; st++) {                // a wave walks its own segment in order: its counter column needs no synchronisation
                if (lo + 64 * st < hi && (pk[st] >> 24)) {
                    const int d = (pk[st] >> 16) & 255, rank = rk[st] & 255;
                    const int off = s_cnt[d * SR_NW + wave];
                    dst[off + rank] = (unsigned short)(pk[st] & 0xffffu);
                    if (rank == 0) s_cnt[d * SR_NW + wave] = (unsigned short)(off + (rk[st] >> 8));
                }
            }
            __syncthreads();
            unsigned short *t = src; src = dst; dst = t;
        }
    }
    S2M_STAMP(kid, 3, true);
    // ---- D. leaves: the sorted position of every leaf's first run (ranks from a running count of the key changes) ...
    carry = 0;
    for (int t0 = 0; t0 < nruns; t0 += SR_T) {
        const int j = t0 + tid, jc = min(j, nruns - 1);
        const unsigned int k = s_rkey[src[jc]], kp = s_rkey[src[max(jc - 1, 0)]];
        const int head = (j < nruns && (j == 0 || kp != k)) ? 1 : 0;
        const unsigned long long hb = __ballot(head);
        if (lane == 0) s_w2[par][0][wave] = __popcll(hb);
        __syncthreads();
        int off = 0, tot = 0;
#pragma unroll
        for (int k2 = 0; k2 < SR_NW; k2++) { const int x = s_w2[par][0][k2]; if (k2 < wave) off += x; tot += x; }
        if (head) dst[carry + off + __popcll(hb & ((1ULL << lane) - 1ULL))] = (unsigned short)j;
        carry += tot; par ^= 1;
    }
    const int nleaf = carry;
    __syncthreads();
    S2M_STAMP(kid, 4, true);
    // ... then a lane per leaf, no barrier: its runs in sorted (= scan) order, each run's points in index order, up to eight points per step — the rest of the lane's current
    // run and, behind it, the start of the next (a batch per run: a 100-point leaf of 4-point runs took 20 steps). The loop is the same straight line for every lane (a lane
    // without work loads point 0 again): a batch is cut by the lane's cursor and requested one step before its points are added, and the compiler can count the loads in
    // flight (divergent per-leaf loops ended in s_waitcnt vmcnt(0) at every turn: a memory round trip per batch). A lane that completes a leaf DRAWS the next one from a
    // counter in LDS: the phase is bound by the instructions a wave issues, and a wave runs as long as its busiest lane — with leaves dealt out in advance (lane = leaf mod
    // 512) that was 25 steps against a mean of 10.
    {
        __shared__ int s_next;
        if (tid == 0) s_next = SR_T;
        __syncthreads();
        int L = tid, jj = 0, j1 = 0, k = 0, e0 = 0, cnt = 0;
        bool active = L < nleaf;
        if (active) { jj = dst[L]; j1 = (L + 1 < nleaf) ? dst[L + 1] : nruns; }
        struct Batch { float4 v[8]; int m, leaf, cnt; };      // m points; leaf >= 0: the batch completes that leaf (cnt points in all)
        auto cut = [&](Batch &bt) {
            int na = active ? e0 - k : 0;
            if (active && na == 0 && jj < j1) { const int r = src[jj]; k = s_rstart[r]; e0 = s_rstart[r + 1]; na = e0 - k; cnt += na; jj++; }
            na = min(na, 8);
            const int ka = k;
            k += na;
            int nb = 0, kb = 0;
            if (active && na < 8 && jj < j1) { const int r = src[jj]; k = s_rstart[r]; e0 = s_rstart[r + 1]; cnt += e0 - k; jj++; nb = min(8 - na, e0 - k); kb = k - na; k += nb; }
            bt.m = na + nb;
#pragma unroll
            for (int u = 0; u < 8; u++) bt.v[u] = u < bt.m ? p[u < na ? ka + u : kb + u] : sr_neutral;
            bt.leaf = -1; bt.cnt = cnt;
            if (bt.m > 0 && k >= e0 && jj >= j1) {             // the leaf is complete: the lane draws its next one
                bt.leaf = L;
                L = atomicAdd(&s_next, 1); active = L < nleaf; cnt = 0; k = 0; e0 = 0;
                if (active) { jj = dst[L]; j1 = (L + 1 < nleaf) ? dst[L + 1] : nruns; }
            }
        };
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        Batch b0, b1;
        cut(b0);
        while (__any(b0.m > 0)) {
            cut(b1);
#pragma unroll
            for (int u = 0; u < 8; u++) { acc.x = __fadd_rn(acc.x, b0.v[u].x); acc.y = __fadd_rn(acc.y, b0.v[u].y); acc.z = __fadd_rn(acc.z, b0.v[u].z); acc.w = __fadd_rn(acc.w, b0.v[u].w); }      // (an empty slot holds -0: x + -0 = x, for every x)
            if (b0.leaf >= 0) {
                const float nn = (float)b0.cnt;
                o[b0.leaf] = make_float4(acc.x / nn, acc.y / nn, acc.z / nn, acc.w / nn);
                acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
            b0 = b1;
        }
    }
    S2M_STAMP(kid, 5, true);
#ifdef VILF_STAMPS
    if (blockIdx.x == S2M_STAMP_WG && tid == 0) { s2m_dbg[kid * 32 + 30] = n; s2m_dbg[kid * 32 + 31] = vbits; s2m_dbg[kid * 32 + 29] = nruns; s2m_dbg[kid * 32 + 28] = nleaf; }
#endif
    if (tid == 0) out.n[sid] = nleaf;
}

// ---- fused steady-state map update: crop box + voxel grid of (leaf-ordered old map ++ appended scan) in ONE pass -------------
// createSubMap (:298-352) on a map that is already a voxel grid: the old map is in ascending leaf order (one point per leaf, up to
// rounding), only the few thousand appended points are not. One workgroup per stream:
//   A. the appended tail: crop, ABSOLUTE leaf key (z | y | x leaf coordinates, AXB bits each — lexicographic = pcl::VoxelGrid's index order
//      for any min corner), packed with the tail index into one 64-bit word, bitonic-sorted in LDS; sorted points -> scratch; the
//      index bits are then replaced by the number of distinct tail leaves before that position.
//   B. one streaming sweep over the old map (MU_E points per thread and tile, neighbours i - 1 / i + 1 from cache): crop test, leaf key,
//      binary search of the key in the LDS tail. An old leaf goes to (#old leaves before) + (#tail leaves before) - (#old leaves before
//      that also occur in the tail) — the first and third counts come from one packed block scan. A tail run with the same leaf is summed
//      into the old point (old first, then the tail in index order = the stable order of a full sort); tail leaves that fall strictly
//      between two old keys are emitted by the thread of the upper key, the ones beyond the last old key afterwards.
// HBM traffic: the map is read once and written once (the unfused path: crop copy, keys, merge, gather + write: ~3.5 x that).
// Streams whose tail does not fit the LDS buffer are handled by the BIG instantiation (tail in global memory), launched only when the
// scan capacity allows such tails; each instantiation skips the other's streams.
#ifndef MU_T
#define MU_T 512             // two workgroups of eight waves per CU instead of one of sixteen: the sweep waits 40 % of its cycles at its three barriers per tile and another stream's
                             // workgroup fills part of them. Same-box A/B of the voxel-grid group per 4096-frame step (tools/dev_ab_mut.sh): 1024 threads 9.01 ms, 512: 8.82, 256: 8.94
#endif
#ifndef MU_E
#define MU_E 2
#endif
#define MU_TILE (MU_T * MU_E)
#define MU_LDS_TAIL 8192
#define S2B_ERR_ORDER 8
// The local maps are kept in CELL-MAJOR leaf order: the absolute leaf coordinates (ix, iy, iz), AXB bits each, are packed as
//     cy | cx | iz | iy_low | ix_low        with cx = ix >> cs, cy = iy >> cs (a cell = 2^cs x 2^cs leaf columns, all z), *_low = the cs low bits
// — a permutation of the bits of pcl::VoxelGrid's index order z | y | x. Points of one leaf share a key under both orders, so the voxel grid (merge runs of equal
// keys, old points first, then the new ones in scan order) is the same computation; what the permutation buys is that every cell's points are contiguous, and
// the cells of a row (cy) follow each other in x: the map IS its own neighbour index, a directory of cell starts (b_dir_build) is all the 5-NN needs, and the
// per-frame re-sort of the whole map into a bucket-sorted copy is gone. PCL's order is what the outside sees: vilf_scan2map_get_map sorts the copy it hands out
// (s2b_get_map), and the 5-NN breaks exact distance ties by the PCL key (b_associate_ties).
template <int AXB>
__device__ __forceinline__ bool mu_leaf(const float4 q, float inv, int cs, unsigned long long &k) {
    const float fx = floorf(__fmul_rn(q.x, inv)), fy = floorf(__fmul_rn(q.y, inv)), fz = floorf(__fmul_rn(q.z, inv));
    const float lim = (float)(1 << (AXB - 1));
    const bool ok = fx >= -lim && fx < lim - 1.0f && fy >= -lim && fy < lim - 1.0f && fz >= -lim && fz < lim - 1.0f;   // false for NaN
    const int off = 1 << (AXB - 1);
    const unsigned long long ix = (unsigned long long)((int)fx + off), iy = (unsigned long long)((int)fy + off), iz = (unsigned long long)((int)fz + off);
    const unsigned long long lm = (1ULL << cs) - 1ULL;
    k = ok ? (((iy >> cs) << (2 * AXB + cs)) | ((ix >> cs) << (AXB + 2 * cs)) | (iz << (2 * cs)) | ((iy & lm) << cs) | (ix & lm)) : 0ULL;
    return ok;
}
// ---- the neighbour directory's constants and slot arithmetic (the design: see "radix-hashed voxel neighbour index" below)
#define S2B_DB 9
#define S2B_NB (1 << (2 * S2B_DB))
#define S2B_NBS (S2B_NB + 4)
#define S2B_DMARGIN 8
#define S2B_LOFF 65536
#define S2B_QR 1.001f        // search radius in metres: the reference's gate is 1 (squared distance < 1); the margin covers the rounding of q -+ 1 and of the squared distance
__device__ __forceinline__ int cm_leaf(float v, float inv) { return (int)floorf(__fmul_rn(v, inv)) + S2B_LOFF; }
__device__ __forceinline__ int dir_slot(int cy, int cx) { return ((cy & 511) << S2B_DB) | (cx & 511); }
// One point of the cell-major array: (cy, cx) its cell's row / column modulo 512, (cyp, cxp) its predecessor's. Differences are taken modulo 512 too: a map spans fewer cells.
__device__ __forceinline__ void dir_point(unsigned *T, unsigned tag, int i, int n, int cy, int cx, int cyp, int cxp) {
    if (i == n - 1) for (int c = cx + 1; c <= cx + S2B_DMARGIN; c++) T[dir_slot(cy, c)] = tag | (unsigned)n;
    const bool first = i == 0, newrow = first || cy != cyp;
    if (!newrow && cx == cxp) return;
    const unsigned v = tag | (unsigned)i;
    // a long run of empty cells inside a row is treated like a row break: S2B_DMARGIN cells behind the previous occupied cell and S2B_DMARGIN in front of this one are
    // filled, the slots in between keep their old tag (a span can only end there if it holds no occupied cell — it then reads as empty, which it is)
    const int gapfull = ((cx - cxp) & 511) - 1;
    const bool brk = newrow || gapfull > 2 * S2B_DMARGIN;
    const int gap = brk ? S2B_DMARGIN : gapfull;                              // cells to fill before this one
    for (int c = cx - gap; c <= cx; c++) T[dir_slot(cy, c)] = v;
    if (brk && !first) for (int c = cxp + 1; c <= cxp + S2B_DMARGIN; c++) T[dir_slot(cyp, c)] = v;
}
// the directory slot (dir_slot: nine low bits of the cell's row and column) of a cell-major key. b_map_update writes the directory of the map it emits — the index of
// the NEXT step — itself: an emitted point whose predecessor in the output is known in the sweep (the usual case: the previous old point survives the crop and no new
// leaf falls between the two) compares the two cells on the spot and, when they differ, fills the slots exactly as b_dir_build would (dir_point); every other emitted
// point (first of the map, behind a cropped point, new leaves, merged leaves) goes on a short list that is resolved from the written map after the sweep.
template <int AXB>
__device__ __forceinline__ unsigned mu_cid(unsigned long long k, int cs) {
    const int adj = (65536 - (1 << (AXB - 1))) >> cs;              // the directory counts cells from leaf offset 65536 (cm_leaf), the key from 2^(AXB-1)
    const int cy = (int)(k >> (2 * AXB + cs)) + adj, cx = (int)((k >> (AXB + 2 * cs)) & ((1ULL << (AXB - cs)) - 1ULL)) + adj;
    return (unsigned)(((cy & 511) << 9) | (cx & 511));
}
struct MuBox { float mnx, mny, mnz, mxx, mxy, mxz; };
__device__ __forceinline__ bool mu_inside(const float4 q, const MuBox &b) { return !(q.x < b.mnx || q.y < b.mny || q.z < b.mnz || q.x > b.mxx || q.y > b.mxy || q.z > b.mxz); }
__device__ __forceinline__ void mu_acc(float4 &s, const float4 q) { s.x = __fadd_rn(s.x, q.x); s.y = __fadd_rn(s.y, q.y); s.z = __fadd_rn(s.z, q.z); s.w = __fadd_rn(s.w, q.w); }
__device__ __forceinline__ float4 mu_centroid(const float4 *ts, int a, int e) {
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int j = a; j < e; j++) mu_acc(s, ts[j]);
    const float nn = (float)(e - a);
    return make_float4(s.x / nn, s.y / nn, s.z / nn, s.w / nn);
}
// the uncommon old point of the sweep (old index i; H / M = old leaves / matched old leaves before it): tail leaves to emit between the
// previous old key and this one, a tail run in this leaf, or a leaf that holds several old points. Such points are queued during the
// sweep and handled afterwards, one per lane, from global memory — their dependent loads never sit on the sweep's critical path.
template <int AXB, int IDXB>
__device__ __forceinline__ void mu_rare(int i, int H, int M, bool head, bool between, int jlo, int nOld, const float4 *p, float4 *o, int *fixq, int *s_fn, const float4 *ts, const unsigned long long *T,
                                        int ntv, int THtot, float inv, int cs, const MuBox &box) {
    constexpr unsigned long long LOW = (1ULL << IDXB) - 1;
    auto lower = [&](unsigned long long k) { int lo = 0, hi = ntv; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((T[mid] >> IDXB) < k) lo = mid + 1; else hi = mid; } return lo; };
    auto thp = [&](int j) { return j < ntv ? (int)(T[j] & LOW) : THtot; };
    const float4 q = p[i], qp = p[max(i - 1, 0)];
    unsigned long long key, kp;
    mu_leaf<AXB>(q, inv, cs, key); mu_leaf<AXB>(qp, inv, cs, kp);
    const bool sv = mu_inside(q, box), hasp = i > 0, first = !hasp || kp != key;
    int jup = jlo;                                           // jlo = lower(key): found by the sweep, handed over through the queue
    while (jup < ntv && (T[jup] >> IDXB) == key) jup++;
    const bool tm = jup > jlo;
    if (first) {
        if (between)                                         // (the sweep saw tail leaves between the two old keys: only then is the search worth its dependent LDS reads)
        for (int jj = hasp ? lower(kp + 1) : 0; jj < jlo;) {
            const unsigned long long lf = T[jj] >> IDXB;
            int f = jj + 1;
            while (f < jlo && (T[f] >> IDXB) == lf) f++;
            o[H + thp(jj) - M] = mu_centroid(ts, jj, f); fixq[atomicAdd(s_fn, 1)] = H + thp(jj) - M;
            jj = f;
        }
        if (tm && !sv) {                                     // the tail has this leaf; the old run may have no survivor at all -> a new leaf
            bool any = false;
            for (int b = i + 1; b < nOld; b++) { const float4 r = p[b]; unsigned long long kr; mu_leaf<AXB>(r, inv, cs, kr); if (kr != key) break; if (mu_inside(r, box)) { any = true; break; } }
            if (!any) { o[H + thp(jlo) - M] = mu_centroid(ts, jlo, jup); fixq[atomicAdd(s_fn, 1)] = H + thp(jlo) - M; }
        }
    }
    if (head) {
        float4 s = make_float4(__fadd_rn(0.0f, q.x), __fadd_rn(0.0f, q.y), __fadd_rn(0.0f, q.z), __fadd_rn(0.0f, q.w));
        int cnt = 1;
        for (int b = i + 1; b < nOld; b++) { const float4 r = p[b]; unsigned long long kr; mu_leaf<AXB>(r, inv, cs, kr); if (kr != key) break; if (mu_inside(r, box)) { mu_acc(s, r); cnt++; } }
        for (int jj = jlo; jj < jup; jj++) { mu_acc(s, ts[jj]); cnt++; }
        const float nn = (float)cnt;
        o[H + thp(jlo) - M] = make_float4(s.x / nn, s.y / nn, s.z / nn, s.w / nn); fixq[atomicAdd(s_fn, 1)] = H + thp(jlo) - M;
    }
}
#ifdef MU_WPE
#define MU_LB __launch_bounds__(MU_T, MU_WPE)
#else
#define MU_LB __launch_bounds__(MU_T)
#endif
template <bool BIG>
__global__ MU_LB void b_map_update(CSet map, const int *n_old, const double *pose_all, double half, float inv, int cs, CSet out, unsigned *dir_all, unsigned dtag, int *fix_all, float4 *ts_all, int ts_stride,
                                                       unsigned long long *gT_all, int gT_stride, int lds_lo, int lds_cap, int *gq_all, size_t gq_stride, int *err) {
    constexpr int AXB = BIG ? 16 : 17, IDXB = BIG ? 16 : 13;
    constexpr unsigned long long LOW = (1ULL << IDXB) - 1;
    extern __shared__ unsigned long long s_T[];
    __shared__ int s_w[MU_E][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sid = blockIdx.x;
    const int n = map.n[sid], nOld = min(max(n_old[sid], 0), n), nt = n - nOld;
    if (BIG ? nt <= lds_cap : (nt <= lds_lo || nt > lds_cap)) return;          // each launch takes the streams of its tail-size class
    unsigned long long *T = BIG ? gT_all + (size_t)sid * gT_stride : s_T;
    const float4 *p = map.p + (size_t)sid * map.cap;
    float4 *o = out.p + (size_t)sid * out.cap;
    unsigned *Tdir = dir_all + (size_t)sid * S2B_NBS;          // the directory of the map this launch emits
    int *fixq = fix_all + (size_t)sid * out.cap;               // positions of emitted points whose predecessor the sweep does not know
    float4 *ts = ts_all + (size_t)sid * ts_stride;
    const double *pose = pose_all + 24 * sid;
    MuBox box;
    box.mnx = (float)(pose[4] - half); box.mny = (float)(pose[5] - half); box.mnz = (float)(pose[6] - half);
    box.mxx = (float)(pose[4] + half); box.mxy = (float)(pose[5] + half); box.mxz = (float)(pose[6] + half);
    int bad = 0;
    [[maybe_unused]] const int skid = nOld > 45000 ? 4 : 5;
    S2M_STAMP(skid, 0, true);

    // ---- A. tail: crop, key, sort, points in sorted order, distinct-leaf prefix
    int P2 = 2;
    while (P2 < nt) P2 <<= 1;
    int myvalid = 0;
    // Round 5: tails of up to MU_RADIX points (every stream of the steady state) are sorted by a stable LSD radix sort on a COMPACT key instead of the 78-step bitonic
    // network on 64-bit words (22 % of the kernel): the key's fields cy | cx | iz | lows are taken relative to their minima over the tail and packed into the bits
    // their ranges need (a scan inside the crop box: 7 + 7 + 5 + 2 cs bits), one more bit on top marks the points the crop or the grid rejects, so they sort behind
    // the rest. Three 8-bit passes of sv_runs' ballot-matching sort (b_scan_voxel_runs, C) over 16-bit indices; equal keys keep their index order = the order of
    // the (key << IDXB | index) words the network sorted. T[] is composed from the sorted indices and the fields at the end: the same array, to the bit.
    constexpr int NW = MU_T / 64, MU_RADIX = 4096, RK = MU_RADIX / MU_T;
    __shared__ int s_mm[6][NW], s_geo[8];
    bool radix = !BIG && nt <= MU_RADIX && nt <= lds_cap;
    if (radix) {
        unsigned long long kf[RK];
        unsigned vm = 0;
        const int fb = AXB - cs;                                                       // bits of a cell coordinate
        int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-1, -1, -1};
#pragma unroll
        for (int k = 0; k < RK; k++) {
            const int j = tid + MU_T * k;
            kf[k] = 0;
            if (j < nt) {
                const float4 q = p[nOld + j];
                unsigned long long kk;
                if (mu_inside(q, box)) {
                    if (mu_leaf<AXB>(q, inv, cs, kk)) {
                        kf[k] = kk; vm |= 1u << k; myvalid++;
                        const int f0 = (int)(kk >> (2 * AXB + cs)), f1 = (int)((kk >> (AXB + 2 * cs)) & ((1ULL << fb) - 1ULL)), f2 = (int)((kk >> (2 * cs)) & ((1ULL << AXB) - 1ULL));
                        mn[0] = min(mn[0], f0); mx[0] = max(mx[0], f0); mn[1] = min(mn[1], f1); mx[1] = max(mx[1], f1); mn[2] = min(mn[2], f2); mx[2] = max(mx[2], f2);
                    } else bad |= S2B_ERR_VOXEL;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) {
#pragma unroll
            for (int of = 32; of > 0; of >>= 1) { mn[k] = min(mn[k], __shfl_xor(mn[k], of, 64)); mx[k] = max(mx[k], __shfl_xor(mx[k], of, 64)); }
            if (lane == 0) { s_mm[k][wave] = mn[k]; s_mm[3 + k][wave] = mx[k]; }
        }
        __syncthreads();
        if (tid == 0) {
            int vb = 2 * cs;
            for (int k = 0; k < 3; k++) {
                int a = s_mm[k][0], b = s_mm[3 + k][0];
                for (int w2 = 1; w2 < NW; w2++) { a = min(a, s_mm[k][w2]); b = max(b, s_mm[3 + k][w2]); }
                if (b < a) { a = 0; b = 0; }                                           // no valid tail point
                int nb = 0;
                while (((b - a) >> nb) != 0) nb++;
                s_geo[k] = a; s_geo[3 + k] = nb; vb += nb;
            }
            s_geo[6] = vb;
        }
        __syncthreads();
        const int vbits = s_geo[6];
        radix = vbits <= 31;                                                           // (never more inside a crop box; the network below takes such a tail)
        if (radix) {
            unsigned int *s_rkey = reinterpret_cast<unsigned int *>(s_T);
            unsigned short *src = reinterpret_cast<unsigned short *>(s_T) + 2 * lds_cap, *dst = src + lds_cap;
            unsigned short *s_cnt = reinterpret_cast<unsigned short *>(s_T + lds_cap);           // [256 digits][NW waves]: the sweep's key tile, not in use yet
            const int m0 = s_geo[0], m1 = s_geo[1], m2 = s_geo[2], b1 = s_geo[4], b2 = s_geo[5];
#pragma unroll
            for (int k = 0; k < RK; k++) {
                const int j = tid + MU_T * k;
                if (j < nt) {
                    const unsigned long long kk = kf[k];
                    const unsigned f0 = (unsigned)((int)(kk >> (2 * AXB + cs)) - m0), f1 = (unsigned)((int)((kk >> (AXB + 2 * cs)) & ((1ULL << fb) - 1ULL)) - m1),
                                   f2 = (unsigned)((int)((kk >> (2 * cs)) & ((1ULL << AXB) - 1ULL)) - m2), lw = (unsigned)(kk & ((1ULL << (2 * cs)) - 1ULL));
                    s_rkey[j] = ((vm >> k) & 1u) ? ((((f0 << b1) | f1) << b2 | f2) << (2 * cs)) | lw : (1u << vbits);
                    src[j] = (unsigned short)j;
                }
            }
            const int seg = (nt + NW - 1) / NW, lo = wave * seg, hi = min(nt, lo + seg);
            unsigned int *cnt32 = reinterpret_cast<unsigned int *>(s_cnt);
            for (int shift = 0; shift < vbits + 1; shift += 8) {
                for (int t = tid; t < 128 * NW; t += MU_T) cnt32[t] = 0;
                __syncthreads();
                unsigned int pk[RK], rk[RK];                     // index | digit << 16 | valid << 24;  rank | same-digit count << 8
                const int nbit = min(8, vbits + 1 - shift);
#pragma unroll
                for (int st = 0; st < RK; st++) {
                    pk[st] = 0; rk[st] = 0;
                    if (lo + 64 * st < hi) {
                        const int j = lo + 64 * st + lane;
                        const bool valid = j < hi;
                        const unsigned int idx = valid ? src[j] : 0u;
                        const int d = valid ? (int)((s_rkey[idx] >> shift) & 255u) : 0;
                        unsigned long long mask = __ballot(valid);
#pragma unroll
                        for (int b = 0; b < 8; b++) if (b < nbit) { const unsigned long long bal = __ballot((d >> b) & 1); mask &= ((d >> b) & 1) ? bal : ~bal; }
                        const int rank = __popcll(mask & ((1ULL << lane) - 1ULL)), same = __popcll(mask);
                        if (valid && rank == 0) s_cnt[d * NW + wave] += (unsigned short)same;
                        pk[st] = idx | ((unsigned int)d << 16) | (valid ? 1u << 24 : 0u); rk[st] = (unsigned int)rank | ((unsigned int)same << 8);
                    }
                }
                __syncthreads();
                {   // exclusive scan in (digit, wave) order: thread t owns entries 4 t .. 4 t + 3 of [digit][wave] (256 NW / MU_T = 4)
                    int c4[4], loc = 0;
#pragma unroll
                    for (int u = 0; u < 4; u++) { c4[u] = s_cnt[4 * tid + u]; loc += c4[u]; }
                    int tot;
                    int run = block_excl_scan_nw<NW>(loc, s_w[0], tot);
#pragma unroll
                    for (int u = 0; u < 4; u++) { s_cnt[4 * tid + u] = (unsigned short)run; run += c4[u]; }
                }
                __syncthreads();
#pragma unroll
                for (int st = 0; st < RK; st++) {                // a wave walks its own segment in order: its counter column needs no synchronisation
                    if (lo + 64 * st < hi && (pk[st] >> 24)) {
                        const int d = (pk[st] >> 16) & 255, rank = rk[st] & 255;
                        const int off = s_cnt[d * NW + wave];
                        dst[off + rank] = (unsigned short)(pk[st] & 0xffffu);
                        if (rank == 0) s_cnt[d * NW + wave] = (unsigned short)(off + (rk[st] >> 8));
                    }
                }
                __syncthreads();
                unsigned short *t = src; src = dst; dst = t;
            }
            // T[j] = (full key << IDXB) | index in sorted order; the rejected points (behind the valid ones) and the padding up to P2: ~0
            int ntv0;
            { int tot; block_excl_scan_nw<NW>(myvalid, s_w[0], tot); ntv0 = tot; }
            unsigned int ri[RK], rv[RK];
#pragma unroll
            for (int k = 0; k < RK; k++) { const int j = tid + MU_T * k; ri[k] = j < ntv0 ? src[j] : 0u; rv[k] = j < ntv0 ? s_rkey[ri[k]] : 0u; }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RK; k++) {
                const int j = tid + MU_T * k;
                if (j < P2) {
                    unsigned long long w = ~0ULL;
                    if (j < ntv0) {
                        const unsigned long long c = rv[k];
                        const unsigned long long lw = c & ((1ULL << (2 * cs)) - 1ULL), f2 = ((c >> (2 * cs)) & ((1ULL << b2) - 1ULL)) + (unsigned long long)m2,
                                                 f1 = ((c >> (2 * cs + b2)) & ((1ULL << b1) - 1ULL)) + (unsigned long long)m1, f0 = (c >> (2 * cs + b2 + b1)) + (unsigned long long)m0;
                        const unsigned long long kk = (f0 << (2 * AXB + cs)) | (f1 << (AXB + 2 * cs)) | (f2 << (2 * cs)) | lw;
                        w = (kk << IDXB) | (unsigned long long)ri[k];
                    }
                    T[j] = w;
                }
            }
            __syncthreads();
        } else myvalid = 0;
    }
    if (!radix) {
    for (int j = tid; j < P2; j += MU_T) {
        unsigned long long w = ~0ULL;
        if (j < nt) {
            const float4 q = p[nOld + j];
            unsigned long long k;
            if (mu_inside(q, box)) { if (mu_leaf<AXB>(q, inv, cs, k)) { w = (k << IDXB) | (unsigned long long)j; myvalid++; } else bad |= S2B_ERR_VOXEL; }
        }
        T[j] = w;
    }
    __syncthreads();
    }
    if (radix) { }
    else if (!BIG && P2 >= 2 * MU_T) {
        // bitonic network with every wave owning a contiguous block of ppw pairs = 2 ppw elements: a step whose partner distance j is <= ppw stays inside the
        // blocks, so it needs no workgroup barrier (LDS accesses of one wave are ordered) — 68 of the 78 steps of a 4096-entry sort
        const int ppw = (P2 >> 1) / (MU_T / 64);
        bool was_local = true;
        for (int k = 2; k <= P2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool local = j <= ppw;
                if (!local && was_local) __syncthreads();
                for (int q = lane; q < ppw; q += 64) {
                    const int t = wave * ppw + q;
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                    const unsigned long long a = T[i], b = T[l];
                    if ((a > b) == ((i & k) == 0)) { T[i] = b; T[l] = a; }
                }
                if (local) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } else __syncthreads();
                was_local = local;
            }
        __syncthreads();
    } else
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (P2 >> 1); t += MU_T) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i | j;
                const unsigned long long a = T[i], b = T[l];
                if ((a > b) == ((i & k) == 0)) { T[i] = b; T[l] = a; }
            }
            __syncthreads();
        }
    int ntv;
    { int tot; block_excl_scan_nw<MU_T / 64>(myvalid, s_w[0], tot); ntv = tot; }
    for (int j = tid; j < ntv; j += MU_T) ts[j] = p[nOld + (int)(T[j] & LOW)];
    __syncthreads();
    int THtot = 0;
    {   // thp(j) = number of distinct leaves among tail positions < j (blocked chunks of C positions per thread)
        const int C = (P2 + MU_T - 1) / MU_T, j0 = tid * C;
        int cnt = 0;
        for (int j = j0; j < j0 + C && j < ntv; j++) cnt += (j == 0 || (T[j - 1] >> IDXB) != (T[j] >> IDXB)) ? 1 : 0;
        int tot;
        int run = block_excl_scan_nw<MU_T / 64>(cnt, s_w[1], tot);
        THtot = tot;
        // flags are re-derived from the leaf bits, which the rewrite of the low bits does not touch
        unsigned long long prev = (j0 > 0 && j0 <= ntv) ? (T[j0 - 1] >> IDXB) : ~0ULL;
        __syncthreads();
        for (int j = j0; j < j0 + C && j < ntv; j++) {
            const unsigned long long lf = T[j] >> IDXB;
            const int hd = (j == 0 || prev != lf) ? 1 : 0;
            T[j] = (lf << IDXB) | (unsigned long long)run;
            run += hd; prev = lf;
        }
    }
    __syncthreads();
    auto lower = [&](unsigned long long k) { int lo = 0, hi = ntv; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((T[mid] >> IDXB) < k) lo = mid + 1; else hi = mid; } return lo; };
    auto thp = [&](int j) { return j < ntv ? (int)(T[j] & LOW) : THtot; };

    S2M_STAMP(skid, 1, true);
    // ---- B. sweep over the old map. Per tile: keys (+ survivor bit 63) go through LDS, so the neighbours i - 1 / i + 1 cost no global
    // load; the next tile's points are requested before this tile is processed. The common case — an old leaf that survives, alone in
    // its leaf, no tail point in it, no tail leaf between it and its predecessor — is handled inline; everything else is flagged and
    // done by mu_rare() from LDS state (kept out of the unrolled body: registers).
    unsigned long long *s_key = s_T + (BIG ? 0 : lds_cap);
    int *s_te = reinterpret_cast<int *>(s_key + MU_TILE);
    int *gq = gq_all + (size_t)sid * gq_stride;      // queue of uncommon points in global memory (4 ints each: index | head bit 30 | between bit 29, H, M, lower bound in the tail): room for every old point
    __shared__ int s_qn, s_fn, s_jlo;
    __shared__ unsigned long long s_nk;
    if (tid == 0) { s_qn = 0; s_fn = 0; }
    constexpr unsigned long long SVB = 1ULL << 63;
    int carryH = 0, carryM = 0, carryTE = 0, carryJ = 0;
    unsigned long long carryK = 0;
    float4 qn[MU_E], qn2[MU_E];                                     // two tiles of points in flight: one tile ahead leaves the sweep waiting a memory round trip per tile
#pragma unroll
    for (int u = 0; u < MU_E; u++) { qn[u] = p[min(u * MU_T + tid, max(nOld - 1, 0))]; qn2[u] = p[min(MU_TILE + u * MU_T + tid, max(nOld - 1, 0))]; }
    S2M_ACC_DECL
    for (int t0 = 0; t0 < nOld; t0 += MU_TILE) {
        S2M_ACC(0);
        float4 q[MU_E];
        unsigned long long key[MU_E];
        int tb[MU_E], flag[MU_E], incl[MU_E];
        unsigned hm = 0, fm = 0, cm = 0;                             // per-element bit masks: head, first of its run, needs mu_rare
#pragma unroll
        for (int u = 0; u < MU_E; u++) { q[u] = qn[u]; qn[u] = qn2[u]; qn2[u] = p[min(t0 + 2 * MU_TILE + u * MU_T + tid, nOld - 1)]; }
#pragma unroll
        for (int u = 0; u < MU_E; u++) {
            const int i = t0 + u * MU_T + tid;
            const bool valid = i < nOld;
            if (!mu_leaf<AXB>(q[u], inv, cs, key[u]) && valid) bad |= S2B_ERR_VOXEL;
            const bool sv = valid && mu_inside(q[u], box);
            s_key[u * MU_T + tid] = valid ? (key[u] | (sv ? SVB : 0ULL)) : ~0ULL;
            if (sv) hm |= 1u << u;
        }
        if (tid == 0) { unsigned long long kx; mu_leaf<AXB>(qn[0], inv, cs, kx); s_nk = kx; }
        S2M_ACC(1);
        lds_barrier();               // (the three barriers of the tile loop order LDS only — s_key / s_te / s_w; the points requested two tiles ahead and this tile's stores stay in flight across them: the map and the queue are read again only behind the full barrier that follows the loop)
        S2M_ACC(2);
        const unsigned long long lastK = s_key[MU_TILE - 1];
#pragma unroll
        for (int u = 0; u < MU_E; u++) {
            const int e = u * MU_T + tid, i = t0 + e;
            const bool valid = i < nOld, hasp = i > 0, hasn = i + 1 < nOld;
            const unsigned long long wp = e > 0 ? s_key[e - 1] : carryK;
            unsigned long long wn = e + 1 < MU_TILE ? s_key[e + 1] : 0ULL;
            if (e + 1 == MU_TILE && hasn) wn = s_nk;                  // the next tile's first key: from the point thread 0 already holds (a global load here made the tile's last wave wait out every prefetch in flight)
            const unsigned long long kp = wp & ~SVB, kn = wn & ~SVB;
            if (valid && hasp && kp > key[u]) bad |= S2B_ERR_ORDER;
            if (valid && (!hasp || kp != key[u])) fm |= 1u << u;
            if (valid && hasn && kn == key[u]) cm |= 1u << u;       // the leaf continues: summed by mu_rare
            if (((hm >> u) & 1) && hasp && kp == key[u]) {          // an earlier survivor of the same leaf owns it (rare: rounding put two centroids in one leaf)
                bool mine = !(wp & SVB);
                if (mine) for (int b = i - 2; b >= 0; b--) { const float4 r = p[b]; unsigned long long kr; mu_leaf<AXB>(r, inv, cs, kr); if (kr != key[u]) break; if (mu_inside(r, box)) { mine = false; break; } }
                if (!mine) hm &= ~(1u << u);
            }
        }
        S2M_ACC(3);
        int lo[MU_E];
        {   // lower bound of every old key in the sorted tail. The tail array is padded with ~0 to the power of two P2, so the search is the branch-free halving
            // form: one LDS read, one 64-bit compare and one conditional add per step and chain (the sweep is bound by VALU issue — sixteen waves share four
            // SIMDs — and the textbook lo / hi / mid form cost four times the instructions)
            // Round 5: the search runs over a WINDOW of the tail. Both sequences ascend, so every key of this tile has its lower bound at or behind the previous
            // tile's last one (carryJ), and at or before the first tail entry whose leaf is >= this tile's last key: the first of four power-of-two windows behind
            // carryJ that ends on such an entry is taken (four uniform LDS reads; a tile of 2048 old points meets ~130 tail entries: 8 steps instead of 12).
            // A full tile only (a partial one carries ~0 keys), the whole padded array otherwise. Same lower bounds, fewer steps.
            unsigned long long kk[MU_E];
            int W = P2, jb = 0;
            if (t0 + MU_TILE <= nOld && carryJ + 64 <= P2) {
                const unsigned long long kl = (lastK & ~SVB) << IDXB;
                const unsigned long long e0 = T[carryJ + 63], e1 = T[min(carryJ + 127, P2 - 1)], e2 = T[min(carryJ + 255, P2 - 1)], e3 = T[min(carryJ + 511, P2 - 1)];
                if (e0 >= kl) { W = 64; jb = carryJ; }
                else if (carryJ + 128 <= P2 && e1 >= kl) { W = 128; jb = carryJ; }
                else if (carryJ + 256 <= P2 && e2 >= kl) { W = 256; jb = carryJ; }
                else if (carryJ + 512 <= P2 && e3 >= kl) { W = 512; jb = carryJ; }
            }
#pragma unroll
            for (int u = 0; u < MU_E; u++) { lo[u] = jb; kk[u] = key[u] << IDXB; }
            for (int half = W >> 1; half > 0; half >>= 1) {
                unsigned long long tv[MU_E];
#pragma unroll
                for (int u = 0; u < MU_E; u++) tv[u] = T[lo[u] + half - 1];
#pragma unroll
                for (int u = 0; u < MU_E; u++) lo[u] += tv[u] < kk[u] ? half : 0;
            }
#pragma unroll
            for (int u = 0; u < MU_E; u++) lo[u] += T[lo[u]] < kk[u] ? 1 : 0;
#pragma unroll
            for (int u = 0; u < MU_E; u++) {
                const bool valid = t0 + u * MU_T + tid < nOld;
                const bool tm = valid && lo[u] < ntv && (T[lo[u]] >> IDXB) == key[u];
                if (tm) cm |= 1u << u;
                tb[u] = thp(lo[u]);                                  // tail leaves before this key
                s_te[u * MU_T + tid] = tb[u] + (tm ? 1 : 0);         // ... up to and including it
                if (u == MU_E - 1 && tid == MU_T - 1) s_jlo = lo[u];
                const bool head = (hm >> u) & 1;
                flag[u] = (head ? 1 : 0) | ((head && tm) ? (1 << 16) : 0);
            }
        }
#pragma unroll
        for (int u = 0; u < MU_E; u++) {           // inclusive wave scan of the two packed counts from two ballots
            const unsigned long long below = (2ULL << lane) - 1ULL;
            incl[u] = __popcll(__ballot(flag[u] & 1) & below) | (__popcll(__ballot(flag[u] >> 16) & below) << 16);
        }
        if (lane == 63) {
#pragma unroll
            for (int u = 0; u < MU_E; u++) s_w[u][wave] = incl[u];
        }
        S2M_ACC(4);
        lds_barrier();
        S2M_ACC(5);
        int base = 0;
#pragma unroll
        for (int u = 0; u < MU_E; u++) {
            int off = 0, tot = 0;
#pragma unroll
            for (int k = 0; k < MU_T / 64; k++) { const int x = s_w[u][k]; if (k < wave) off += x; tot += x; }
            const int ex = base + off + incl[u] - flag[u];
            base += tot;
            const int e = u * MU_T + tid;
            const int te_prev = e > 0 ? s_te[e - 1] : carryTE;
            const bool btw = ((fm >> u) & 1) && tb[u] > te_prev;     // tail leaves strictly between the previous old key and this one
            if (btw) cm |= 1u << u;
            const int H = carryH + (ex & 0xffff), M = carryM + (ex >> 16);
            if ((cm >> u) & 1) {
                const int k = atomicAdd(&s_qn, 1);
                reinterpret_cast<int4 *>(gq)[k] = make_int4((t0 + e) | (((hm >> u) & 1) << 30) | (btw ? (1 << 29) : 0), H, M, lo[u]);
            }
            else if ((hm >> u) & 1) {                                // the common case: the old point is its leaf's centroid, sum from +0 as the reference does
                const int pos = H + tb[u] - M;
                o[pos] = make_float4(__fadd_rn(0.0f, q[u].x), __fadd_rn(0.0f, q[u].y), __fadd_rn(0.0f, q[u].z), __fadd_rn(0.0f, q[u].w));
                // directory: the previous old point survives (its leaf is in the output) and no new leaf lies between -> that leaf's output is this point's predecessor
                const unsigned long long wp6 = e > 0 ? s_key[e - 1] : carryK;
                // (only the two cheap cases are settled here — same cell: nothing to write; the next cell of the same row: one store. Gaps and row starts mean loops
                // of stores, which a sixteen-wave workgroup between two barriers pays sixteen-fold: they go on the list with the rest.)
                const unsigned cme = mu_cid<AXB>(key[u], cs), cpr = mu_cid<AXB>(wp6 & ~SVB, cs);
                const bool known = t0 + e > 0 && (wp6 & SVB) && !btw;
                if (known && cme == cpr) { }
                else if (known && cme == cpr + 1u && (cme & 511u) != 0u) Tdir[cme] = dtag | (unsigned)pos;
                else fixq[atomicAdd(&s_fn, 1)] = pos;
            }
        }
        const int nextTE = s_te[MU_TILE - 1];
        carryH += base & 0xffff; carryM += base >> 16;
        carryTE = nextTE; carryK = lastK; carryJ = min(s_jlo, P2 - 1);
        S2M_ACC(6);
        lds_barrier();                                               // s_key / s_te / s_w are rewritten by the next tile
        S2M_ACC(7);
    }
    S2M_ACC_OUT(skid);
    S2M_STAMP(skid, 2, true);
    __syncthreads();
    {   // the queued points, one per lane: their dependent global loads run in parallel here instead of stalling a tile of the sweep
        const int qn = s_qn;
        for (int k = tid; k < qn; k += MU_T) {
            const int4 e = reinterpret_cast<const int4 *>(gq)[k];
            mu_rare<AXB, IDXB>(e.x & 0x1fffffff, e.y, e.z, (e.x >> 30) & 1, (e.x >> 29) & 1, e.w, nOld, p, o, fixq, &s_fn, ts, T, ntv, THtot, inv, cs, box);
        }
    }
    __syncthreads();
    S2M_STAMP(skid, 3, true);
    // ---- tail leaves beyond the last old key (all of them when there is no old map)
    int jlast = 0;
    if (nOld > 0) { unsigned long long kl; mu_leaf<AXB>(p[nOld - 1], inv, cs, kl); jlast = lower(kl + 1); }
    for (int jj = jlast + tid; jj < ntv; jj += MU_T) {
        const unsigned long long lf = T[jj] >> IDXB;
        if (jj > jlast && (T[jj - 1] >> IDXB) == lf) continue;
        int e = jj + 1;
        while (e < ntv && (T[e] >> IDXB) == lf) e++;
        o[carryH + thp(jj) - carryM] = mu_centroid(ts, jj, e); fixq[atomicAdd(&s_fn, 1)] = carryH + thp(jj) - carryM;
    }
    S2M_STAMP(skid, 4, true);
#ifdef VILF_STAMPS
    if (blockIdx.x == S2M_STAMP_WG && tid == 0) { s2m_dbg[skid * 32 + 30] = nOld; s2m_dbg[skid * 32 + 29] = nt; s2m_dbg[skid * 32 + 28] = s_qn; }
#endif
    const int n_out = carryH + THtot - carryM;
    if (tid == 0) out.n[sid] = n_out;
    if (bad) atomicOr(err + sid, bad);
    // ---- the directory entries the sweep could not write: predecessor and own cell from the emitted map (this workgroup's own stores: visible after the barrier)
    __threadfence_block();
    __syncthreads();
    for (int k = tid; k < s_fn; k += MU_T) {
        const int pos = fixq[k];
        const float4 a = o[pos], ap = o[max(pos - 1, 0)];
        dir_point(Tdir, dtag, pos, n_out, (cm_leaf(a.y, inv) >> cs) & 511, (cm_leaf(a.x, inv) >> cs) & 511, (cm_leaf(ap.y, inv) >> cs) & 511, (cm_leaf(ap.x, inv) >> cs) & 511);
    }
    if (tid == 0 && n_out > 0) {                      // the margin behind the last point (its own thread may have resolved it in the sweep, where the count was not known yet)
        const float4 a = o[n_out - 1];
        const int cy = (cm_leaf(a.y, inv) >> cs) & 511, cx = (cm_leaf(a.x, inv) >> cs) & 511;
        for (int c = cx + 1; c <= cx + S2B_DMARGIN; c++) Tdir[dir_slot(cy, c)] = dtag | (unsigned)n_out;
    }
}
// the old maps of every stream are in non-decreasing leaf order with leaf coordinates inside the AXB-bit range? flag[0] |= 1 if not
__global__ void b_check_order(CSet map, float inv, int cs, int axb, int *flag) {
    const int sid = blockIdx.y, n = map.n[sid];
    const float4 *p = map.p + (size_t)sid * map.cap;
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        unsigned long long k, kp = 0;
        const bool ok = axb == 16 ? mu_leaf<16>(p[i], inv, cs, k) : mu_leaf<17>(p[i], inv, cs, k);
        if (i > 0) { if (axb == 16) mu_leaf<16>(p[i - 1], inv, cs, kp); else mu_leaf<17>(p[i - 1], inv, cs, kp); }
        if (!ok || kp > k) bad = true;
    }
    if (bad) atomicOr(flag, 1);
}

// ---- radix-hashed voxel neighbour index -----------------------------------------------------------------------------------
// pcl::KdTreeFLANN::nearestKSearch(k = 5) (EstimationMapping.hpp:128,185) is replaced by a search over the cells of the cell-major map order (mu_leaf): a cell is
// 2^cs x 2^cs leaf columns (0.8 m x 0.8 m for both maps of the KITTI configuration), all z; its points are one contiguous run of the map, and the cells of a row
// follow each other in x. The DIRECTORY hashes a cell to a slot by the low 9 bits of its coordinates — slot = (cy & 511) << 9 | (cx & 511) — and holds the
// position of the first map point whose cell is >= that cell in the map's order (a lower bound, also for empty cells next to occupied ones), tagged with the
// step's epoch in the top byte. A query visits the rows cy of [qy - 1, qy + 1] and reads, per row, the two slots of cells cx_lo and cx_hi + 1: the points in
// between are ONE span of the map. Every map point within 1 m of the query lies in such a span, so the 5-NN is exact whenever the 5th squared distance is
// < 1 — the only case the reference uses (EstimationMapping.hpp:129,189). Slots the current step did not write carry an old tag: the span is empty.
// b_dir_build writes the directory from the map in one pass (a thread per point; a point whose cell differs from its predecessor's fills the slots of the cells
// in between, and S2B_DMARGIN cells before the first / after the last cell of a row, so that a span that touches an occupied cell always finds both its
// ends). No sort, no copy of the map, no pass over the table. Maps wider than 512 - 2 S2B_DMARGIN cells would alias: the host picks cs so that the crop box
// fits (s2b_cell_shift), and for a map that is not a voxel grid yet (as initialised) from the cloud's extent.
#define DIR_PT 8            // points per thread: every load of a thread is issued before the first is used, and a block covers 2048 points (the grid is sized for the capacity)
// pts: the map in cell-major order (or the cell-major-sorted copy of a map that is not a voxel grid yet), n points per stream
__global__ void b_dir_build(const float4 *pts_all, const int *n_all, int cap, float inv, int cs, unsigned tag, unsigned *dir_all) {
    const int sid = blockIdx.y, n = min(n_all[sid], cap), i0 = blockIdx.x * DIR_PT * blockDim.x + threadIdx.x;
    if (i0 >= n) return;
    const float4 *p = pts_all + (size_t)sid * cap;
    unsigned *T = dir_all + (size_t)sid * S2B_NBS;
    float2 q[DIR_PT], qp[DIR_PT];
#pragma unroll
    for (int u = 0; u < DIR_PT; u++) {
        const int ic = min(i0 + u * (int)blockDim.x, n - 1);
        const float4 *a = p + ic, *b = p + max(ic - 1, 0);
        q[u] = make_float2(a->x, a->y); qp[u] = make_float2(b->x, b->y);
    }
#pragma unroll
    for (int u = 0; u < DIR_PT; u++) {
        const int i = i0 + u * (int)blockDim.x;
        if (i < n) dir_point(T, tag, i, n, (cm_leaf(q[u].y, inv) >> cs) & 511, (cm_leaf(q[u].x, inv) >> cs) & 511, (cm_leaf(qp[u].y, inv) >> cs) & 511, (cm_leaf(qp[u].x, inv) >> cs) & 511);
    }
}
// the cell-major-sorted copy of a map that is not a voxel grid yet: point vals[g] of the stream to position g, its index in w (ties of the 5-NN go by it)
__global__ void b_gather_sorted(CSet in, const int *vals_all, float4 *sorted_all) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, sid = blockIdx.y;
    if (i >= min(in.n[sid], in.cap)) return;
    const size_t base = (size_t)sid * in.cap;
    const int v = vals_all[base + i];
    float4 q = in.p[base + v];
    q.w = __int_as_float(v);
    sorted_all[base + i] = q;
}
#ifndef KNN_FL
#define KNN_FL 8            // candidates in flight per lane and round
#endif
#define KNN_ROWS 4          // rows of cells walked as one sequence (cells of >= 2/3 m: at most four rows)
struct KnnSpans { int nrow, cylo, cxlo, cxhi1; };
__device__ __forceinline__ KnnSpans knn_rows(float qx, float qy, float inv, int cs) {
    KnnSpans r;
    r.cylo = cm_leaf(qy - S2B_QR, inv) >> cs; r.nrow = (cm_leaf(qy + S2B_QR, inv) >> cs) - r.cylo + 1;
    r.cxlo = cm_leaf(qx - S2B_QR, inv) >> cs; r.cxhi1 = (cm_leaf(qx + S2B_QR, inv) >> cs) + 1;
    return r;
}
// the span of row cy: [st, en) or an empty one when either slot was not written by this step / is inconsistent
__device__ __forceinline__ void knn_span(const unsigned *T, unsigned tag, int n, int cy, int cxlo, int cxhi1, int &st, int &en) {
    const unsigned a = T[dir_slot(cy, cxlo)], b = T[dir_slot(cy, cxhi1)];
    const int sa = (int)(a & 0xffffffu), sb = (int)(b & 0xffffffu);
    const bool ok = (a & 0xff000000u) == tag && (b & 0xff000000u) == tag && sa <= sb && sb <= n;
    st = ok ? sa : 0; en = ok ? sb : 0;
}
typedef unsigned knn_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 knn_bload(__amdgpu_buffer_rsrc_t rs, unsigned byte_off) {
    const knn_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    float4 r; r.x = __uint_as_float(v.x); r.y = __uint_as_float(v.y); r.z = __uint_as_float(v.z); r.w = __uint_as_float(v.w);
    return r;
}
// exact 5-NN within the rows of cells around the query: pos[] = positions in the cell-major array, ordered by squared distance; candidates at exactly equal distances are
// ordered by position here and reported through `tie` — b_associate_ties then redoes the query with the reference order (rare)
__device__ void knn5_cells(const float4 *sorted, const unsigned *T, unsigned tag, int n, float inv, int cs, float qx, float qy, float qz, int pos[5], float d2[5], bool &tie) {
#pragma unroll
    for (int k = 0; k < 5; k++) { pos[k] = -1; d2[k] = 3.0e38f; }
    float disc = 3.0e38f;                              // the smallest distance among the candidates that were turned away or pushed out of the list
    const KnnSpans R = knn_rows(qx, qy, inv, cs);
    // all span bounds first (independent loads), then the candidates eight at a time (unconditional loads): the lane keeps several loads in flight instead of
    // one dependent load per candidate. A lane past its last candidate reads entry 0 of the array — the SAME line for every such lane of the wave: a wave walks
    // as many rounds as its longest lane (6.4 against a mean of 1.3 for edge queries), and a load instruction costs the vector cache a cycle per distinct line
    int st[KNN_ROWS], en[KNN_ROWS];
#pragma unroll
    for (int r = 0; r < KNN_ROWS; r++) { if (r < R.nrow) knn_span(T, tag, n, R.cylo + r, R.cxlo, R.cxhi1, st[r], en[r]); else { st[r] = 0; en[r] = 0; } }
    // one candidate: sorted insertion with compile-time indices only (the five best stay in registers). The distances move by median-of-three (the list is sorted:
    // med3(d2[k - 1], d2[k], d) is the new entry k), the positions by selects; equal distances are NOT looked for here — see the end of the function
#define KNN_TRY(M, POS) { const float4 m = (M); const float ex = m.x - qx, ey = m.y - qy, ez = m.z - qz; \
        const float d = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez)); \
        if (d <= d2[4]) { bool lt[5]; const int ps_ = (POS); \
            _Pragma("unroll") for (int k = 0; k < 5; k++) lt[k] = d < d2[k]; \
            disc = __builtin_amdgcn_fmed3f(-3.0e38f, disc, lt[4] ? d2[4] : d);        /* min (fminf would canonicalise both operands first) */ \
            _Pragma("unroll") for (int k = 4; k >= 1; k--) { pos[k] = lt[k - 1] ? pos[k - 1] : (lt[k] ? ps_ : pos[k]); d2[k] = __builtin_amdgcn_fmed3f(d2[k - 1], d2[k], d); } \
            pos[0] = lt[0] ? ps_ : pos[0]; d2[0] = lt[0] ? d : d2[0]; } }
    {
        // the rows' spans are walked as ONE sequence (virtual index t -> position t + offset of its span) so that a batch never ends at a span boundary,
        // with the next batch's loads issued before this batch is ranked: the walk is bound by the latency of these gathers, not by the arithmetic
        const int n0 = en[0] - st[0], n01 = n0 + en[1] - st[1], n012 = n01 + en[2] - st[2], ntot = n012 + en[3] - st[3];
        const int o0 = st[0], o1 = st[1] - n0, o2 = st[2] - n01, o3 = st[3] - n012;
#define KNN_AT(t) ((t) + ((t) < n0 ? o0 : ((t) < n01 ? o1 : ((t) < n012 ? o2 : o3))))
        if (ntot > 0) {
            // the candidates through a buffer resource over the stream's map (round 5): a 32-bit byte offset instead of a 64-bit address per load (the kernel is bound by
            // instructions issued), and a lane past its last candidate asks for an offset beyond the buffer — the load returns zeros without touching memory
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)sorted, 0, n * 16, 0x00027000);
#define KNN_LD(t) knn_bload(rs, (t) < ntot ? 16u * (unsigned)KNN_AT(t) : 0x7ffffff0u)
            float4 cur[KNN_FL], nxt[KNN_FL];
#pragma unroll
            for (int u = 0; u < KNN_FL; u++) cur[u] = KNN_LD(u);
            for (int t0 = 0; t0 < ntot; t0 += KNN_FL) {
#pragma unroll
                for (int u = 0; u < KNN_FL; u++) { const int t = t0 + KNN_FL + u; nxt[u] = KNN_LD(t); }
#pragma unroll
                for (int u = 0; u < KNN_FL; u++) {
                    // a slot past the lane's last candidate holds zeros (the buffer load's answer): its distance is replaced by +inf, which no list admits — a select
                    // instead of the chain of nested `break`s (an exec-mask save, a branch and a restore per slot: the kernel is bound by instructions issued)
                    const int t = t0 + u;
                    const float4 m = cur[u];
                    const float ex = m.x - qx, ey = m.y - qy, ez = m.z - qz;
                    const float dd = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
                    const float d = t < ntot ? dd : __builtin_inff();
                    if (d <= d2[4]) { bool lt[5];                 // (KNN_TRY's insertion; the list holds the VIRTUAL index until the walk is over)
#pragma unroll
                        for (int k = 0; k < 5; k++) lt[k] = d < d2[k];
                        disc = __builtin_amdgcn_fmed3f(-3.0e38f, disc, lt[4] ? d2[4] : d);
#pragma unroll
                        for (int k = 4; k >= 1; k--) { pos[k] = lt[k - 1] ? pos[k - 1] : (lt[k] ? t : pos[k]); d2[k] = __builtin_amdgcn_fmed3f(d2[k - 1], d2[k], d); }
                        pos[0] = lt[0] ? t : pos[0]; d2[0] = lt[0] ? d : d2[0]; }
                }
#pragma unroll
                for (int u = 0; u < KNN_FL; u++) cur[u] = nxt[u];
            }
#pragma unroll
            for (int k = 0; k < 5; k++) if (pos[k] >= 0) pos[k] = KNN_AT(pos[k]);
        }
#undef KNN_LD
#undef KNN_AT
    }
    for (int r = KNN_ROWS; r < R.nrow; r++) {          // cells smaller than 2/3 m: the further rows one by one
        int s0, e0;
        knn_span(T, tag, n, R.cylo + r, R.cxlo, R.cxhi1, s0, e0);
        for (int j0 = s0; j0 < e0; j0 += KNN_FL) {
            float4 m4[KNN_FL];
#pragma unroll
            for (int u = 0; u < KNN_FL; u++) m4[u] = sorted[min(j0 + u, e0 - 1)];
#pragma unroll
            for (int u = 0; u < KNN_FL; u++) {
                if (j0 + u >= e0) break;
                KNN_TRY(m4[u], j0 + u)
            }
        }
    }
#undef KNN_TRY
    // Equal distances change the result only (i) between neighbours of the final list (their order) or (ii) between its last entry and a candidate that is not in it
    // (which of the two is kept): the list is strictly sorted otherwise and every other candidate is strictly farther, whatever the order they came in.
    tie = pos[4] >= 0 && disc == d2[4];
#pragma unroll
    for (int k = 0; k < 4; k++) tie = tie || (pos[k + 1] >= 0 && d2[k] == d2[k + 1]);
}
// The same search with the reference's order of equal distances: ascending map index in PCL's order (a linear scan keeps the first of equals) — for the cell-major map
// that is the PCL leaf key z | y | x of the candidate, then its position (points of one leaf keep their relative order); for the sorted copy of a map that is not a
// voxel grid yet the original index travels in w. Only queries that met a tie come here (b_associate_ties): plain loops, no attempt at speed.
__device__ void knn5_cells_exact(const float4 *sorted, const unsigned *T, unsigned tag, int n, float inv, int cs, bool w_is_index, float qx, float qy, float qz, int pos[5], float d2[5]) {
    unsigned long long ky[5];
    for (int k = 0; k < 5; k++) { pos[k] = -1; d2[k] = 3.0e38f; ky[k] = ~0ULL; }
    const KnnSpans R = knn_rows(qx, qy, inv, cs);
    for (int r = 0; r < R.nrow; r++) {
        int s0, e0;
        knn_span(T, tag, n, R.cylo + r, R.cxlo, R.cxhi1, s0, e0);
        for (int j = s0; j < e0; j++) {
            const float4 m = sorted[j];
            const float ex = m.x - qx, ey = m.y - qy, ez = m.z - qz;
            const float d = __fadd_rn(__fadd_rn(__fmul_rn(ex, ex), __fmul_rn(ey, ey)), __fmul_rn(ez, ez));
            const unsigned long long key = w_is_index ? (unsigned long long)(unsigned)__float_as_int(m.w)
                                                      : (((unsigned long long)(unsigned)cm_leaf(m.z, inv) << 40) | ((unsigned long long)(unsigned)cm_leaf(m.y, inv) << 20) | (unsigned long long)(unsigned)cm_leaf(m.x, inv));
            int at = 5;
            for (int k = 4; k >= 0; k--) if (d < d2[k] || (d == d2[k] && (key < ky[k] || (key == ky[k] && j < pos[k])))) at = k;
            if (at < 5) {
                for (int k = 4; k > at; k--) { d2[k] = d2[k - 1]; ky[k] = ky[k - 1]; pos[k] = pos[k - 1]; }
                d2[at] = d; ky[at] = key; pos[at] = j;
            }
        }
    }
}

// 3x3 symmetric eigen-decomposition by cyclic Jacobi: eigenvalues ascending, V columns. Every array index is a compile-time constant
// after unrolling, so A / V stay in registers (a dynamically indexed local array would live in scratch memory).
__device__ __forceinline__ void eig3(const double *Ain, double *w, double *V) {
    double A[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { A[k] = Ain[k]; V[k] = (k % 4 == 0) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; sweep++) {
        const double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
        if (off == 0.0) break;
#pragma unroll
        for (int pq = 0; pq < 3; pq++) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double apq = A[3 * p + q];
            if (apq == 0.0) continue;
            const double app = A[4 * p], aqq = A[4 * q];
            const double g = 100.0 * fabs(apq);
            if (sweep > 3 && fabs(app) + g == fabs(app) && fabs(aqq) + g == fabs(aqq)) { A[3 * p + q] = 0; A[3 * q + p] = 0; continue; }
            const double theta = (aqq - app) / (2.0 * apq);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
            for (int j = 0; j < 3; j++) { const double a = A[3 * p + j], b = A[3 * q + j]; A[3 * p + j] = c * a - sn * b; A[3 * q + j] = sn * a + c * b; }
#pragma unroll
            for (int i = 0; i < 3; i++) { const double a = A[3 * i + p], b = A[3 * i + q]; A[3 * i + p] = c * a - sn * b; A[3 * i + q] = sn * a + c * b; const double va = V[3 * i + p], vb = V[3 * i + q]; V[3 * i + p] = c * va - sn * vb; V[3 * i + q] = sn * va + c * vb; }
        }
    }
    double d0 = A[0], d1 = A[4], d2 = A[8];
    // ascending order by the comparators (0,1) (1,2) (0,1) of a bubble sort, columns of V move with their eigenvalue
#define EIG3_CSWAP(da, db, ca, cb) if (da > db) { const double t_ = da; da = db; db = t_; _Pragma("unroll") for (int r = 0; r < 3; r++) { const double v_ = V[3 * r + ca]; V[3 * r + ca] = V[3 * r + cb]; V[3 * r + cb] = v_; } }
    EIG3_CSWAP(d0, d1, 0, 1) EIG3_CSWAP(d1, d2, 1, 2) EIG3_CSWAP(d0, d1, 0, 1)
#undef EIG3_CSWAP
    w[0] = d0; w[1] = d1; w[2] = d2;
}
// 5x3 least squares by column-pivoted Householder QR (Eigen colPivHouseholderQr().solve); fully unrolled, static indices only
__device__ __forceinline__ void qr_solve_5x3(const double *Ain, const double *bin, double *x) {
    double a[5][3], b[5];
#pragma unroll
    for (int i = 0; i < 5; i++) { for (int j = 0; j < 3; j++) a[i][j] = Ain[3 * i + j]; b[i] = bin[i]; }
    int perm[3] = {0, 1, 2};
    double rdiag[3] = {0, 0, 0}, maxpivot = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        int best = k; double bn = -1, cn[3] = {0, 0, 0};
#pragma unroll
        for (int j = k; j < 3; j++) { double s = 0; for (int i = k; i < 5; i++) s += a[i][j] * a[i][j]; cn[j] = s; if (s > bn) { bn = s; best = j; } }
#pragma unroll
        for (int c = k + 1; c < 3; c++) if (best == c) {          // swap columns k <-> c (c is a compile-time constant here)
#pragma unroll
            for (int i = 0; i < 5; i++) { const double t = a[i][k]; a[i][k] = a[i][c]; a[i][c] = t; }
            const int t = perm[k]; perm[k] = perm[c]; perm[c] = t; cn[c] = cn[k]; cn[k] = bn;
        }
        const double nrm = sqrt(cn[k]);
        if (nrm == 0.0) { rdiag[k] = 0; continue; }
        const double alpha = a[k][k] > 0 ? -nrm : nrm;
        double v[5] = {0, 0, 0, 0, 0};
        v[k] = a[k][k] - alpha;
#pragma unroll
        for (int i = k + 1; i < 5; i++) v[i] = a[i][k];
        double vtv = 0;
#pragma unroll
        for (int i = k; i < 5; i++) vtv += v[i] * v[i];
        if (vtv > 0) {
#pragma unroll
            for (int j = k; j < 3; j++) { double s = 0; for (int i = k; i < 5; i++) s += v[i] * a[i][j]; s = 2 * s / vtv; for (int i = k; i < 5; i++) a[i][j] -= s * v[i]; }
            double s = 0;
#pragma unroll
            for (int i = k; i < 5; i++) s += v[i] * b[i];
            s = 2 * s / vtv;
#pragma unroll
            for (int i = k; i < 5; i++) b[i] -= s * v[i];
        }
        rdiag[k] = a[k][k];
        if (fabs(rdiag[k]) > maxpivot) maxpivot = fabs(rdiag[k]);
    }
    const double thresh = 2.220446049250313e-16 * 3.0 * maxpivot;
    int rank = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) if (fabs(rdiag[k]) > thresh) rank++;
    double z[3] = {0, 0, 0};
#pragma unroll
    for (int k = 2; k >= 0; k--) if (k < rank) {
        double s = b[k];
#pragma unroll
        for (int j = k + 1; j < 3; j++) if (j < rank) s -= a[k][j] * z[j];
        z[k] = s / a[k][k];
    }
    x[0] = x[1] = x[2] = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
#pragma unroll
        for (int c = 0; c < 3; c++) if (perm[k] == c) x[c] = z[k];
    }
}

// factor record, ONE aligned 64-byte sector per query: doubles 0..5 = edge: pa[3] pb[3] / surf: n[3] d 0 0; double 6 = the bits of the floats cp.x | cp.y << 32; double 7 = cp.z |
// kind << 32 (kind 0 invalid, 1 edge, 2 surf; the scan point cp IS a float). The LM solve re-reads every record five times per pass and is bound by that traffic: the
// former 80-byte records (+ a separate kind array) straddled two sectors each.
#define S2M_FREC 8
__device__ __forceinline__ double s2m_pack2(unsigned lo, unsigned hi) { return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)); }
// the line / plane fit of one query from its five neighbours (EdgeCostFactor :143-157, SurfCostFactor :198-213) and its 64-byte factor record
__device__ __forceinline__ void assoc_fit_write(const float4 *sorted, const int idx[5], bool gate, int is_surf, const float4 p, double *frec, int *fkind_slot) {
    int kind = 0;
    double rec[S2M_FREC] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (gate) {
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
                for (int a = 0; a < 3; a++) { rec[a] = 0.1 * V[3 * a + 2] + c[a]; rec[3 + a] = -0.1 * V[3 * a + 2] + c[a]; }
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
            if (ok) { kind = 2; rec[0] = nn[0]; rec[1] = nn[1]; rec[2] = nn[2]; rec[3] = d; }
        }
    }
    *fkind_slot = kind;                                          // the solve counts its factors from this array; the sweeps read the kind out of the record
    rec[6] = s2m_pack2(__float_as_uint(p.x), __float_as_uint(p.y)); rec[7] = s2m_pack2(__float_as_uint(p.z), (unsigned)kind);
    double2 *fo = reinterpret_cast<double2 *>(frec);
#pragma unroll
    for (int k = 0; k < 4; k++) fo[k] = make_double2(rec[2 * k], rec[2 * k + 1]);
}
struct AssocArgs {
    CSet ds; int is_surf; const int *n_ds_edge; const double *pose_all;
    const float4 *sorted_all; const int *n_map; int cap_map;      // the cell-major array the search runs on (the map itself, or the sorted copy of an unordered map)
    const unsigned *dir_all; unsigned tag; float inv; int cs; int w_is_index;
    const S2BRes *res; double *frec_all; int *fkind_all; int capq; int *tie_count;
};
#define S2M_KIND_TIE 3      // fkind of a query whose 5-NN met exactly equal distances: b_associate_ties redoes it in the reference's order (b_solve skips kinds other than 1, 2)
// one thread per query point of one stream. Edge queries write records [0, n_ds_edge), surf queries [n_ds_edge, n_ds_edge + n_ds_surf).
// one query i of the cloud described by a. TIES: the exact redo of a query that met equal distances
template <bool TIES>
__device__ __forceinline__ void assoc_one(const AssocArgs &a, int i, int sid, bool stamp) {
    const int slot = a.is_surf ? a.n_ds_edge[sid] + i : i;
    int *fk = a.fkind_all + (size_t)sid * a.capq + slot;
    const double *pose = a.pose_all + 24 * sid;
    const float4 *sorted = a.sorted_all + (size_t)sid * a.cap_map;
    const unsigned *T = a.dir_all + (size_t)sid * S2B_NBS;
    const int nmap = min(a.n_map[sid], a.cap_map);
    double *frec = a.frec_all + ((size_t)sid * a.capq + slot) * S2M_FREC;
    const float4 p = a.ds.p[(size_t)sid * a.ds.cap + i];
    const double cp[3] = {p.x, p.y, p.z};
    double pw[3];
    q_rot(q_load(pose), cp, pw);
    const float qx = (float)(pw[0] + pose[4]), qy = (float)(pw[1] + pose[5]), qz = (float)(pw[2] + pose[6]);
    int idx[5] = {-1, -1, -1, -1, -1}; float d2[5] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
#ifdef VILF_STAMPS
    const long long st0_ = __builtin_readcyclecounter();
    long long st1_ = st0_;
#endif
    if (nmap >= 5) {
        if (TIES) knn5_cells_exact(sorted, T, a.tag, nmap, a.inv, a.cs, a.w_is_index != 0, qx, qy, qz, idx, d2);
        else {
            bool tie;
            knn5_cells(sorted, T, a.tag, nmap, a.inv, a.cs, qx, qy, qz, idx, d2, tie);
            if (tie) { *fk = S2M_KIND_TIE; atomicAdd(a.tie_count + sid, 1); return; }
        }
    }
#ifdef VILF_STAMPS
    st1_ = __builtin_readcyclecounter();
#endif
    assoc_fit_write(sorted, idx, nmap >= 5 && d2[4] < 1.0f, a.is_surf, p, frec, fk);
#ifdef VILF_STAMPS
    if (stamp) { const int kid = 6 + a.is_surf; s2m_dbg[kid * 32 + 0] = st1_ - st0_; s2m_dbg[kid * 32 + 1] = __builtin_readcyclecounter() - st1_; s2m_dbg[kid * 32 + 30] = a.ds.n[sid]; }
#endif
    (void)stamp;
}
// ONE launch serves both query sets: blocks [0, nblk_edge) take the edge cloud against the edge map (arguments ae), the others the surf cloud (as).
__global__ void b_associate(AssocArgs ae, AssocArgs as, int nblk_edge) {
    const bool second = (int)blockIdx.x >= nblk_edge;
    const AssocArgs &a = second ? as : ae;
    const int i = ((int)blockIdx.x - (second ? nblk_edge : 0)) * blockDim.x + threadIdx.x, sid = blockIdx.y;
    if (i >= a.ds.n[sid] || !a.res[sid].do_opt) return;
    bool stamp = false;
#ifdef VILF_STAMPS
    stamp = blockIdx.y == S2M_STAMP_WG && (blockIdx.x == 2 || (int)blockIdx.x == nblk_edge + 2) && threadIdx.x == 0;
#endif
    assoc_one<false>(a, i, sid, stamp);
}
// the queries that met exactly equal distances, redone in the reference's order: one block per stream, which leaves at once unless its stream counted a tie
__global__ void b_associate_ties(AssocArgs ae, AssocArgs as) {
    const int sid = blockIdx.x;
    if (ae.tie_count[sid] == 0) return;
    const int ne = ae.ds.n[sid], nq = ne + as.ds.n[sid];
    const int *fkind = ae.fkind_all + (size_t)sid * ae.capq;
    for (int slot = threadIdx.x; slot < nq; slot += blockDim.x) {
        if (fkind[slot] != S2M_KIND_TIE) continue;
        if (slot < ne) assoc_one<true>(ae, slot, sid, false); else assoc_one<true>(as, slot - ne, sid, false);
    }
    __syncthreads();
    if (threadIdx.x == 0) ae.tie_count[sid] = 0;
}

// ---- the persistent LM solve -------------------------------------------------------------------------------------------

#define S2M_NT 256
#define S2M_LCAP 16384         // valid-factor slots listed in LDS by b_solve (32 KB); more valid factors than this: every slot is visited
#define S2M_NW (S2M_NT / 64)
// block sum with a fixed order: butterfly inside each wave, then the per-wave partials in wave order (two barriers)
__device__ __forceinline__ double wave_sum(double v) { return vilf_wave_sum64(v); }       // callers: all lanes active
__device__ double s2m_block_sum(double v, double *s_red) {
    const int tid = threadIdx.x;
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = v;
    __syncthreads();
    double r = 0;
#pragma unroll
    for (int w = 0; w < S2M_NW; w++) r += s_red[w];
    return r;
}
// cost (and, JAC: gradient g[6], hessian H[21] lower-packed) at pose x over all valid factors
template <bool JAC>
// list != nullptr: the slots of the VALID factors (edge factors first, then plane factors, ascending), n_edge / nfac counting list entries; nullptr: every query slot is
// visited and the kind in the record decides
__device__ void s2m_evaluate(const double *x, const double *frec, const unsigned short *list, int n_edge, int nfac, double huber_a, double *s_red, double *s_out /*28*/) {
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0;
    // one factor row: Huber-corrected J^T J (lower-packed), J^T r
#define S2M_ROW(Jrow, rk_) { const double rk = sw * (rk_); double jr[6]; \
        _Pragma("unroll") for (int c = 0; c < 6; c++) { jr[c] = sw * (Jrow)[c]; acc[21 + c] += jr[c] * rk; } \
        int e = 0; _Pragma("unroll") for (int a = 0; a < 6; a++) _Pragma("unroll") for (int b2 = 0; b2 <= a; b2++) acc[e++] += jr[a] * jr[b2]; }
    // The lane walks its factors i = tid, tid + 256, ... as before; the records [0, n_edge) hold the edge queries (kind 0 or 1), the rest the plane queries (kind 0 or
    // 2), so the walk is two loops with one code path each. A record is requested together with its kind, not after the kind test (that made every factor two
    // dependent memory round trips — eighteen factors per lane and sweep, most of the solve); the plane loop, lighter in registers, also keeps the next record in flight.
    // unpack: the scan point (three floats) and the kind from doubles 6, 7
#define S2M_UNPACK(R6, R7, CP, KIND) { const unsigned long long a_ = (unsigned long long)__double_as_longlong(R6), b_ = (unsigned long long)__double_as_longlong(R7); \
        CP[0] = (double)__uint_as_float((unsigned)a_); CP[1] = (double)__uint_as_float((unsigned)(a_ >> 32)); CP[2] = (double)__uint_as_float((unsigned)b_); KIND = (int)(b_ >> 32); }
    int i = threadIdx.x;
    for (; i < n_edge; i += S2M_NT) {
        const double2 *rp = reinterpret_cast<const double2 *>(frec + (size_t)(list ? (int)list[i] : i) * S2M_FREC);
        double2 rr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) rr[k] = rp[k];
        double cp[3]; int kind;
        S2M_UNPACK(rr[3].x, rr[3].y, cp, kind)
        if (kind != 1) continue;
        const double pa[3] = {rr[0].x, rr[0].y, rr[1].x}, pb[3] = {rr[1].y, rr[2].x, rr[2].y};
        double r[3], J[18];
        edge_eval<JAC>(x, cp, pa, pb, r, J);
        double rho0, sw;
        huber(r[0] * r[0] + r[1] * r[1] + r[2] * r[2], huber_a, rho0, sw);
        acc[27] += 0.5 * rho0;
        if (JAC) {
#pragma unroll
            for (int k = 0; k < 3; k++) S2M_ROW(J + 6 * k, r[k])
        }
    }
    if (i < nfac) {
        double2 rn[4];
        {
            const double2 *rp = reinterpret_cast<const double2 *>(frec + (size_t)(list ? (int)list[i] : i) * S2M_FREC);
#pragma unroll
            for (int k = 0; k < 4; k++) rn[k] = rp[k];
        }
        for (; i < nfac; i += S2M_NT) {
            const double nv[3] = {rn[0].x, rn[0].y, rn[1].x}, dd = rn[1].y;
            double cp[3]; int kind;
            S2M_UNPACK(rn[3].x, rn[3].y, cp, kind)
            {
                const int ic = min(i + S2M_NT, nfac - 1);
                const double2 *rp = reinterpret_cast<const double2 *>(frec + (size_t)(list ? (int)list[ic] : ic) * S2M_FREC);
#pragma unroll
                for (int k = 0; k < 4; k++) rn[k] = rp[k];
            }
            if (kind != 2) continue;
            double r[1], J[6];
            surf_eval<JAC>(x, cp, nv, dd, r, J);
            double rho0, sw;
            huber(r[0] * r[0], huber_a, rho0, sw);
            acc[27] += 0.5 * rho0;
            if (JAC) S2M_ROW(J, r[0])
        }
    }
#undef S2M_UNPACK
#undef S2M_ROW
    // 28 sums at once: wave butterflies, per-wave partials to LDS, thread k adds the partials of sum k in wave order
    __syncthreads();
#pragma unroll
    for (int k = JAC ? 0 : 27; k < 28; k++) { const double v = wave_sum(acc[k]); if ((threadIdx.x & 63) == 0) s_red[(threadIdx.x >> 6) * 28 + k] = v; }
    __syncthreads();
    if (threadIdx.x < 28 && (JAC || threadIdx.x == 27)) { double r = 0; for (int w = 0; w < S2M_NW; w++) r += s_red[w * 28 + threadIdx.x]; s_out[threadIdx.x] = r; }
    __syncthreads();
}
// ONE persistent 256-thread workgroup per stream; the optimised pose is written back to the stream's pose slot.
#ifndef S2M_SOLVE_WPE
#define S2M_SOLVE_WPE 2
#endif
__global__ __launch_bounds__(S2M_NT, S2M_SOLVE_WPE) void b_solve(double *pose_all, const double *frec_all, const int *fkind_all, int capq, const int *n_ds_edge, const int *n_ds_surf, double huber_a, int max_it, int pass, S2BRes *res) {
    const int sid = blockIdx.x;
    if (!res[sid].do_opt) return;
    double *pose_in = pose_all + 24 * sid;
    const double *frec = frec_all + (size_t)sid * capq * S2M_FREC;
    const int *fkind = fkind_all + (size_t)sid * capq;
    const int n_edge_q = n_ds_edge[sid], n_surf_q = n_ds_surf[sid];
    S2BRes *out = res + sid;
    __shared__ double s_red[S2M_NW * 28], s_ev[28], s_cand[28], s_x[7], s_c[7], s_scale[6], s_diag[6], s_step[6];
    __shared__ int s_ctl[4];
    const int tid = threadIdx.x, nfac = n_edge_q + n_surf_q;
#ifdef VILF_STAMPS
    const long long sb_t0 = __builtin_readcyclecounter();
#endif
    if (tid < 7) s_x[tid] = pose_in[tid];
    // The valid factors' slots as a compact list in LDS (about a third of the queries have no valid factor; the five sweeps of a pass are bound by the traffic of the
    // records): every thread counts the edge / plane factors of its contiguous slice of the slots, a block scan gives the offsets, a second walk over the slice fills
    // the list — edge factors first, both parts ascending. Falls back to visiting every slot when the list would not fit.
    __shared__ unsigned short s_list[S2M_LCAP];
    __shared__ int s_scan[S2M_NW];
    const int per = (nfac + S2M_NT - 1) / S2M_NT, a0 = min(nfac, tid * per), a1 = min(nfac, a0 + per);
    int ne = 0, ns = 0;
    for (int i = a0; i < a1; i++) { const int k = fkind[i]; if (k == 1) ne++; else if (k == 2) ns++; }
    int incl = ne | (ns << 16);                                   // two packed counts (<= 65535 each: slots are 16-bit)
    const int mine = incl, lane_ = tid & 63, wave_ = tid >> 6;
#pragma unroll
    for (int of = 1; of < 64; of <<= 1) { const int u = __shfl_up(incl, of, 64); if (lane_ >= of) incl += u; }
    if (lane_ == 63) s_scan[wave_] = incl;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < S2M_NW; k++) { if (k < wave_) off += s_scan[k]; tot += s_scan[k]; }
    const int tne = tot & 0xffff, tns = tot >> 16, excl = off + incl - mine;
    if (tne + tns == 0) { if (tid == 0) { out->cost[pass] = 0; out->its[pass] = 0; out->nfe[pass] = 0; out->nfs[pass] = 0; } return; }
    const bool use_list = tne + tns <= S2M_LCAP && nfac <= 65535;
    if (use_list) {
        int pe = excl & 0xffff, ps = tne + (excl >> 16);
        for (int i = a0; i < a1; i++) { const int k = fkind[i]; if (k == 1) s_list[pe++] = (unsigned short)i; else if (k == 2) s_list[ps++] = (unsigned short)i; }
    }
    const unsigned short *list = use_list ? s_list : nullptr;
    const int ev_ne = use_list ? tne : min(n_edge_q, nfac), ev_n = use_list ? tne + tns : nfac;
    __syncthreads();
#ifdef VILF_STAMPS
    long long sb_t1 = __builtin_readcyclecounter(), sb_step = 0, sb_eval = 0, sb_tail = 0, sb_t;
#define SB_T0() sb_t = __builtin_readcyclecounter()
#define SB_ADD(v) v += __builtin_readcyclecounter() - sb_t
#else
#define SB_T0() do { } while (0)
#define SB_ADD(v) do { } while (0)
#endif
    SB_T0();
    s2m_evaluate<true>(s_x, frec, list, ev_ne, ev_n, huber_a, s_red, s_ev);
    SB_ADD(sb_eval);
    // wave-0 scalars of the trust-region loop (trust_region_minimizer.cc + levenberg_marquardt_strategy.cc)
    double x_cost = s_ev[27], radius = 1e4, decrease_factor = 2.0, x_norm = 0, mcc = 0;
    bool reuse_diagonal = false;
    int iteration = 0, invalid = 0;
    if (tid == 0) {
        for (int c = 0; c < 6; c++) { const int dd = c * (c + 1) / 2 + c; s_scale[c] = 1.0 / (1.0 + sqrt(s_ev[dd])); }
    }
    for (int k = 0; k < 7; k++) x_norm += s_x[k] * s_x[k];
    x_norm = sqrt(x_norm);
    __syncthreads();
    for (;;) {
        SB_T0();
        // The trust-region step of an iteration, by WAVE 0 (every lane runs the scalar part with the same values; LDS is written by one lane). The 6 x 6 part —
        // LevenbergMarquardtStrategy::ComputeStep on the Jacobi-scaled system: Cholesky, two triangular solves, the model cost change — has a lane per ROW in registers
        // and passes pivots / solved entries between lanes by v_readlane. Until round 5 thread 0 did it alone on 6 x 6 arrays in LDS (so that they would not set the
        // kernel's register count): some 400 dependent LDS round trips and 35 IEEE divisions / square roots per iteration, ~45 k cycles of a ~70 k-cycle iteration — the
        // larger part of b_solve. Every entry sees the same operations in the same order as before (left- and right-looking Cholesky subtract the same products in
        // the same k order): results to the bit.
        if (wave_ == 0) {
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
                const int r = min(lane_, 5);                       // lane r < 6 = row r (lanes 6 .. 63 shadow row 5)
                double hs[6], v[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { const int a = r > c ? r : c, b2 = r > c ? c : r; hs[c] = s_ev[a * (a + 1) / 2 + b2] * s_scale[a] * s_scale[b2]; }
                const double gs_r = s_ev[21 + r] * s_scale[r];
                double dr = 0;
#pragma unroll
                for (int c = 0; c < 6; c++) if (c == r) dr = hs[c];
                dr = reuse_diagonal ? s_diag[r] : fmin(fmax(dr, 1e-6), 1e32);
                if (!reuse_diagonal && lane_ < 6) s_diag[r] = dr;
#pragma unroll
                for (int c = 0; c < 6; c++) v[c] = hs[c];
                {
                    const double addv = dr / radius;
#pragma unroll
                    for (int c = 0; c < 6; c++) if (c == r) v[c] += addv;
                }
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    if (!ok) continue;
                    const double sd = readlane_f64(v[j], j);
                    if (!(sd > 0)) { ok = false; continue; }
                    const double l = sqrt(sd);
                    const double lij = (r == j) ? l : v[j] / l;
                    v[j] = lij;
#pragma unroll
                    for (int c = j + 1; c < 6; c++) { const double lcj = readlane_f64(lij, c); v[c] = v[c] - lij * lcj; }
                }
                reuse_diagonal = true;
                if (ok) {
                    double acc = gs_r, yv = 0;
#pragma unroll
                    for (int k = 0; k < 6; k++) {                 // forward: y[k] = (gs[k] - sum_(t < k) L[k][t] y[t]) / L[k][k]
                        const double yk = readlane_f64(acc, k) / readlane_f64(v[k], k);
                        if (r == k) yv = yk;
                        acc = acc - v[k] * yk;
                    }
                    double yb[6];
#pragma unroll
                    for (int i = 5; i >= 0; i--) {                // backward: y[i] = (y[i] - sum_(k > i) L[k][i] y[k]) / L[i][i], k ascending
                        double sd = readlane_f64(yv, i);
#pragma unroll
                        for (int k = i + 1; k < 6; k++) sd = sd - readlane_f64(v[i], k) * yb[k];
                        yb[i] = sd / readlane_f64(v[i], i);
                    }
                    double step[6];
#pragma unroll
                    for (int a = 0; a < 6; a++) step[a] = -yb[a];
                    if (lane_ == 0) {
#pragma unroll
                        for (int a = 0; a < 6; a++) s_step[a] = step[a];
                    }
                    double t_r = 0;
#pragma unroll
                    for (int b2 = 0; b2 < 6; b2++) t_r += hs[b2] * step[b2];
                    double sg = 0, sHs = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++) { sg += step[a] * readlane_f64(gs_r, a); sHs += step[a] * readlane_f64(t_r, a); }
                    mcc = -sg - 0.5 * sHs;
                    if (mcc > 0) valid = 1;
                    if (valid) {
                        double d[6];
#pragma unroll
                        for (int c = 0; c < 6; c++) d[c] = step[c] * s_scale[c];
                        double xc[7];
                        se3_plus(s_x, d, xc);
                        if (lane_ == 0) { for (int k = 0; k < 7; k++) s_c[k] = xc[k]; }
                    }
                }
                if (!valid) {   // StepIsInvalid -> StepRejected(0)
                    invalid++;
                    radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
                    if (invalid >= 5) go = 0;
                } else invalid = 0;
            }
            if (lane_ == 0) { s_ctl[0] = go; s_ctl[1] = valid; s_ctl[2] = (iteration >= max_it) ? 1 : 0; }
        }
        __syncthreads();
        const int go = s_ctl[0], valid = s_ctl[1], last = s_ctl[2];
        __syncthreads();
        SB_ADD(sb_step);
        if (!go) break;
        if (!valid) continue;
        SB_T0();
        // cost AND linearisation at the candidate in one sweep over the factor records: an accepted step (the usual case) then needs no
        // second sweep; a rejected one leaves s_ev (the linearisation at x) untouched
        // ... except at the last iteration of the budget: nothing would use that linearisation, the cost alone decides the step
        if (last) s2m_evaluate<false>(s_c, frec, list, ev_ne, ev_n, huber_a, s_red, s_cand);
        else s2m_evaluate<true>(s_c, frec, list, ev_ne, ev_n, huber_a, s_red, s_cand);
        SB_ADD(sb_eval);
        SB_T0();
        if (wave_ == 0) {                  // (uniform over the wave, like the step above: its scalars — radius, x_cost, x_norm ... — live in every lane of wave 0)
            const double cand = s_cand[27];
            double sn = 0, xc[7];
            for (int k = 0; k < 7; k++) { xc[k] = s_c[k]; sn += (s_x[k] - xc[k]) * (s_x[k] - xc[k]); }
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
                    x_norm = 0; for (int k = 0; k < 7; k++) x_norm += xc[k] * xc[k];
                    x_norm = sqrt(x_norm);
                    x_cost = cand;
                } else { radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true; }
            }
            if (lane_ == 0) { s_ctl[2] = stop; s_ctl[3] = accept; }
        }
        __syncthreads();
        const int stop = s_ctl[2], accept = s_ctl[3];
        __syncthreads();
        if (stop) break;
        if (accept) {     // a rejected step keeps the linearisation at x (s_ev) for the next ComputeStep
            if (tid < 28) s_ev[tid] = s_cand[tid];
            if (tid < 7) s_x[tid] = s_c[tid];
            __syncthreads();
        }
        SB_ADD(sb_tail);
    }
#ifdef VILF_STAMPS
    if (blockIdx.x == S2M_STAMP_WG && tid == 0 && pass == 0) {
        s2m_dbg[8 * 32 + 0] = __builtin_readcyclecounter() - sb_t0; s2m_dbg[8 * 32 + 1] = sb_t1 - sb_t0; s2m_dbg[8 * 32 + 2] = sb_eval; s2m_dbg[8 * 32 + 3] = sb_step; s2m_dbg[8 * 32 + 4] = sb_tail;
        s2m_dbg[8 * 32 + 5] = iteration; s2m_dbg[8 * 32 + 6] = ev_n; s2m_dbg[8 * 32 + 7] = nfac;
    }
#endif
    if (tid == 0) {
        for (int k = 0; k < 7; k++) pose_in[k] = s_x[k];
        out->cost[pass] = x_cost; out->its[pass] = iteration; out->nfe[pass] = tne; out->nfs[pass] = tns;
    }
}

// ---- createSubMap (:298-352): transform + append, crop box, voxel grid -------------------------------------------------
__global__ void b_transform_append(CSet ds, const double *pose_all, CSet map, int *err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, sid = blockIdx.y;
    if (i >= ds.n[sid]) return;
    const double *pose = pose_all + 24 * sid;
    const int dst = map.n[sid] + i;
    if (dst >= map.cap) { atomicOr(err + sid, S2B_ERR_MAPCAP); return; }
    const float4 p = ds.p[(size_t)sid * ds.cap + i];
    const double cp[3] = {p.x, p.y, p.z};
    double pw[3];
    q_rot(q_load(pose), cp, pw);
    map.p[(size_t)sid * map.cap + dst] = make_float4((float)(pw[0] + pose[4]), (float)(pw[1] + pose[5]), (float)(pw[2] + pose[6]), p.w);
}
__global__ void b_bump(CSet map, CSet ds, int *n_old, int S) {
    const int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid < S) { n_old[sid] = map.n[sid]; map.n[sid] = min(map.n[sid] + ds.n[sid], map.cap); }
}
// globalOdom_est = globalOdom * (globalOdom_last^-1 * globalOdom) (EstimationMapping.hpp:238-243), rotation via matrices
__global__ void b_predict(double *pose_all, int S) {
    const int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= S) return;
    double *pose = pose_all + 24 * sid, *pose_last = pose + 8, *prev = pose + 16;
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
// optimisation gate of optimation_processing (:254: both maps big enough) + result reset
__global__ void b_gate(const int *n_map_e, const int *n_map_s, const int *n_ds_e, const int *n_ds_s, S2BRes *res, int *err, int S) {
    const int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= S) return;
    S2BRes r;
    memset(&r, 0, sizeof(r));
    r.n_ds[0] = n_ds_e[sid]; r.n_ds[1] = n_ds_s[sid];
    r.do_opt = (n_map_e[sid] > 10 && n_map_s[sid] > 50 && n_ds_e[sid] + n_ds_s[sid] > 0) ? 1 : 0;
    res[sid] = r;
}
__global__ void b_finish(const double *pose_all, const int *n_map_e, const int *n_map_s, const int *err, S2BRes *res, int S) {
    const int sid = blockIdx.x * blockDim.x + threadIdx.x;
    if (sid >= S) return;
    const double *pose = pose_all + 24 * sid;
    for (int k = 0; k < 7; k++) { res[sid].pose[k] = pose[k]; res[sid].prev[k] = pose[16 + k]; }
    res[sid].map_n[0] = n_map_e[sid]; res[sid].map_n[1] = n_map_s[sid];
    res[sid].err = err[sid];
}

// ---------------------------------------------------------------------------------------------------------------------
// host
struct S2B {
    int S = 0;
    int capScan[2] = {0, 0}, capMap[2] = {0, 0};
    DBuf scan[2], nScan[2], ds[2], nDs[2], map[2], mapAlt[2], nMap[2], tmpB, nTmp, sorted[2], bstart[2], bcnt;   // map: current local maps; mapAlt: where the next step writes its maps
    DBuf nOld, mOld;                                     // map point counts before the append / surviving the crop (unsorted-map path)
    DBuf keys, keys2, vals, vals2, temp, mm, frec, fkind, pose, res, err, bits;
    DBuf map0[2], nMap0[2], pose0, muT, tileHeads;
    DBuf svlist[2];                              // streams b_scan_voxel_runs hands back to b_scan_voxel: [0] count, [1..] stream ids
    DBuf bigmap[2]; std::vector<int> h_big[2];   // streams whose scan cloud does not fit the in-LDS voxel grid (b_scan_voxel): the global-sort path takes them as a sub-batch
    // The neighbour directory travels with its map: bstart pairs with map, bstartAlt with mapAlt (the map update writes the directory of the map it emits), dir0 with map0.
    DBuf bstartAlt[2], dir0[2], fixq[2];
    bool dir_ok[2] = {false, false}, snap_dir_ok[2] = {false, false};      // bstart[w] is the directory of map[w] for every stream (tag dir_tag[w])
    unsigned dir_tag[2] = {0, 0}, snap_dir_tag[2] = {0, 0};
    unsigned next_tag() { epoch++; return (epoch % 255u + 1u) << 24; }
    int order_state[2] = {0, 0}, snap_order[2] = {0, 0};   // local maps in ascending (cell-major) leaf order? 0 unknown, 1 yes (every step leaves them so), 2 no (as initialised)
    int cs_cfg[2] = {0, 0};            // cell shift of the maps' cell-major order (s2b_cell_shift of the leaf size and the crop box)
    int cs_idx[2] = {0, 0};            // ... of this step's neighbour directory (larger for an unordered map whose extent needs it)
    bool idx_copy[2] = {false, false}; // this step's search runs on the cell-major-sorted COPY of an unordered map (w = original index) instead of the map itself
    unsigned epoch = 0;                // directory entries carry (epoch % 255 + 1) << 24: a slot this step did not write reads as empty
    // how many leading points of a stream's map are in cell-major order and therefore handed out in PCL order by get_map (INT_MAX: all of it — the state after a step)
    std::vector<int> h_cmn[2], snap_cmn[2];
    bool has_snapshot = false, scan_dirty = true;
    bool snap_live = false;            // the snapshot's maps still live in a map / mapAlt buffer (rewind = pointer swap, no copy)
    void *snap_ptr[2] = {nullptr, nullptr};
    size_t temp_bytes = 0, work_n = 0;
    std::vector<int> h_nScan[2], h_nMap[2];
    std::vector<S2BRes> h_res;
    CSet cs_scan(int w) { return CSet{scan[w].as<float4>(), nScan[w].as<int>(), capScan[w]}; }
    CSet cs_ds(int w) { return CSet{ds[w].as<float4>(), nDs[w].as<int>(), capScan[w]}; }
    CSet cs_map(int w) { return CSet{map[w].as<float4>(), nMap[w].as<int>(), capMap[w]}; }
    CSet cs_tmp(int w) { return CSet{tmpB.as<float4>(), nTmp.as<int>(), capMap[w]}; }
    CSet cs_mapout(int w) { return CSet{mapAlt[w].as<float4>(), nMap[w].as<int>(), capMap[w]}; }
    void release() {
        DBuf *all[] = {&scan[0], &scan[1], &nScan[0], &nScan[1], &ds[0], &ds[1], &nDs[0], &nDs[1], &map[0], &map[1], &mapAlt[0], &mapAlt[1], &nMap[0], &nMap[1], &tmpB, &nTmp, &sorted[0], &sorted[1],
                       &bstart[0], &bstart[1], &bcnt, &nOld, &mOld, &keys, &keys2, &vals, &vals2, &temp, &mm, &frec, &fkind, &pose, &res, &err, &bits,
                       &map0[0], &map0[1], &nMap0[0], &nMap0[1], &pose0, &muT, &tileHeads, &bstartAlt[0], &bstartAlt[1], &dir0[0], &dir0[1], &fixq[0], &fixq[1], &bigmap[0], &bigmap[1]};
        for (DBuf *b : all) b->release();
    }
};

void vilf_s2m_release(vilf_handle *h) {
    for (S2B **pc : {&h->s2m, &h->s2b}) if (*pc) { (*pc)->release(); delete *pc; *pc = nullptr; }
}

// profiling (vilf_set_profiling): HIP events between groups of launches on the handle's stream.
// group 0 voxel grid, 1 radix sort (rocPRIM), 2 neighbour index (cell keys, hash build), 3 associate (5-NN + fits), 4 LM solve,
// 5 sub-map (append, crop, compact), 6 other
static void s2m_prof_mark(vilf_handle *h, int group) {       // the launches since the previous mark belong to `group`
    hipEvent_t e = vilf_prof_event(h);
    if (!h->s2m_groups.empty()) vilf_prof_span(h, h->s2m_ev.back(), e, &h->s2m_ms[group], &h->s2m_launches[group]);
    h->s2m_ev.push_back(e);
    h->s2m_groups.push_back(group);
}
#define PROF(g) if (h->profiling) s2m_prof_mark(h, (g));
#define GRID2(cap, S) dim3(((cap) + 255) / 256, (S)), dim3(256)
#define GRIDS(S) dim3(((S) + 63) / 64), dim3(64)

static int sbits_of(int S) { int b = 0; while ((1 << b) < S) b++; return b; }
// the cells of a local map: 2^cs leaves wide, at least half a metre (at most six rows of cells per query), and the crop box plus the directory's margins within its 512 slots per axis
static int s2b_cell_shift(float leaf, double half) {
    int cs = 0;
    while (cs < 15 && ((double)leaf * (1 << cs) < 0.5 || 2.0 * half / ((double)leaf * (1 << cs)) + 2.0 + 2 * S2B_DMARGIN > (double)(1 << S2B_DB))) cs++;
    return cs;
}

// (Re)size the context. Map contents, counters and poses survive a capacity growth; a change of S starts from scratch.
static int s2b_reserve(vilf_handle *h, S2B *c, int S, int capScanE, int capScanS, int capMapE, int capMapS) {
    const int wantScan[2] = {std::max(capScanE, 1), std::max(capScanS, 1)}, wantMap[2] = {std::max(capMapE, 1), std::max(capMapS, 1)};
    if (S != c->S) {
        c->release();
        c->S = S; c->capScan[0] = c->capScan[1] = c->capMap[0] = c->capMap[1] = 0; c->work_n = 0; c->has_snapshot = false; c->order_state[0] = c->order_state[1] = 0;
        c->h_big[0].clear(); c->h_big[1].clear();
        if (!c->pose.ensure((size_t)S * 24 * 8) || !c->res.ensure((size_t)S * sizeof(S2BRes)) || !c->err.ensure((size_t)S * 4) || !c->mm.ensure((size_t)S * sizeof(MinMax)) || !c->nTmp.ensure((size_t)S * 4) || !c->bits.ensure(64) || !c->nOld.ensure((size_t)S * 4) || !c->mOld.ensure((size_t)S * 4) || !c->bcnt.ensure((size_t)S * 4)) return VILF_ERR_DEVICE;
        HIPCHECK(h, hipMemsetAsync(c->bcnt.p, 0, (size_t)S * 4, h->stream));
        for (int w = 0; w < 2; w++) {
            if (!c->nScan[w].ensure((size_t)S * 4) || !c->nDs[w].ensure((size_t)S * 4) || !c->nMap[w].ensure((size_t)S * 4) || !c->bstart[w].ensure((size_t)S * S2B_NBS * 4)) return VILF_ERR_DEVICE;
            if (!c->bstartAlt[w].ensure((size_t)S * S2B_NBS * 4)) return VILF_ERR_DEVICE;
            HIPCHECK(h, hipMemsetAsync(c->bstart[w].p, 0, (size_t)S * S2B_NBS * 4, h->stream));      // tag 0: no slot is valid yet
            HIPCHECK(h, hipMemsetAsync(c->bstartAlt[w].p, 0, (size_t)S * S2B_NBS * 4, h->stream));
            c->dir_ok[w] = false;
            c->h_cmn[w].assign(S, 0);
            c->cs_cfg[w] = s2b_cell_shift((float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size), h->opts.s2m_crop_half);
            HIPCHECK(h, hipMemsetAsync(c->nScan[w].p, 0, (size_t)S * 4, h->stream));
            HIPCHECK(h, hipMemsetAsync(c->nDs[w].p, 0, (size_t)S * 4, h->stream));
            HIPCHECK(h, hipMemsetAsync(c->nMap[w].p, 0, (size_t)S * 4, h->stream));
            c->h_nScan[w].assign(S, 0); c->h_nMap[w].assign(S, 0);
        }
        std::vector<double> ident((size_t)S * 24, 0.0);
        for (int s = 0; s < S; s++) { ident[24 * s + 3] = 1.0; ident[24 * s + 11] = 1.0; ident[24 * s + 19] = 1.0; }
        HIPCHECK(h, hipMemcpyAsync(c->pose.p, ident.data(), ident.size() * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipMemsetAsync(c->err.p, 0, (size_t)S * 4, h->stream));
        HIPCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(b_scan_voxel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, SV_MAXPTS32 * 8 + 8192));
        HIPCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(b_scan_voxel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, SV_MAXPTS24 * 7 + 8192));
        HIPCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(b_scan_voxel_runs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sr_lds_bytes(SR_RC)));
        HIPCHECK(h, hipFuncSetAttribute(reinterpret_cast<const void *>(b_map_update<false>), hipFuncAttributeMaxDynamicSharedMemorySize, MU_LDS_TAIL * 8 + MU_TILE * 12));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        c->h_res.assign(S, S2BRes{});
    }
    bool grew = false;
    for (int w = 0; w < 2; w++) {
        if (wantScan[w] > c->capScan[w]) {
            const int nc = c->capScan[w] ? std::max(wantScan[w], 2 * c->capScan[w]) : wantScan[w];
            if (!c->scan[w].ensure((size_t)S * nc * 16) || !c->ds[w].ensure((size_t)S * nc * 16)) return VILF_ERR_DEVICE;
            c->capScan[w] = nc; grew = true; c->scan_dirty = true;
        }
        if (wantMap[w] > c->capMap[w]) {
            const int oc = c->capMap[w], nc = oc ? std::max(wantMap[w], 2 * oc) : wantMap[w];
            if (nc >= (1 << 24)) { h->err = "scan2map: local-map capacity beyond 2^24 points per stream (directory positions are 24 bits)"; return VILF_ERR_UNSUPPORTED; }
            DBuf nb;
            if (!nb.ensure((size_t)S * nc * 16)) return VILF_ERR_DEVICE;
            if (oc) {
                HIPCHECK(h, hipMemcpy2DAsync(nb.p, (size_t)nc * 16, c->map[w].p, (size_t)oc * 16, (size_t)oc * 16, S, hipMemcpyDeviceToDevice, h->stream));
                HIPCHECK(h, hipStreamSynchronize(h->stream));
            }
            c->map[w].release();
            c->map[w] = nb;
            c->capMap[w] = nc; grew = true; c->has_snapshot = false; c->snap_live = false;
            if (!c->sorted[w].ensure((size_t)S * nc * 16) || !c->mapAlt[w].ensure((size_t)S * nc * 16)) return VILF_ERR_DEVICE;
            if (!c->fixq[w].ensure((size_t)S * nc * 4)) return VILF_ERR_DEVICE;
        }
    }
    if (grew) {
        const int maxMap = std::max(c->capMap[0], c->capMap[1]);
        const size_t n = (size_t)S * std::max(std::max(c->capScan[0], c->capScan[1]), maxMap);
        if (!c->tmpB.ensure((size_t)S * maxMap * 16)) return VILF_ERR_DEVICE;
        const size_t capq = (size_t)c->capScan[0] + c->capScan[1];
        if (!c->frec.ensure((size_t)S * capq * S2M_FREC * 8) || !c->fkind.ensure((size_t)S * capq * 4)) return VILF_ERR_DEVICE;
        if (n > c->work_n) {
            if (!c->keys.ensure(n * 8) || !c->keys2.ensure(n * 8) || !c->vals.ensure(n * 4) || !c->vals2.ensure(n * 4)) return VILF_ERR_DEVICE;
            const size_t need = vilf_sort_temp_bytes(n, 8) + 256;
            if (!c->temp.ensure(need)) return VILF_ERR_DEVICE;
            c->temp_bytes = c->temp.cap;
            c->work_n = n;
        }
    }
    return VILF_OK;
}

// keys (PCL leaf index, or the maps' cell-major order when cs >= 0) of every stream's cloud + ONE stable radix sort over all streams: (sorted keys, point indices) land in
// keys2 / vals2. Host round trip: the key width (and, for *cs_fit, the cloud extents) come back from the device. Returns the key type used through *wide.
static int s2b_sort_keys(vilf_handle *h, S2B *c, CSet in, float leaf, int cs, int *cs_fit, bool *wide, int nrows = -1) {
    const int S = nrows < 0 ? c->S : nrows;
    const float inv = 1.0f / leaf;
    int hb[8];
    HIPCHECK(h, hipMemsetAsync(c->bits.p, 0, 32, h->stream));
    hipLaunchKernelGGL(b_minmax, dim3(S), dim3(1024), 0, h->stream, in, inv, c->mm.as<MinMax>(), c->bits.as<int>());
    HIPCHECK(h, hipMemcpyAsync(hb, c->bits.p, 32, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    int vbits = std::max(hb[0], 1);
    if (cs_fit) {             // an unordered map's cells: at least the configured size, and large enough for the cloud's extent to fit the directory without aliasing
        while (cs < 15 && ((std::max(hb[4], hb[5]) >> cs) + 2 + 2 * S2B_DMARGIN > (1 << S2B_DB))) cs++;
        *cs_fit = cs;
    }
    if (cs >= 0) {            // bound of the cell-major key over all streams (every factor maximised on its own)
        const long long bound = ((long long)((hb[5] >> cs) + 2) * ((hb[4] >> cs) + 2) * std::max(hb[6], 1)) << (2 * cs);
        vbits = 1; while (vbits < 62 && (1LL << vbits) < bound) vbits++;
    }
    if (vbits + sbits_of(S) > 63) { h->err = "scan2map: voxel index too wide (leaf size too small for the cloud extent)"; return VILF_ERR_UNSUPPORTED; }
    PROF(0)
    const int kbits = vbits + sbits_of(S);
    size_t tb = c->temp_bytes;
    *wide = kbits > 32;                                           // 32-bit keys: a third less sort traffic
    if (!*wide) {
        unsigned int *k1 = c->keys.as<unsigned int>(), *k2 = c->keys2.as<unsigned int>();
        hipLaunchKernelGGL(b_voxel_keys<unsigned int>, GRID2(in.cap, S), 0, h->stream, in, inv, c->mm.as<MinMax>(), k1, c->vals.as<int>(), c->err.as<int>(), vbits, cs);
        PROF(0)
        if (vilf_sort_pairs_u32(h->stream, c->temp.p, tb, k1, k2, c->vals.as<int>(), c->vals2.as<int>(), (size_t)S * in.cap, kbits) != 0) { h->err = "scan2map: radix sort failed"; return VILF_ERR_DEVICE; }
    } else {
        unsigned long long *k1 = c->keys.as<unsigned long long>(), *k2 = c->keys2.as<unsigned long long>();
        hipLaunchKernelGGL(b_voxel_keys<unsigned long long>, GRID2(in.cap, S), 0, h->stream, in, inv, c->mm.as<MinMax>(), k1, c->vals.as<int>(), c->err.as<int>(), vbits, cs);
        PROF(0)
        if (vilf_sort_pairs_u64(h->stream, c->temp.p, tb, k1, k2, c->vals.as<int>(), c->vals2.as<int>(), (size_t)S * in.cap, kbits) != 0) { h->err = "scan2map: radix sort failed"; return VILF_ERR_DEVICE; }
    }
    PROF(1)
    return VILF_OK;
}
// pcl::VoxelGrid over every stream by ONE global sort: in -> out (device counters), leaves in PCL order (cs < 0: scan clouds too large for the in-LDS grid, b_scan_voxel)
// or in the maps' cell-major order (cs >= 0: local maps that are not voxel grids yet — b_map_update needs that order)
static int s2b_voxel(vilf_handle *h, S2B *c, CSet in, float leaf, CSet out, int cs, int nrows = -1) {
    bool wide;
    int rc = s2b_sort_keys(h, c, in, leaf, cs, nullptr, &wide, nrows);
    if (rc != VILF_OK) return rc;
    const int S = nrows < 0 ? c->S : nrows, ntiles = (in.cap + S2B_VT - 1) / S2B_VT;
    if (!c->tileHeads.ensure((size_t)S * ntiles * 4)) return VILF_ERR_DEVICE;
    if (!wide) {
        hipLaunchKernelGGL(b_voxel_heads<unsigned int>, dim3(ntiles, S), dim3(S2B_VT), 0, h->stream, in, c->keys2.as<unsigned int>(), c->tileHeads.as<int>(), ntiles);
        hipLaunchKernelGGL(b_voxel_reduce<unsigned int>, dim3(ntiles, S), dim3(S2B_VT), 0, h->stream, in, c->keys2.as<unsigned int>(), c->vals2.as<int>(), out, c->tileHeads.as<int>(), ntiles);
    } else {
        hipLaunchKernelGGL(b_voxel_heads<unsigned long long>, dim3(ntiles, S), dim3(S2B_VT), 0, h->stream, in, c->keys2.as<unsigned long long>(), c->tileHeads.as<int>(), ntiles);
        hipLaunchKernelGGL(b_voxel_reduce<unsigned long long>, dim3(ntiles, S), dim3(S2B_VT), 0, h->stream, in, c->keys2.as<unsigned long long>(), c->vals2.as<int>(), out, c->tileHeads.as<int>(), ntiles);
    }
    PROF(0)
    return VILF_OK;
}

// The neighbour directory of local map w for this step. A map in cell-major order (every step leaves it so) is searched in place: one pass writes the directory. A map
// that is not a voxel grid yet (as initialised) is first copied in cell-major order (global sort; the original index travels in w and breaks distance ties).
static int s2b_build_index(vilf_handle *h, S2B *c, int w) {
    CSet map = c->cs_map(w);
    const float leaf = (float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size);
    c->cs_idx[w] = c->cs_cfg[w]; c->idx_copy[w] = false;
    if (c->order_state[w] == 1 && c->dir_ok[w]) return VILF_OK;        // the map update that emitted this map wrote its directory: nothing to do
    const float4 *arr = map.p;
    if (c->order_state[w] != 1) {
        bool wide;
        int rc = s2b_sort_keys(h, c, map, leaf, c->cs_cfg[w], &c->cs_idx[w], &wide);
        if (rc != VILF_OK) return rc;
        hipLaunchKernelGGL(b_gather_sorted, GRID2(map.cap, c->S), 0, h->stream, map, c->vals2.as<int>(), c->sorted[w].as<float4>());
        arr = c->sorted[w].as<float4>(); c->idx_copy[w] = true;
    }
    c->dir_tag[w] = c->next_tag();
    const dim3 dgrid((map.cap + 256 * DIR_PT - 1) / (256 * DIR_PT), c->S);
    hipLaunchKernelGGL(b_dir_build, dgrid, dim3(256), 0, h->stream, arr, map.n, map.cap, 1.0f / leaf, c->cs_idx[w], c->dir_tag[w], c->bstart[w].as<unsigned>());
    c->dir_ok[w] = !c->idx_copy[w];         // (a directory over the sorted copy of an unordered map serves this step only)
    PROF(2)
    return VILF_OK;
}

static int mu_lds_cap(int cap_scan) { int p2 = 2; while (p2 < cap_scan) p2 <<= 1; return std::min(p2, MU_LDS_TAIL); }
// is every stream's local map w in ascending leaf order (a voxel grid of an earlier step)? One check + 4-byte read-back when unknown
// (after an init); a step always leaves the maps ordered, so the steady state never comes here.
static int s2b_resolve_order(vilf_handle *h, S2B *c, int w) {
    if (c->order_state[w] != 0) return VILF_OK;
    const float leaf = (float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size);
    int flag = 0;
    HIPCHECK(h, hipMemsetAsync(c->bits.p, 0, 4, h->stream));
    hipLaunchKernelGGL(b_check_order, dim3(64, c->S), dim3(256), 0, h->stream, c->cs_map(w), 1.0f / leaf, c->cs_cfg[w], c->capScan[w] > mu_lds_cap(c->capScan[w]) ? 16 : 17, c->bits.as<int>());
    HIPCHECK(h, hipMemcpyAsync(&flag, c->bits.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    c->order_state[w] = flag ? 2 : 1;
    return VILF_OK;
}

// one optimation_processing() for every stream; everything is enqueued on the handle's stream, no host round trip
static int s2b_step(vilf_handle *h, S2B *c) {
    const int S = c->S;
    int rc;
    if (c->scan_dirty) {
        for (int w = 0; w < 2; w++) HIPCHECK(h, hipMemcpyAsync(c->nScan[w].p, c->h_nScan[w].data(), (size_t)S * 4, hipMemcpyHostToDevice, h->stream));
        c->scan_dirty = false;
    }
    double *d_pose = c->pose.as<double>();
    S2BRes *d_res = c->res.as<S2BRes>();
    int *d_err = c->err.as<int>();
    h->s2m_groups.clear(); h->s2m_ev.clear();
    PROF(6)
    hipLaunchKernelGGL(b_predict, GRIDS(S), 0, h->stream, d_pose, S);
    PROF(6)
    const float leaf[2] = {(float)h->opts.edge_leaf_size, (float)h->opts.surf_leaf_size};
    for (int w = 0; w < 2; w++) {
        // the host knows every stream's scan size (vilf_scan2map_batch_set_scan): clouds that fit the LDS get the one-workgroup grid (its layout sized for the largest of them),
        // the others — if any — go through the global-sort path as a sub-batch (grid row -> stream map), not the whole batch
        int nmax = 0;
        std::vector<int> big;
        for (int i = 0; i < S; i++) { const int n = c->h_nScan[w][i]; if (n > SV_MAXPTS24) big.push_back(i); else nmax = std::max(nmax, n); }
        if ((int)big.size() < S) {
            int cap = (std::max(nmax, 1) + 15) & ~15;
            // small clouds: the tile staging area (32 KB) gets its own LDS — unless that would not fit (15.1 k .. 16.4 k points: the launch asked for up to 172 KB and failed with
            // "invalid argument"; found by tools/dev_soak_voxel.py): such a batch is laid out for 16384 points, whose index buffer holds the staging area
            if (cap * 2 < 2 * SV_T * 16 && (size_t)cap * 8 + 8192 + (size_t)2 * SV_T * 16 > (size_t)SV_MAXPTS32 * 8 + 8192) cap = SV_T * 16;
            const size_t stage = (cap * 2 >= 2 * SV_T * 16) ? 0 : (size_t)2 * SV_T * 16;
            // the run-sorting grid first (b_scan_voxel_runs); the streams it hands back — clouds without scan order, more than rc runs — are shared out over one workgroup
            // per CU of the point-sorting grid. VILF_SV_NO_RUNS=1: the point-sorting grid for every stream (as before round 4); VILF_SV_RC: run capacity (tests force the hand-back with it)
            static const bool no_runs = std::getenv("VILF_SV_NO_RUNS") != nullptr;
            static const int rc_env = std::getenv("VILF_SV_RC") ? std::atoi(std::getenv("VILF_SV_RC")) : SR_RC;
            const int rc = std::max(64, std::min(SR_RC, rc_env)) & ~1;
            int *d_list = nullptr;
            if (!no_runs) {
                if (!c->svlist[w].ensure(((size_t)S + 1) * 4)) return VILF_ERR_DEVICE;
                d_list = c->svlist[w].as<int>();
                HIPCHECK(h, hipMemsetAsync(d_list, 0, 4, h->stream));
                hipLaunchKernelGGL(b_scan_voxel_runs, dim3(S), dim3(SR_T), sr_lds_bytes(rc), h->stream, c->cs_scan(w), 1.0f / leaf[w], c->cs_ds(w), SV_MAXPTS24, rc, d_list, w == 1 ? 0 : 1);
            }
            const dim3 gsv(d_list ? (unsigned)std::min(S, 128) : (unsigned)S);
            if (cap <= SV_MAXPTS32) hipLaunchKernelGGL(b_scan_voxel<false>, gsv, dim3(SV_T), (size_t)cap * 8 + 8192 + stage, h->stream, c->cs_scan(w), 1.0f / leaf[w], c->cs_ds(w), cap, d_err, d_list);
            else hipLaunchKernelGGL(b_scan_voxel<true>, gsv, dim3(SV_T), (size_t)cap * 7 + 8192 + stage, h->stream, c->cs_scan(w), 1.0f / leaf[w], c->cs_ds(w), cap, d_err, d_list);
            PROF(0)
        }
        if (!big.empty()) {
            if (big != c->h_big[w]) {
                if (!c->bigmap[w].ensure(big.size() * 4)) return VILF_ERR_DEVICE;
                HIPCHECK(h, hipMemcpyAsync(c->bigmap[w].p, big.data(), big.size() * 4, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipStreamSynchronize(h->stream));          // `big` is a temporary
                c->h_big[w] = big;
            }
            CSet in = c->cs_scan(w), out = c->cs_ds(w);
            in.smap = out.smap = c->bigmap[w].as<int>();
            if ((rc = s2b_voxel(h, c, in, leaf[w], out, -1, (int)big.size())) != VILF_OK) return rc;
        }
    }
    hipLaunchKernelGGL(b_gate, GRIDS(S), 0, h->stream, c->nMap[0].as<int>(), c->nMap[1].as<int>(), c->nDs[0].as<int>(), c->nDs[1].as<int>(), d_res, d_err, S);
    PROF(6)
    for (int w = 0; w < 2; w++) {
        if ((rc = s2b_resolve_order(h, c, w)) != VILF_OK) return rc;
        if ((rc = s2b_build_index(h, c, w)) != VILF_OK) return rc;
    }
    const int capq = c->capScan[0] + c->capScan[1];
    AssocArgs aa[2];
    for (int w = 0; w < 2; w++) {
        AssocArgs &a = aa[w];
        a.ds = c->cs_ds(w); a.is_surf = w; a.n_ds_edge = c->nDs[0].as<int>(); a.pose_all = d_pose;
        a.sorted_all = c->idx_copy[w] ? c->sorted[w].as<float4>() : c->map[w].as<float4>(); a.n_map = c->nMap[w].as<int>(); a.cap_map = c->capMap[w];
        a.dir_all = c->bstart[w].as<unsigned>(); a.tag = c->dir_tag[w]; a.inv = 1.0f / leaf[w]; a.cs = c->cs_idx[w]; a.w_is_index = c->idx_copy[w] ? 1 : 0;
        a.res = d_res; a.frec_all = c->frec.as<double>(); a.fkind_all = c->fkind.as<int>(); a.capq = capq; a.tie_count = c->bcnt.as<int>();
    }
    const int nblk_e = (c->capScan[0] + 255) / 256, nblk_s = (c->capScan[1] + 255) / 256;
    for (int pass = 0; pass < h->opts.s2m_outer_iterations && pass < 2; pass++) {
        static const size_t assoc_lds_probe = std::getenv("VILF_ASSOC_LDS") ? (size_t)std::atoi(std::getenv("VILF_ASSOC_LDS")) : 0;     // occupancy experiment: unused dynamic LDS caps the workgroups per CU
        hipLaunchKernelGGL(b_associate, dim3(nblk_e + nblk_s, S), dim3(256), assoc_lds_probe, h->stream, aa[0], aa[1], nblk_e);
        hipLaunchKernelGGL(b_associate_ties, dim3(S), dim3(256), 0, h->stream, aa[0], aa[1]);      // queries that met exactly equal distances (rare), in the reference's order
        PROF(3)
        hipLaunchKernelGGL(b_solve, dim3(S), dim3(S2M_NT), 0, h->stream, d_pose, c->frec.as<double>(), c->fkind.as<int>(), capq, c->nDs[0].as<int>(), c->nDs[1].as<int>(), h->opts.huber_a,
                           h->opts.s2m_max_iterations, pass, d_res);
        PROF(4)
    }
    bool new_dir_ok[2] = {false, false};
    const unsigned new_tag[2] = {c->next_tag(), c->next_tag()};      // tags of the directories the map updates write (for the maps they emit)
    for (int w = 0; w < 2; w++) {     // createSubMap: append registered points, crop, voxel grid
        CSet map = c->cs_map(w), dsw = c->cs_ds(w), tmp = c->cs_tmp(w);
        hipLaunchKernelGGL(b_transform_append, GRID2(dsw.cap, S), 0, h->stream, dsw, d_pose, map, d_err);
        hipLaunchKernelGGL(b_bump, GRIDS(S), 0, h->stream, map, dsw, c->nOld.as<int>(), S);
        PROF(5)
        // the new map goes to the other buffer (the old one stays intact up to its old count: a snapshot taken on it can be
        // restored by swapping back). If that other buffer is where a live snapshot sits, save the snapshot first.
        if (c->has_snapshot && c->snap_live && c->mapAlt[w].p == c->snap_ptr[w]) {
            for (int v = 0; v < 2; v++) {
                if (!c->map0[v].ensure((size_t)S * c->capMap[v] * 16)) return VILF_ERR_DEVICE;
                const bool cur = c->map[v].p == c->snap_ptr[v];
                HIPCHECK(h, hipMemcpyAsync(c->map0[v].p, cur ? c->map[v].p : c->mapAlt[v].p, (size_t)S * c->capMap[v] * 16, hipMemcpyDeviceToDevice, h->stream));
                if (c->snap_dir_ok[v]) {
                    if (!c->dir0[v].ensure((size_t)S * S2B_NBS * 4)) return VILF_ERR_DEVICE;
                    HIPCHECK(h, hipMemcpyAsync(c->dir0[v].p, cur ? c->bstart[v].p : c->bstartAlt[v].p, (size_t)S * S2B_NBS * 4, hipMemcpyDeviceToDevice, h->stream));
                }
            }
            c->snap_live = false;
        }
        if (c->order_state[w] == 1) {  // steady state: one fused pass (crop + merge of the sorted tail + centroids), no host round trip
            // tail-size classes: <= 4096 new points (32 KB of LDS for the sorted tail: two workgroups per CU), <= 8192 (64 KB), larger (global memory).
            // The bucket-sorted copy of this map is dead by now (the next step rebuilds it): its buffer holds the queue of uncommon points.
            const int lds_cap = mu_lds_cap(c->capScan[w]), lds_half = std::min(lds_cap, MU_LDS_TAIL / 2);
            int *gq = c->sorted[w].as<int>();
            const size_t gq_stride = (size_t)c->capMap[w] * 4;
            hipLaunchKernelGGL(b_map_update<false>, dim3(S), dim3(MU_T), ((size_t)lds_half * 8 + (size_t)MU_TILE * 12), h->stream, map, c->nOld.as<int>(), d_pose, h->opts.s2m_crop_half, 1.0f / leaf[w], c->cs_cfg[w], c->cs_mapout(w), c->bstartAlt[w].as<unsigned>(), new_tag[w], c->fixq[w].as<int>(),
                               c->tmpB.as<float4>(), c->capScan[w], (unsigned long long *)nullptr, 0, -1, lds_half, gq, gq_stride, d_err);
            if (lds_cap > lds_half)
                hipLaunchKernelGGL(b_map_update<false>, dim3(S), dim3(MU_T), ((size_t)lds_cap * 8 + (size_t)MU_TILE * 12), h->stream, map, c->nOld.as<int>(), d_pose, h->opts.s2m_crop_half, 1.0f / leaf[w], c->cs_cfg[w], c->cs_mapout(w), c->bstartAlt[w].as<unsigned>(), new_tag[w], c->fixq[w].as<int>(),
                                   c->tmpB.as<float4>(), c->capScan[w], (unsigned long long *)nullptr, 0, lds_half, lds_cap, gq, gq_stride, d_err);
            if (c->capScan[w] > lds_cap) {
                int p2 = 2; while (p2 < c->capScan[w]) p2 <<= 1;
                if (!c->muT.ensure((size_t)S * p2 * 8)) return VILF_ERR_DEVICE;
                hipLaunchKernelGGL(b_map_update<true>, dim3(S), dim3(MU_T), (size_t)MU_TILE * 12, h->stream, map, c->nOld.as<int>(), d_pose, h->opts.s2m_crop_half, 1.0f / leaf[w], c->cs_cfg[w], c->cs_mapout(w), c->bstartAlt[w].as<unsigned>(), new_tag[w], c->fixq[w].as<int>(),
                                   c->tmpB.as<float4>(), c->capScan[w], c->muT.as<unsigned long long>(), p2, 0, lds_cap, gq, gq_stride, d_err);
            }
            PROF(7)
            new_dir_ok[w] = true;      // (of the map the swap below makes current)
        } else {                       // a map that is not a voxel grid yet (as initialised): crop copy + full sort
            hipLaunchKernelGGL(b_crop_compact, dim3(S), dim3(S2B_VT), 0, h->stream, map, d_pose, h->opts.s2m_crop_half, tmp, c->nOld.as<int>(), c->mOld.as<int>());
            PROF(5)
            if ((rc = s2b_voxel(h, c, tmp, leaf[w], c->cs_mapout(w), c->cs_cfg[w])) != VILF_OK) return rc;
        }
        c->order_state[w] = 1;
        std::fill(c->h_cmn[w].begin(), c->h_cmn[w].end(), INT_MAX);
    }
    for (int w = 0; w < 2; w++) {
        std::swap(c->map[w], c->mapAlt[w]); std::swap(c->bstart[w], c->bstartAlt[w]);
        c->dir_ok[w] = new_dir_ok[w]; c->dir_tag[w] = new_tag[w];
    }
    hipLaunchKernelGGL(b_finish, GRIDS(S), 0, h->stream, d_pose, c->nMap[0].as<int>(), c->nMap[1].as<int>(), d_err, d_res, S);
    PROF(6)
    HIPCHECK(h, hipGetLastError());
    if (h->profiling && h->prof_used.size() > 4096) return vilf_prof_flush(h);      // asynchronous steps without a reader: bound the pool
    return VILF_OK;                                                                 // the spans are read by the next call that waits for the stream
}

static void s2b_fill_result(const S2BRes &r, vilf_scan2map_result *res) {
    std::memset(res, 0, sizeof(*res));
    std::memcpy(res->pose_qt, r.pose, 56);
    res->n_edge_ds = r.n_ds[0]; res->n_surf_ds = r.n_ds[1];
    for (int k = 0; k < 2; k++) { res->n_edge_factors[k] = r.nfe[k]; res->n_surf_factors[k] = r.nfs[k]; res->iterations[k] = r.its[k]; res->final_cost[k] = r.cost[k]; }
    res->map_edge_size = r.map_n[0]; res->map_surf_size = r.map_n[1];
    // /Odometry relative pose: q_last^-1 * q, q_last^-1 * (t - t_last) (feature_tracker_node.cpp:392-394)
    const double *pv = r.prev, *q = r.pose;
    const double n2 = pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2] + pv[3] * pv[3];
    const double qi[4] = {-pv[0] / n2, -pv[1] / n2, -pv[2] / n2, pv[3] / n2};
    res->rel_q[0] = qi[3] * q[0] + qi[0] * q[3] + qi[1] * q[2] - qi[2] * q[1];
    res->rel_q[1] = qi[3] * q[1] + qi[1] * q[3] + qi[2] * q[0] - qi[0] * q[2];
    res->rel_q[2] = qi[3] * q[2] + qi[2] * q[3] + qi[0] * q[1] - qi[1] * q[0];
    res->rel_q[3] = qi[3] * q[3] - qi[0] * q[0] - qi[1] * q[1] - qi[2] * q[2];
    const double v[3] = {q[4] - pv[4], q[5] - pv[5], q[6] - pv[6]};
    const double ux = 2 * (qi[1] * v[2] - qi[2] * v[1]), uy = 2 * (qi[2] * v[0] - qi[0] * v[2]), uz = 2 * (qi[0] * v[1] - qi[1] * v[0]);
    res->rel_t[0] = v[0] + qi[3] * ux + (qi[1] * uz - qi[2] * uy);
    res->rel_t[1] = v[1] + qi[3] * uy + (qi[2] * ux - qi[0] * uz);
    res->rel_t[2] = v[2] + qi[3] * uz + (qi[0] * uy - qi[1] * ux);
}
static int s2b_err_to_rc(vilf_handle *h, int err) {
    if (!err) return VILF_OK;
    h->err = std::string("scan2map: ") + ((err & S2B_ERR_MAPCAP) ? "local-map capacity exceeded; " : "") + ((err & S2B_ERR_EXTENT) ? "more than 65535 local-map points in one 1 m column bucket; " : "") +
             ((err & S2B_ERR_VOXEL) ? "voxel index overflow (leaf too small for the cloud extent); " : "") +
             ((err & S2B_ERR_ORDER) ? "local map not in leaf order (internal); " : "");
    return VILF_ERR_UNSUPPORTED;
}
static int s2b_set_cloud(vilf_handle *h, S2B *c, DBuf &buf, int cap, int sid, int offset, const float *xyzi, int n) {
    if (n > 0) HIPCHECK(h, hipMemcpyAsync(buf.as<float4>() + (size_t)sid * cap + offset, xyzi, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    return VILF_OK;
}

// ---- the maps' two orders on the host ---------------------------------------------------------------------------------------------
// Leaf coordinates as the device computes them: single-precision product, floor.
static inline long long h_leaf(float v, float inv) { return (long long)std::floor(v * inv); }
static inline unsigned long long h_pcl_key(const float *q, float inv) {       // z | y | x (21 bits each): pcl::VoxelGrid's order for any min corner
    return ((unsigned long long)((h_leaf(q[2], inv) + (1 << 20)) & 0x1fffff) << 42) | ((unsigned long long)((h_leaf(q[1], inv) + (1 << 20)) & 0x1fffff) << 21) | (unsigned long long)((h_leaf(q[0], inv) + (1 << 20)) & 0x1fffff);
}
// A cloud that IS a voxel grid in PCL's order (strictly ascending leaf index: one point per leaf — e.g. a map that vilf_scan2map_get_map handed out) is uploaded in the
// maps' cell-major order, so that the first step finds an ordered map (b_check_order) and takes the fused update. This changes nothing observable: no two points share a
// leaf, the 5-NN breaks distance ties by the PCL key, and get_map hands the points back in PCL order. Anything else (a raw scan) is uploaded as it is.
static bool s2b_host_cell_major(const float *xyzi, int n, float leaf, int cs, std::vector<float> &out) {
    if (n <= 0) return false;
    const float inv = 1.0f / leaf;
    std::vector<std::pair<unsigned long long, int>> kc(n);
    unsigned long long prev = 0;
    const unsigned long long lm = (1ULL << cs) - 1ULL;
    for (int i = 0; i < n; i++) {
        const float *q = xyzi + 4 * (size_t)i;
        if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) return false;
        const long long lx = h_leaf(q[0], inv), ly = h_leaf(q[1], inv), lz = h_leaf(q[2], inv);
        if (lx < -32768 || lx >= 32767 || ly < -32768 || ly >= 32767 || lz < -32768 || lz >= 32767) return false;      // the narrower of the two ranges of b_map_update
        const unsigned long long kp = h_pcl_key(q, inv);
        if (i > 0 && kp <= prev) return false;
        prev = kp;
        const unsigned long long ix = (unsigned long long)(lx + 65536), iy = (unsigned long long)(ly + 65536), iz = (unsigned long long)(lz + 65536);
        kc[i] = {((iy >> cs) << (34 + cs)) | ((ix >> cs) << (17 + 2 * cs)) | (iz << (2 * cs)) | ((iy & lm) << cs) | (ix & lm), i};
    }
    std::sort(kc.begin(), kc.end());
    out.resize((size_t)n * 4);
    for (int i = 0; i < n; i++) std::memcpy(&out[4 * (size_t)i], xyzi + 4 * (size_t)kc[i].second, 16);
    return true;
}
// the first m points of a downloaded map are in cell-major order: into PCL order (points of one leaf keep their relative order)
static void s2b_host_pcl_order(float *xyzi, int m, float leaf) {
    if (m < 2) return;
    const float inv = 1.0f / leaf;
    std::vector<std::pair<unsigned long long, int>> kp(m);
    for (int i = 0; i < m; i++) kp[i] = {h_pcl_key(xyzi + 4 * (size_t)i, inv), i};
    std::sort(kp.begin(), kp.end());                       // (key, position): stable by construction
    std::vector<float> tmp((size_t)m * 4);
    for (int i = 0; i < m; i++) std::memcpy(&tmp[4 * (size_t)i], xyzi + 4 * (size_t)kp[i].second, 16);
    std::memcpy(xyzi, tmp.data(), (size_t)m * 16);
}

// ---- single-stream ABI (S = 1, capacities grow on demand) ---------------------------------------------------------------
static S2B *single(vilf_handle *h) { if (!h->s2m) h->s2m = new S2B(); return h->s2m; }

extern "C" int vilf_scan2map_init(vilf_handle *h, const float *e, int ne, const float *s, int ns) {
    if (!h || ne < 0 || ns < 0 || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    S2B *c = single(h);
    const int oe = c->S ? c->h_nMap[0][0] : 0, os = c->S ? c->h_nMap[1][0] : 0;
    int rc = s2b_reserve(h, c, 1, ne, ns, oe + ne, os + ns);
    if (rc != VILF_OK) return rc;
    const float *src[2] = {e, s}; const int nn[2] = {ne, ns};                       // localMapInited (:105): map += cloud
    std::vector<float> cm[2];
    for (int w = 0; w < 2; w++) {
        const int old_n = c->h_nMap[w][0];
        const int keep = c->order_state[w] == 1 ? old_n : std::min(c->h_cmn[w][0], old_n);      // what was in cell-major order stays so; the appended cloud is raw ...
        const bool conv = old_n == 0 && s2b_host_cell_major(src[w], nn[w], (float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size), c->cs_cfg[w], cm[w]);   // ... unless it is a whole voxel grid
        if ((rc = s2b_set_cloud(h, c, c->map[w], c->capMap[w], 0, old_n, conv ? cm[w].data() : src[w], nn[w])) != VILF_OK) return rc;
        c->h_cmn[w][0] = conv ? nn[w] : keep;
        c->h_nMap[w][0] += nn[w]; c->order_state[w] = 0; c->dir_ok[w] = false;
        HIPCHECK(h, hipMemcpyAsync(c->nMap[w].p, c->h_nMap[w].data(), 4, hipMemcpyHostToDevice, h->stream));
    }
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}

extern "C" int vilf_scan2map_set_pose(vilf_handle *h, const double p[7], const double pl[7]) {
    if (!h || !p || !pl) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    S2B *c = single(h);
    int rc = s2b_reserve(h, c, 1, c->capScan[0], c->capScan[1], c->capMap[0], c->capMap[1]);
    if (rc != VILF_OK) return rc;
    HIPCHECK(h, vilf_copy_sync(h, c->pose.p, p, 56, hipMemcpyHostToDevice));
    HIPCHECK(h, vilf_copy_sync(h, c->pose.as<double>() + 8, pl, 56, hipMemcpyHostToDevice));
    return VILF_OK;
}

static int s2b_get_map(vilf_handle *h, S2B *c, int sid, int which, float *out, int cap, int *n_out) {
    int n = 0;
    if (c->S) HIPCHECK(h, vilf_copy_sync(h, &n, c->nMap[which].as<int>() + sid, 4, hipMemcpyDeviceToHost));
    *n_out = n;
    const int k = std::min(cap, n);
    if (k > 0 && out) {
        // the map lives in cell-major order on the device (mu_leaf); what the caller sees is pcl::VoxelGrid's order, as the reference's map has it
        const int m = std::min(c->order_state[which] == 1 ? n : c->h_cmn[which][sid], n);
        const float leaf = (float)(which == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size);
        if (m <= k) {
            HIPCHECK(h, vilf_copy_sync(h, out, c->map[which].as<float4>() + (size_t)sid * c->capMap[which], (size_t)k * 16, hipMemcpyDeviceToHost));
            s2b_host_pcl_order(out, m, leaf);
        } else {                                            // a truncated read: order the whole prefix first, then hand out its first k points
            std::vector<float> all((size_t)n * 4);
            HIPCHECK(h, vilf_copy_sync(h, all.data(), c->map[which].as<float4>() + (size_t)sid * c->capMap[which], (size_t)n * 16, hipMemcpyDeviceToHost));
            s2b_host_pcl_order(all.data(), m, leaf);
            std::memcpy(out, all.data(), (size_t)k * 16);
        }
    }
    return VILF_OK;
}
extern "C" int vilf_scan2map_get_map(vilf_handle *h, int which, float *out, int cap, int *n_out) {
    if (!h || !n_out || which < 0 || which > 1) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    return s2b_get_map(h, single(h), 0, which, out, cap, n_out);
}

extern "C" int vilf_scan2map_step(vilf_handle *h, const float *e, int ne, const float *s, int ns, vilf_scan2map_result *res) {
    if (!h || !res || ne < 0 || ns < 0 || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    S2B *c = single(h);
    std::memset(res, 0, sizeof(*res));
    const int oe = c->S ? c->h_nMap[0][0] : 0, os = c->S ? c->h_nMap[1][0] : 0;
    int rc = s2b_reserve(h, c, 1, ne, ns, oe + ne, os + ns);
    if (rc != VILF_OK) return rc;
    const float *src[2] = {e, s}; const int nn[2] = {ne, ns};
    for (int w = 0; w < 2; w++) {
        if ((rc = s2b_set_cloud(h, c, c->scan[w], c->capScan[w], 0, 0, src[w], nn[w])) != VILF_OK) return rc;
        c->h_nScan[w][0] = nn[w];
    }
    c->scan_dirty = true;
    if ((rc = s2b_step(h, c)) != VILF_OK) return rc;
    HIPCHECK(h, hipMemcpyAsync(c->h_res.data(), c->res.p, sizeof(S2BRes), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    const S2BRes &r = c->h_res[0];
    c->h_nMap[0][0] = r.map_n[0]; c->h_nMap[1][0] = r.map_n[1];
    s2b_fill_result(r, res);
    if (r.err) { HIPCHECK(h, hipMemsetAsync(c->err.p, 0, 4, h->stream)); }
    return s2b_err_to_rc(h, r.err);
}

// ---- batched ABI: S independent LiDAR streams with fixed capacities ------------------------------------------------------
extern "C" int vilf_scan2map_batch_create(vilf_handle *h, int n_streams, int cap_scan_edge, int cap_scan_surf, int cap_map_edge, int cap_map_surf) {
    if (!h || n_streams < 1 || n_streams > 65535 || cap_scan_edge < 1 || cap_scan_surf < 1 || cap_map_edge < 1 || cap_map_surf < 1) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    if (h->s2b) { h->s2b->release(); delete h->s2b; h->s2b = nullptr; }
    h->s2b = new S2B();
    return s2b_reserve(h, h->s2b, n_streams, cap_scan_edge, cap_scan_surf, cap_map_edge, cap_map_surf);
}
#define S2B_CHECK(h, sid)                                                                   \
    if (!(h) || !(h)->s2b) return VILF_ERR_INVALID_ARGUMENT;                                \
    S2B *c = (h)->s2b;                                                                      \
    if ((sid) < 0 || (sid) >= c->S) return VILF_ERR_INVALID_ARGUMENT;                       \
    HIPCHECK(h, hipSetDevice((h)->device));

extern "C" int vilf_scan2map_batch_init(vilf_handle *h, int stream, const float *e, int ne, const float *s, int ns, const double *pose_qt, const double *pose_last_qt) {
    S2B_CHECK(h, stream)
    if (ne < 0 || ns < 0 || ne > c->capMap[0] || ns > c->capMap[1] || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    const float *src[2] = {e, s}; const int nn[2] = {ne, ns};
    int rc;
    std::vector<float> cm[2];
    for (int w = 0; w < 2; w++) {                                                   // the stream's local map := cloud
        if (c->order_state[w] == 1) std::fill(c->h_cmn[w].begin(), c->h_cmn[w].end(), INT_MAX);   // the other streams' maps stay in cell-major order
        const bool conv = s2b_host_cell_major(src[w], nn[w], (float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size), c->cs_cfg[w], cm[w]);
        if ((rc = s2b_set_cloud(h, c, c->map[w], c->capMap[w], stream, 0, conv ? cm[w].data() : src[w], nn[w])) != VILF_OK) return rc;
        c->h_cmn[w][stream] = conv ? nn[w] : 0;
        c->h_nMap[w][stream] = nn[w]; c->order_state[w] = 0; c->dir_ok[w] = false;
        HIPCHECK(h, hipMemcpyAsync(c->nMap[w].as<int>() + stream, &c->h_nMap[w][stream], 4, hipMemcpyHostToDevice, h->stream));
    }
    double p[24] = {0};
    p[3] = p[11] = p[19] = 1.0;
    if (pose_qt) for (int k = 0; k < 7; k++) p[k] = p[8 + k] = p[16 + k] = pose_qt[k];
    if (pose_last_qt) for (int k = 0; k < 7; k++) p[8 + k] = pose_last_qt[k];
    HIPCHECK(h, hipMemcpyAsync(c->pose.as<double>() + 24 * (size_t)stream, p, sizeof(p), hipMemcpyHostToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}
// stream dst := stream src (local maps, poses, resident scan), copied on the device: a batch of replicas of a few distinct streams without one upload per stream.
// One launch of the library's own copy kernel per call (thousands of runtime memcpy calls in a row were also what a counter-collecting profiler choked on).
struct CopyJob { const void *src[8]; void *dst[8]; unsigned long long bytes[8]; int n; };
__global__ void b_copy_regions(CopyJob job) {
    const int r = blockIdx.y;
    if (r >= job.n) return;
    const size_t n4 = job.bytes[r] / 4;                           // every region is a whole number of 4-byte words; 16-byte aligned ones go by uint4
    const bool wide = ((size_t)job.src[r] % 16 == 0) && ((size_t)job.dst[r] % 16 == 0) && n4 % 4 == 0;
    if (wide) { const uint4 *s4 = static_cast<const uint4 *>(job.src[r]); uint4 *d4 = static_cast<uint4 *>(job.dst[r]); for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4 / 4; i += (size_t)gridDim.x * blockDim.x) d4[i] = s4[i]; }
    else { const unsigned *s1 = static_cast<const unsigned *>(job.src[r]); unsigned *d1 = static_cast<unsigned *>(job.dst[r]); for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) d1[i] = s1[i]; }
}
extern "C" int vilf_scan2map_batch_copy_stream(vilf_handle *h, int src, int dst) {
    S2B_CHECK(h, src)
    if (dst < 0 || dst >= c->S) return VILF_ERR_INVALID_ARGUMENT;
    if (src == dst) return VILF_OK;
    CopyJob job; job.n = 0;
    auto add = [&](const void *s_, void *d_, size_t bytes) { job.src[job.n] = s_; job.dst[job.n] = d_; job.bytes[job.n] = bytes; job.n++; };
    size_t largest = 0;
    for (int w = 0; w < 2; w++) {
        add(c->map[w].as<float4>() + (size_t)src * c->capMap[w], c->map[w].as<float4>() + (size_t)dst * c->capMap[w], (size_t)c->capMap[w] * 16);
        if (c->dir_ok[w]) add(c->bstart[w].as<unsigned>() + (size_t)src * S2B_NBS, c->bstart[w].as<unsigned>() + (size_t)dst * S2B_NBS, (size_t)S2B_NBS * 4);
        add(c->nMap[w].as<int>() + src, c->nMap[w].as<int>() + dst, 4);
        c->h_nMap[w][dst] = c->h_nMap[w][src]; c->h_nScan[w][dst] = c->h_nScan[w][src]; c->h_cmn[w][dst] = c->h_cmn[w][src];
    }
    hipLaunchKernelGGL(b_copy_regions, dim3(64, job.n), dim3(256), 0, h->stream, job);
    job.n = 0;
    for (int w = 0; w < 2; w++) add(c->scan[w].as<float4>() + (size_t)src * c->capScan[w], c->scan[w].as<float4>() + (size_t)dst * c->capScan[w], (size_t)c->capScan[w] * 16);
    add(c->pose.as<double>() + 24 * (size_t)src, c->pose.as<double>() + 24 * (size_t)dst, 24 * 8);
    hipLaunchKernelGGL(b_copy_regions, dim3(16, job.n), dim3(256), 0, h->stream, job);
    (void)largest;
    HIPCHECK(h, hipGetLastError());
    c->scan_dirty = true;
    return VILF_OK;
}
extern "C" int vilf_scan2map_batch_set_scan(vilf_handle *h, int stream, const float *e, int ne, const float *s, int ns) {
    S2B_CHECK(h, stream)
    if (ne < 0 || ns < 0 || ne > c->capScan[0] || ns > c->capScan[1] || (ne && !e) || (ns && !s)) return VILF_ERR_INVALID_ARGUMENT;
    const float *src[2] = {e, s}; const int nn[2] = {ne, ns};
    int rc;
    for (int w = 0; w < 2; w++) {
        if ((rc = s2b_set_cloud(h, c, c->scan[w], c->capScan[w], stream, 0, src[w], nn[w])) != VILF_OK) return rc;
        c->h_nScan[w][stream] = nn[w];
    }
    c->scan_dirty = true;
    HIPCHECK(h, hipStreamSynchronize(h->stream));          // the caller's buffers may be re-used after return
    return VILF_OK;
}
extern "C" int vilf_scan2map_batch_step(vilf_handle *h, int sync) {
    S2B_CHECK(h, 0)
    int rc = s2b_step(h, c);
    if (rc != VILF_OK) return rc;
    if (sync) { HIPCHECK(h, hipStreamSynchronize(h->stream)); if ((rc = vilf_prof_flush(h)) != VILF_OK) return rc; }
    return VILF_OK;
}
extern "C" int vilf_get_profile_scan2map(vilf_handle *h, double ms_out[8], long launches_out[8]) {
    if (!h || !ms_out || !launches_out) return VILF_ERR_INVALID_ARGUMENT;
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    for (int i = 0; i < 8; i++) { ms_out[i] = h->s2m_ms[i]; launches_out[i] = h->s2m_launches[i]; }
    return VILF_OK;
}
extern "C" int vilf_scan2map_batch_snapshot(vilf_handle *h) {
    S2B_CHECK(h, 0)
    for (int w = 0; w < 2; w++) {                 // the maps are not copied: a step never overwrites its input maps (see s2b_step)
        int rc = s2b_resolve_order(h, c, w);
        if (rc != VILF_OK) return rc;
        c->snap_order[w] = c->order_state[w]; c->snap_cmn[w] = c->h_cmn[w];
        if (c->order_state[w] == 1 && !c->dir_ok[w]) {          // the snapshot carries the maps' directory, as a map that came out of a step does
            const float leaf = (float)(w == 0 ? h->opts.edge_leaf_size : h->opts.surf_leaf_size);
            c->dir_tag[w] = c->next_tag();
            hipLaunchKernelGGL(b_dir_build, dim3((c->capMap[w] + 256 * DIR_PT - 1) / (256 * DIR_PT), c->S), dim3(256), 0, h->stream, c->map[w].as<float4>(), c->nMap[w].as<int>(), c->capMap[w], 1.0f / leaf,
                               c->cs_cfg[w], c->dir_tag[w], c->bstart[w].as<unsigned>());
            c->dir_ok[w] = true;
        }
        c->snap_dir_ok[w] = c->dir_ok[w]; c->snap_dir_tag[w] = c->dir_tag[w];
        if (!c->nMap0[w].ensure((size_t)c->S * 4)) return VILF_ERR_DEVICE;
        HIPCHECK(h, hipMemcpyAsync(c->nMap0[w].p, c->nMap[w].p, (size_t)c->S * 4, hipMemcpyDeviceToDevice, h->stream));
        c->snap_ptr[w] = c->map[w].p;
    }
    if (!c->pose0.ensure((size_t)c->S * 24 * 8)) return VILF_ERR_DEVICE;
    HIPCHECK(h, hipMemcpyAsync(c->pose0.p, c->pose.p, (size_t)c->S * 24 * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    c->has_snapshot = true; c->snap_live = true;
    return VILF_OK;
}
extern "C" int vilf_scan2map_batch_rewind(vilf_handle *h) {
    S2B_CHECK(h, 0)
    if (!c->has_snapshot) { h->err = "scan2map_batch_rewind: no snapshot"; return VILF_ERR_INVALID_ARGUMENT; }
    for (int w = 0; w < 2; w++) {
        if (c->snap_live) { if (c->map[w].p != c->snap_ptr[w]) { std::swap(c->map[w], c->mapAlt[w]); std::swap(c->bstart[w], c->bstartAlt[w]); } }
        else {
            HIPCHECK(h, hipMemcpyAsync(c->map[w].p, c->map0[w].p, (size_t)c->S * c->capMap[w] * 16, hipMemcpyDeviceToDevice, h->stream));
            if (c->snap_dir_ok[w]) HIPCHECK(h, hipMemcpyAsync(c->bstart[w].p, c->dir0[w].p, (size_t)c->S * S2B_NBS * 4, hipMemcpyDeviceToDevice, h->stream));
        }
        c->dir_ok[w] = c->snap_dir_ok[w]; c->dir_tag[w] = c->snap_dir_tag[w];
        HIPCHECK(h, hipMemcpyAsync(c->nMap[w].p, c->nMap0[w].p, (size_t)c->S * 4, hipMemcpyDeviceToDevice, h->stream));
        c->order_state[w] = c->snap_order[w]; c->h_cmn[w] = c->snap_cmn[w];
    }
    HIPCHECK(h, hipMemcpyAsync(c->pose.p, c->pose0.p, (size_t)c->S * 24 * 8, hipMemcpyDeviceToDevice, h->stream));
    HIPCHECK(h, hipMemsetAsync(c->err.p, 0, (size_t)c->S * 4, h->stream));
    return VILF_OK;
}
extern "C" int vilf_scan2map_batch_results(vilf_handle *h, int first, int n, vilf_scan2map_result *out) {
    S2B_CHECK(h, first)
    if (n < 0 || first + n > c->S || (n && !out)) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipMemcpyAsync(c->h_res.data() + first, c->res.as<S2BRes>() + first, (size_t)n * sizeof(S2BRes), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    int err = 0;
    for (int i = 0; i < n; i++) { s2b_fill_result(c->h_res[first + i], out + i); err |= c->h_res[first + i].err; }
    return s2b_err_to_rc(h, err);
}
extern "C" int vilf_scan2map_batch_get_map(vilf_handle *h, int stream, int which, float *out, int cap, int *n_out) {
    S2B_CHECK(h, stream)
    if (!n_out || which < 0 || which > 1) return VILF_ERR_INVALID_ARGUMENT;
    return s2b_get_map(h, c, stream, which, out, cap, n_out);
}

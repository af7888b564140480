// vilf_sort.hip — the library's own stable radix sort of (key, value) pairs in global memory (gfx950).
//
// Used OUTSIDE the steady state only: local maps that arrive unordered (vilf_scan2map_batch_init: pcl::VoxelGrid of the first map in cell-major order, vilf_s2m.hip),
// scan clouds beyond the in-LDS voxel grid's capacity, and the ring sort of the LOAM feature extraction (vilf_feat.hip). Rounds 1-4 called rocprim::radix_sort_pairs
// there; the steady-state sorts were always this library's (LDS radix / bitonic sorts inside b_scan_voxel_runs, b_scan_voxel, b_map_update).
//
// LSD, 8-bit digits, three launches per pass over tiles of 2048 elements:
//   rs_hist      per tile: 256-bin histogram of the pass's digit (LDS integer atomics: counts, order-free) -> hist[digit][tile]
//   rs_scan_*    exclusive prefix of hist in (digit, tile) order = the global position of every (digit, tile) run: chunk sums, scan of the sums, apply
//   rs_scatter   per tile, in index order, 256 elements at a time: rank among the equal digits of the round by ballot matching inside a wave + wave counts in LDS;
//                the round's base per digit carried in LDS. Equal keys keep their input order (stable), so passes compose and the result does not depend on timing.
// Passes ping-pong between the output arrays and scratch so that the LAST pass lands in the output; the input arrays are not written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include "vilf_sort.hpp"

namespace {
constexpr int RS_T = 256, RS_E = 8, RS_TILE = RS_T * RS_E, RS_CH = 4096;

template <typename K>
__global__ __launch_bounds__(RS_T) void rs_hist(const K *keys, size_t n, int shift, unsigned dmask, unsigned *hist, unsigned ntiles) {
    __shared__ unsigned s_h[256];
    const int tid = threadIdx.x;
    s_h[tid] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int j = 0; j < RS_E; j++) {
        const size_t i = base + (size_t)j * RS_T + tid;
        if (i < n) atomicAdd(&s_h[(unsigned)(keys[i] >> shift) & dmask], 1u);
    }
    __syncthreads();
    hist[(size_t)tid * ntiles + blockIdx.x] = s_h[tid];
}
// exclusive scan of m unsigned values, three launches: (a) sums of chunks of RS_CH, (b) exclusive scan of the chunk sums by one workgroup, (c) scan inside the chunks + offset
__global__ __launch_bounds__(256) void rs_scan_sums(const unsigned *v, size_t m, unsigned *sums) {
    __shared__ unsigned s_w[4];
    const size_t base = (size_t)blockIdx.x * RS_CH;
    unsigned a = 0;
    for (int j = 0; j < RS_CH / 256; j++) { const size_t i = base + (size_t)j * 256 + threadIdx.x; if (i < m) a += v[i]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(1024) void rs_scan_top(unsigned *sums, unsigned nch) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (unsigned c0 = 0; c0 < nch; c0 += 1024) {
        const unsigned i = c0 + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const unsigned x = i < nch ? sums[i] : 0u;
        unsigned incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o, 64); if (lane >= (unsigned)o) incl += u; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned off = s_carry;
        for (unsigned k = 0; k < wv; k++) off += s_w[k];
        if (i < nch) sums[i] = off + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = off + incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void rs_scan_apply(unsigned *v, size_t m, const unsigned *sums) {
    __shared__ unsigned s_w[4];
    __shared__ unsigned s_carry;
    const size_t base = (size_t)blockIdx.x * RS_CH;
    if (threadIdx.x == 0) s_carry = sums[blockIdx.x];
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = 0; j < RS_CH / 256; j++) {
        const size_t i = base + (size_t)j * 256 + threadIdx.x;
        const unsigned x = i < m ? v[i] : 0u;
        unsigned incl = x;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o, 64); if (lane >= (unsigned)o) incl += u; }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        unsigned off = s_carry;
        for (unsigned k = 0; k < wv; k++) off += s_w[k];
        if (i < m) v[i] = off + incl - x;
        __syncthreads();
        if (threadIdx.x == 255) s_carry = off + incl;
        __syncthreads();
    }
}
template <typename K>
__global__ __launch_bounds__(RS_T) void rs_scatter(const K *kin, const int *vin, K *kout, int *vout, size_t n, int shift, unsigned dmask, const unsigned *pos, unsigned ntiles) {
    __shared__ unsigned s_base[256];
    __shared__ unsigned short s_cnt[RS_T / 64][256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    s_base[tid] = pos[(size_t)tid * ntiles + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * RS_TILE;
    for (int j = 0; j < RS_E; j++) {
#pragma unroll
        for (int k = 0; k < RS_T / 64; k++) s_cnt[k][tid] = 0;
        __syncthreads();
        const size_t i = base + (size_t)j * RS_T + tid;
        const bool valid = i < n;
        const K key = valid ? kin[i] : (K)0;
        const int val = valid ? vin[i] : 0;
        const unsigned d = (unsigned)(key >> shift) & dmask;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) { const unsigned long long bal = __ballot((d >> b) & 1u); mask &= ((d >> b) & 1u) ? bal : ~bal; }
        const unsigned rank = (unsigned)__popcll(mask & ((1ULL << lane) - 1ULL));
        if (valid && rank == 0) s_cnt[wv][d] = (unsigned short)__popcll(mask);
        __syncthreads();
        if (valid) {
            unsigned off = s_base[d];
            for (int k = 0; k < wv; k++) off += s_cnt[k][d];
            kout[off + rank] = key; vout[off + rank] = val;
        }
        __syncthreads();
        { unsigned t = 0; for (int k = 0; k < RS_T / 64; k++) t += s_cnt[k][tid]; s_base[tid] += t; }
        __syncthreads();
    }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
template <typename K>
int sort_pairs(hipStream_t st, void *temp, size_t temp_bytes, const K *k_in, K *k_out, const int *v_in, int *v_out, size_t n, int bits) {
    if (n == 0) return 0;
    if (!temp || temp_bytes < vilf_sort_temp_bytes(n, sizeof(K))) return -1;
    const unsigned ntiles = (unsigned)((n + RS_TILE - 1) / RS_TILE);
    const size_t m = (size_t)256 * ntiles;
    const unsigned nch = (unsigned)((m + RS_CH - 1) / RS_CH);
    char *p = static_cast<char *>(temp);
    K *k_tmp = reinterpret_cast<K *>(p); p += al256(n * sizeof(K));
    int *v_tmp = reinterpret_cast<int *>(p); p += al256(n * 4);
    unsigned *hist = reinterpret_cast<unsigned *>(p); p += al256(m * 4);
    unsigned *sums = reinterpret_cast<unsigned *>(p);
    const int npass = bits <= 0 ? 1 : (bits + 7) / 8;
    const K *ks = k_in; const int *vs = v_in;
    for (int pass = 0; pass < npass; pass++) {
        const bool to_out = ((npass - 1 - pass) & 1) == 0;            // the last pass writes the output arrays
        K *kd = to_out ? k_out : k_tmp; int *vd = to_out ? v_out : v_tmp;
        const int shift = 8 * pass;
        const unsigned dmask = (bits - shift >= 8 || bits <= 0) ? 255u : ((1u << (bits - shift)) - 1u);      // only the low `bits` bits take part
        hipLaunchKernelGGL(rs_hist<K>, dim3(ntiles), dim3(RS_T), 0, st, ks, n, shift, dmask, hist, ntiles);
        hipLaunchKernelGGL(rs_scan_sums, dim3(nch), dim3(256), 0, st, hist, m, sums);
        hipLaunchKernelGGL(rs_scan_top, dim3(1), dim3(1024), 0, st, sums, nch);
        hipLaunchKernelGGL(rs_scan_apply, dim3(nch), dim3(256), 0, st, hist, m, sums);
        hipLaunchKernelGGL(rs_scatter<K>, dim3(ntiles), dim3(RS_T), 0, st, ks, vs, kd, vd, n, shift, dmask, hist, ntiles);
        ks = kd; vs = vd;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
}  // namespace

size_t vilf_sort_temp_bytes(size_t n, size_t key_bytes) {
    const size_t ntiles = (n + RS_TILE - 1) / RS_TILE, m = 256 * std::max<size_t>(ntiles, 1), nch = (m + RS_CH - 1) / RS_CH;
    return al256(n * key_bytes) + al256(n * 4) + al256(m * 4) + al256(nch * 4) + 1024;
}
int vilf_sort_pairs_u32(hipStream_t st, void *temp, size_t temp_bytes, const unsigned *k_in, unsigned *k_out, const int *v_in, int *v_out, size_t n, int bits) {
    return sort_pairs<unsigned>(st, temp, temp_bytes, k_in, k_out, v_in, v_out, n, bits);
}
int vilf_sort_pairs_u64(hipStream_t st, void *temp, size_t temp_bytes, const unsigned long long *k_in, unsigned long long *k_out, const int *v_in, int *v_out, size_t n, int bits) {
    return sort_pairs<unsigned long long>(st, temp, temp_bytes, k_in, k_out, v_in, v_out, n, bits);
}

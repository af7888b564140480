// vilf_api.hip — host side of the C ABI (include/vilfusion.h): packing of window snapshots into the HBM batch layout
// (vilf_batch.hpp), kernel launches on the handle's HIP stream, download. No CPU compute fallback: without a GPU
// vilf_create() fails with VILF_ERR_NO_GPU.
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <atomic>
#include "vilf_internal.hpp"
#include "vilf_sort.hpp"

#define IMU_REC 288
#define IMU_SQRT 62

extern "C" {
__global__ void k_imu_prep(int n, const double *cov, double *work, double *imu_rec);
__global__ void k_prior_prep(VbBatch b, double *prior_H, double *prior_g, unsigned lds_bytes, const int *done4);
#define VILF_PRIOR_PREP_LDS ((size_t)(MG_NK + 1) * (MG_NK + 1) * sizeof(double))     // the n x n prior Jacobian in LDS (n <= 96: 73.5 KB, two workgroups per CU)
__global__ void k_linearize(VbBatch b, int iteration_zero);
__global__ void k_linearize_last(VbBatch b);
__global__ void k_solve(VbBatch b);
__global__ void k_solve_sb(VbBatch b);
__global__ void k_finalize(VbBatch b);
__global__ void k_reset(VbBatch b, int rewind_state);
__global__ void k_marg_prepare(VbBatch b, VbMarg g);
__global__ void k_marg_prepare_td(VbBatch b, VbMarg g);
__global__ void k_prior_keep(VbBatch b, VbMarg g);
__global__ void k_marg_schur(VbBatch b, VbMarg g, int exact);
__global__ void k_marg_finish(VbBatch b, VbMarg g, int n_lo, int n_hi, int only_flagged);
__global__ void k_mf_tridiag(VbBatch b, VbMarg g, int n_lo, int n_hi);
__global__ void k_mf_chol(VbBatch b, VbMarg g, int n_lo, int n_hi, int disable);
__global__ void k_mf_chol_tiles(VbBatch b, VbMarg g, int disable);
__global__ void k_linearize_split(VbBatch b, int iteration_zero);
__global__ void k_iter(VbBatch b, int iteration_zero, unsigned *slot_bm, int *err);
__global__ void k_sb_table(int *tab);
__global__ void k_mf_ql(VbBatch b, VbMarg g, int force_overflow);
__global__ void k_mf_apply(VbBatch b, VbMarg g, int n_lo, int n_hi);
#define VILF_MFA_LDS_EXTRA (4 * 64 * 8 + QL_ICAP * 2)      // k_mf_apply behind V: two staged chunks of the rotation log (MFA_CH = 64) + the 16-bit QL iteration table
__global__ void k_hook_projection(const double *, const double *, const double *, double, const double *, const double *, double, double *);
__global__ void k_hook_projection_td(const double *in, double *out);
__global__ void k_time_limit(VbBatch b, const int *mflag, int only_margin_old);
__global__ void k_hook_imu(const double *, const double *, const double *, const double *, const double *, const double *, double *, double *);
__global__ void k_hook_lidar(const double *, const double *, const double *, const double *, const double *, double *);
__global__ void k_hook_edge(const double *, const double *, const double *, const double *, double *);
__global__ void k_hook_surf(const double *, const double *, const double *, double, double *);
__global__ void k_hook_plus(const double *, const double *, int, double *);
}

// the batch descriptor's view of the live prior set (the two sets swap: vilf_batch_marginalize / vilf_batch_rewind)
static void bind_prior_pointers(vilf_handle *h) {
    VbBatch &b = h->batch;
    b.prior_hdr = h->d[D_PHDR].as<int>(); b.prior_x0 = h->d[D_PX0].as<double>(); b.prior_J = h->d[D_PJ].as<double>(); b.prior_r = h->d[D_PR].as<double>();
    b.prior_H = h->d[D_PH].as<double>(); b.prior_g = h->d[D_PG].as<double>();
}

namespace {

// Frame pairs -> the four wave classes of k_linearize (longest-processing-time: pairs by falling factor count onto the lightest class, ties by pair index:
// deterministic), start of every pair inside its class list, and the number of factor slots (whole chunks) the window needs.
int class_plan(const vilf_window_in &in, int *cls, int *cstart, int *ccount) {
    int pcount[VB_NPAIR] = {0};
    for (int f = 0; f < in.n_features; f++) {
        const int s = in.feature_start_frame[f], n = in.feature_obs_offset[f + 1] - in.feature_obs_offset[f];
        for (int k = 1; k < n; k++) { const int j = s + k; pcount[j * (j - 1) / 2 + s]++; }
    }
    int order[VB_NPAIR], load[4] = {0, 0, 0, 0};
    for (int p = 0; p < VB_NPAIR; p++) order[p] = p;
    std::stable_sort(order, order + VB_NPAIR, [&](int a, int b) { return pcount[a] > pcount[b]; });
    for (int k = 0; k < VB_NPAIR; k++) { const int p = order[k]; int best = 0; for (int c = 1; c < 4; c++) if (load[c] < load[best]) best = c; cls[p] = best; load[best] += pcount[p]; }
    int pos[4] = {0, 0, 0, 0};
    for (int p = 0; p < VB_NPAIR; p++) { cstart[p] = pos[cls[p]]; ccount[p] = pcount[p]; pos[cls[p]] += pcount[p]; }
    const int longest = std::max(std::max(pos[0], pos[1]), std::max(pos[2], pos[3]));
    return VB_CHUNK * std::max(1, (longest + VB_CLS - 1) / VB_CLS);
}

// typed spans out of one (pinned) staging allocation, 64-byte aligned
struct Carve { char *base; size_t off = 0; template <typename T> T *take(size_t n) { off = (off + 63) & ~(size_t)63; T *r = reinterpret_cast<T *>(base + off); off += n * sizeof(T); return r; } };

void quat_from_R(const double *m, double *q /*xyzw*/) {   // Eigen Quaterniond(Matrix3d)
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t; q[1] = (m[2] - m[6]) * t; q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[4 * i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (m[3 * k + j] - m[3 * j + k]) * t;
        q[j] = (m[3 * j + i] + m[3 * i + j]) * t;
        q[k] = (m[3 * k + i] + m[3 * i + k]) * t;
    }
}
void quat_to_R(const double *q, double *R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

}  // namespace

// Small batches (the reference's real-time use is ONE window per frame): the host image of an upload goes to the device in ONE copy and a kernel hands its spans
// out to the library's arrays — the ~25 separate copies of the batched path cost 13 us each (4.5 busy + the gap to the next) before the first kernel of a solve
// could start: 0.43 of a single window's 1.7 ms. The same backwards for the results.
struct UpJob { const char *src; char *dst; unsigned long long bytes; };
struct UpJobs { int n; UpJob j[40]; };
__global__ __launch_bounds__(256) void k_copy_spans(UpJobs jobs) {
    const UpJob jb = jobs.j[blockIdx.y];
    const unsigned long long stride = (unsigned long long)gridDim.x * 256, i0 = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if ((((unsigned long long)jb.src | (unsigned long long)jb.dst) & 15ull) == 0) {          // the usual case: 16-byte aligned on both sides
        const unsigned long long n16 = jb.bytes >> 4;
        const uint4 *s4 = reinterpret_cast<const uint4 *>(jb.src); uint4 *d4 = reinterpret_cast<uint4 *>(jb.dst);
        for (unsigned long long i = i0; i < n16; i += stride) d4[i] = s4[i];
        if (blockIdx.x == 0) for (unsigned long long i = (n16 << 4) + threadIdx.x; i < jb.bytes; i += 256) jb.dst[i] = jb.src[i];
    } else if ((((unsigned long long)jb.src | (unsigned long long)jb.dst) & 7ull) == 0) {    // rows of an odd number of doubles, not the first window
        const unsigned long long n8 = jb.bytes >> 3;
        const unsigned long long *s8 = reinterpret_cast<const unsigned long long *>(jb.src); unsigned long long *d8 = reinterpret_cast<unsigned long long *>(jb.dst);
        for (unsigned long long i = i0; i < n8; i += stride) d8[i] = s8[i];
        if (blockIdx.x == 0) for (unsigned long long i = (n8 << 3) + threadIdx.x; i < jb.bytes; i += 256) jb.dst[i] = jb.src[i];
    } else for (unsigned long long i = i0; i < jb.bytes; i += stride) jb.dst[i] = jb.src[i];
}

extern "C" const char *vilf_version(void) { return "vilfusion-hip 0.1 (gfx950)"; }

extern "C" void vilf_default_options(vilf_options *o) {
    std::memset(o, 0, sizeof(*o));
    o->window_size = 10;
    o->max_num_iterations = 8;
    o->max_solver_time = -1.0;
    o->focal_length = 460.0;
    o->cauchy_a = 1.0;
    o->G[2] = 9.81007;
    o->estimate_extrinsic = 0; o->estimate_td = 0; o->use_lidar_const = 1;
    // config/kitti/kitti_config.yaml:10-23,48-61; rotation matrices re-orthonormalised through a quaternion as in
    // vins_estimator/parameters.cpp:108-110,123-125
    const double ric[9] = {0.00781297, -0.0042792, 0.99996, -0.999859, -0.014868, 0.00774856, 0.0148343, -0.99988, -0.00439476};
    const double tic[3] = {1.1439, -0.312718, 0.726546};
    const double rcl[9] = {7.027555e-03, -9.999753e-01, 2.599616e-05, -2.254837e-03, -4.184312e-05, -9.999975e-01, 9.999728e-01, 7.027479e-03, -2.255075e-03};
    const double tcl[3] = {-7.137748e-03, -7.482656e-02, -3.336324e-01};
    double q[4], n;
    quat_from_R(ric, q); n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (double &v : q) v /= n; quat_to_R(q, o->RIC);
    quat_from_R(rcl, q); n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]); for (double &v : q) v /= n; quat_to_R(q, o->RCL);
    for (int i = 0; i < 3; i++) { o->TIC[i] = tic[i]; o->TCL[i] = tcl[i]; }
    o->TR = 0; o->ROW = 370; o->init_depth = 5.0;
    o->edge_leaf_size = 0.4; o->surf_leaf_size = 0.8; o->huber_a = 0.1;
    o->s2m_outer_iterations = 2; o->s2m_max_iterations = 4; o->s2m_crop_half = 100.0;
}

extern "C" int vilf_create(const vilf_options *opts, int device, void *hip_stream, vilf_handle **out) {
    if (!opts || !out) return VILF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VILF_ERR_NO_GPU;
    if (device < 0 || device >= ndev) return VILF_ERR_INVALID_ARGUMENT;
    if (opts->window_size < 1 || opts->window_size > 4096) return VILF_ERR_UNSUPPORTED;   // 10 = the reference's compile-time WINDOW_SIZE (batched LDS kernels); other sizes: single-window general path (vilf_lw.hip)
    vilf_handle *h = new vilf_handle();
    h->opts = *opts;
    h->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete h; return VILF_ERR_DEVICE; }
    if (hip_stream) { h->stream = (hipStream_t)hip_stream; h->own_stream = false; }
    // the library's own stream does not synchronise with the legacy default stream: two handles of one process (the estimator and the LiDAR stage, as the reference's
    // separate nodes) run side by side — with a blocking stream every default-stream operation of the process (a torch tensor op, a hipMemcpy) serialised them
    else { if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return VILF_ERR_DEVICE; } h->own_stream = true; }
    hipEventCreate(&h->ev0); hipEventCreate(&h->ev1);
    h->solve_lds = (size_t)(66 * 256 + 6 * VB_NPAD + 512 + VILF_MAX_FEATURES) * sizeof(double) + (size_t)VILF_MAX_FEATURES * sizeof(int);
    h->lin_lds = (size_t)VB_LIN_LDS_BYTES;                        // factor chunk + the tables behind it (all dynamic: k_iter overlays the solve's plan on it)
    { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ncu <= 0) ncu = 256; h->iter_slots = 2 * ncu; }
    h->solve_sb_lds = (size_t)SB_LDS_DOUBLES * sizeof(double);
    // occupancy experiment (tools/dev_window_occ.sh): unused extra dynamic LDS leaves ONE workgroup per CU instead of two
    if (const char *e = std::getenv("VILF_LIN_LDS_EXTRA")) h->lin_lds += (size_t)std::atoi(e);
    if (const char *e = std::getenv("VILF_SB_LDS_EXTRA")) h->solve_sb_lds += (size_t)std::atoi(e);
    static_assert(VB_LIN_LDS_DOUBLES >= 10 * 512, "IMU staging area");
    h->marg_lds_schur = (size_t)MG_MLDS * MG_MLDS * sizeof(double);
    h->marg_lds_finish = (size_t)(MG_NK + 2) * (MG_NK + 2) * sizeof(double);
    if (hipFuncSetAttribute((const void *)k_marg_schur, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->marg_lds_schur) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_marg_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->marg_lds_finish) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_prior_prep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)VILF_PRIOR_PREP_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_mf_tridiag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->marg_lds_finish) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_mf_chol, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->marg_lds_finish) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_mf_apply, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(h->marg_lds_finish + VILF_MFA_LDS_EXTRA)) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_mf_ql, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (MG_NK + 2) * QL_LPW * sizeof(double))) != hipSuccess) { delete h; return VILF_ERR_DEVICE; }
    if (hipFuncSetAttribute((const void *)k_linearize, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lin_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_linearize_last, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lin_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_linearize_split, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lin_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_iter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(h->lin_lds, h->solve_sb_lds)) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_solve_sb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->solve_sb_lds) != hipSuccess) {
        delete h; return VILF_ERR_DEVICE;
    }
    std::memset(&h->batch, 0, sizeof(h->batch));
    // k_solve_sb's gather index table: decoded once on the device (the same arithmetic the kernel used to run per launch), read by every solve
    if (!h->d[D_SBTAB].ensure((size_t)23 * 256 * 16)) { delete h; return VILF_ERR_DEVICE; }
    hipLaunchKernelGGL(k_sb_table, dim3(1), dim3(256), 0, h->stream, h->d[D_SBTAB].as<int>());
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) { delete h; return VILF_ERR_DEVICE; }
    *out = h;
    return VILF_OK;
}

extern "C" void vilf_destroy(vilf_handle *h) {
    if (!h) return;
    hipSetDevice(h->device);
    hipStreamSynchronize(h->stream);
    vilf_s2m_release(h);
    vilf_feat_release(h);
    vilf_pg_release(h);
    vilf_lw_release(h);
    for (auto &b : h->d) b.release();
    h->pin_up.release(); h->pin_down.release();
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (hipEvent_t e : h->prof_used) hipEventDestroy(e);
    for (hipEvent_t e : h->prof_free) hipEventDestroy(e);
    if (h->wait_ev) hipEventDestroy(h->wait_ev);
    if (h->stamp_ev) hipEventDestroy(h->stamp_ev);
    if (h->split_ev) hipEventDestroy(h->split_ev);
    for (hipStream_t st : h->split_streams) if (st) hipStreamDestroy(st);
    if (h->stamp_pinned) (void)hipHostFree(h->stamp_pinned);
    h->s2m_ev.clear(); h->prof_used.clear(); h->prof_free.clear(); h->prof_pending.clear();
    if (h->own_stream) hipStreamDestroy(h->stream);
    delete h;
}

// ---- deferred profile spans (see vilf_handle::prof_pending)
hipEvent_t vilf_prof_event(vilf_handle *h) {
    hipEvent_t e = nullptr;
    if (!h->prof_free.empty()) { e = h->prof_free.back(); h->prof_free.pop_back(); } else hipEventCreate(&e);
    h->prof_used.push_back(e);
    hipEventRecord(e, h->stream);
    return e;
}
void vilf_prof_span(vilf_handle *h, hipEvent_t a, hipEvent_t b, double *ms, long *cnt) { h->prof_pending.push_back(vilf_handle::ProfSpan{a, b, ms, cnt}); }
int vilf_prof_flush(vilf_handle *h) {
    if (h->prof_used.empty()) return VILF_OK;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (const vilf_handle::ProfSpan &p : h->prof_pending) { float t = 0; hipEventElapsedTime(&t, p.a, p.b); *p.ms += t; *p.cnt += 1; }
    h->prof_pending.clear();
    for (hipEvent_t e : h->prof_used) h->prof_free.push_back(e);
    h->prof_used.clear();
    return VILF_OK;
}
// The work enqueued on h's stream from now on starts after everything enqueued on other's stream so far has finished: a dependency on the device, the host
// does not wait. (The LiDAR stage and the window solve of a frame run on two handles: this orders them without a host round trip between them.)
extern "C" int vilf_set_async_upload(vilf_handle *h, int on) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    h->async_upload = on != 0;
    return VILF_OK;
}
extern "C" int vilf_wait_for(vilf_handle *h, vilf_handle *other) {
    if (!h || !other) return VILF_ERR_INVALID_ARGUMENT;
    if (h->device != other->device) { h->err = "vilf_wait_for: the handles are on different devices"; return VILF_ERR_INVALID_ARGUMENT; }
    if (h == other || h->stream == other->stream) return VILF_OK;
    HIPCHECK(h, hipSetDevice(h->device));
    if (!h->wait_ev) HIPCHECK(h, hipEventCreateWithFlags(&h->wait_ev, hipEventDisableTiming));
    HIPCHECK(h, hipEventRecord(h->wait_ev, other->stream));
    HIPCHECK(h, hipStreamWaitEvent(h->stream, h->wait_ev, 0));
    return VILF_OK;
}

extern "C" const char *vilf_last_error(const vilf_handle *h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int vilf_reset(vilf_handle *h) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    for (auto &p : h->priors) p.valid = 0;
    std::fill(h->prior_dirty.begin(), h->prior_dirty.end(), 1);
    std::fill(h->prior_dev_newer.begin(), h->prior_dev_newer.end(), 0);
    return VILF_OK;
}

extern "C" int vilf_synchronize(vilf_handle *h) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }      // pending profile spans: the stream is idle, reading them costs nothing
    return VILF_OK;
}

static int pull_device_priors(vilf_handle *h) {
    if (!h->resident) return VILF_OK;
    for (int w = 0; w < h->B && w < (int)h->prior_dev_newer.size(); w++) {
        if (!h->prior_dev_newer[w]) continue;
        vilf_prior &p = h->priors[w];
        std::memset(&p, 0, sizeof(p));
        int hdr[VB_PRIOR_HDR];
        HIPCHECK(h, vilf_copy_sync(h, hdr, h->d[D_PHDR].as<int>() + (size_t)w * VB_PRIOR_HDR, sizeof(hdr), hipMemcpyDeviceToHost));
        p.valid = hdr[0]; p.n = hdr[1]; p.n_blocks = hdr[2]; p.m = hdr[75];
        if (p.valid) {
            std::vector<double> x0(24 * 9);
            HIPCHECK(h, vilf_copy_sync(h, x0.data(), h->d[D_PX0].as<double>() + (size_t)w * 24 * 9, x0.size() * 8, hipMemcpyDeviceToHost));
            for (int i = 0; i < p.n_blocks; i++) { p.block_id[i] = hdr[3 + i]; p.block_size[i] = hdr[27 + i]; p.block_idx[i] = hdr[51 + i]; for (int k = 0; k < 9; k++) p.block_x0[i][k] = x0[i * 9 + k]; }
            HIPCHECK(h, vilf_copy_sync(h, p.linearized_jacobians, h->d[D_PJ].as<double>() + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, sizeof(double) * p.n * p.n, hipMemcpyDeviceToHost));
            HIPCHECK(h, vilf_copy_sync(h, p.linearized_residuals, h->d[D_PR].as<double>() + (size_t)w * VB_PRIOR_LD, sizeof(double) * p.n, hipMemcpyDeviceToHost));
        }
        h->prior_dev_newer[w] = 0;
    }
    return VILF_OK;
}

// host -> device for the slots whose host mirror changed (vilf_prior_import, first upload). Slots whose prior was produced on the
// device (marginalization) or is unchanged are left alone: in the running system the prior never crosses PCIe.
static int upload_priors(vilf_handle *h) {
    const int B = h->B;
    std::vector<int> dirty;
    for (int w = 0; w < B; w++) if (h->prior_dirty[w]) dirty.push_back(w);
    if (dirty.empty()) return VILF_OK;
    h->prior_backup_valid = false; h->prior_restore_needed = false;
    for (int w : dirty) {
        const vilf_prior &p = h->priors[w];
        h->prior_dev_newer[w] = 0;
        if (p.valid) for (int i = 0; i < p.n_blocks; i++) if (p.block_id[i] > 2 * VB_NF + (h->opts.estimate_td ? 1 : 0)) { h->err = "prior touches feature blocks (or Td without estimate_td): unsupported"; return VILF_ERR_UNSUPPORTED; }
        char dense_w = 0;
        if (p.valid) for (int i = 0; i < p.n_blocks; i++) if (p.block_id[i] > VB_NF && p.block_id[i] < 2 * VB_NF) dense_w = 1;
        if ((int)h->prior_dense.size() <= w) h->prior_dense.resize(w + 1, 0);
        h->prior_dense_count += dense_w - h->prior_dense[w]; h->prior_dense[w] = dense_w;
    }
    h->solve_dense_fallback = h->prior_dense_count > 0;      // recomputed with every upload: a later prior without such a block returns the handle to k_solve_sb
    auto fill = [&](const vilf_prior &p, int *hd, double *x0) {
        std::memset(hd, 0, VB_PRIOR_HDR * sizeof(int)); std::memset(x0, 0, 24 * 9 * sizeof(double));
        if (!p.valid) return;
        hd[0] = 1; hd[1] = p.n; hd[2] = p.n_blocks; hd[75] = p.m;
        for (int i = 0; i < p.n_blocks; i++) {
            hd[3 + i] = p.block_id[i]; hd[27 + i] = p.block_size[i]; hd[51 + i] = p.block_idx[i];
            for (int k = 0; k < 9; k++) x0[i * 9 + k] = p.block_x0[i][k];
        }
    };
    if ((int)dirty.size() * 4 > B) {          // most slots: one bulk copy per array
        std::vector<int> hdr((size_t)B * VB_PRIOR_HDR, 0);
        std::vector<double> x0((size_t)B * 24 * 9, 0.0), J((size_t)B * VB_PRIOR_LD * VB_PRIOR_LD, 0.0), r((size_t)B * VB_PRIOR_LD, 0.0);
        if ((int)dirty.size() < B) {           // keep what the other slots hold on the device
            { int rcp = pull_device_priors(h); if (rcp != VILF_OK) return rcp; }
            for (int w = 0; w < B; w++) h->prior_dirty[w] = 1;
        }
        for (int w = 0; w < B; w++) {
            const vilf_prior &p = h->priors[w];
            fill(p, &hdr[(size_t)w * VB_PRIOR_HDR], &x0[(size_t)w * 24 * 9]);
            if (!p.valid) continue;
            std::memcpy(&J[(size_t)w * VB_PRIOR_LD * VB_PRIOR_LD], p.linearized_jacobians, sizeof(double) * p.n * p.n);
            std::memcpy(&r[(size_t)w * VB_PRIOR_LD], p.linearized_residuals, sizeof(double) * p.n);
        }
        HIPCHECK(h, hipMemcpyAsync(h->d[D_PHDR].p, hdr.data(), hdr.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipMemcpyAsync(h->d[D_PX0].p, x0.data(), x0.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipMemcpyAsync(h->d[D_PJ].p, J.data(), J.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipMemcpyAsync(h->d[D_PR].p, r.data(), r.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));   // host vectors go out of scope
    } else {                                   // a few slots: per-slot copies
        for (int w : dirty) {
            const vilf_prior &p = h->priors[w];
            int hd[VB_PRIOR_HDR]; double x0[24 * 9];
            fill(p, hd, x0);
            HIPCHECK(h, hipMemcpyAsync(h->d[D_PHDR].as<int>() + (size_t)w * VB_PRIOR_HDR, hd, sizeof(hd), hipMemcpyHostToDevice, h->stream));
            HIPCHECK(h, hipMemcpyAsync(h->d[D_PX0].as<double>() + (size_t)w * 24 * 9, x0, sizeof(x0), hipMemcpyHostToDevice, h->stream));
            if (p.valid) {
                HIPCHECK(h, hipMemcpyAsync(h->d[D_PJ].as<double>() + (size_t)w * VB_PRIOR_LD * VB_PRIOR_LD, p.linearized_jacobians, sizeof(double) * p.n * p.n, hipMemcpyHostToDevice, h->stream));
                HIPCHECK(h, hipMemcpyAsync(h->d[D_PR].as<double>() + (size_t)w * VB_PRIOR_LD, p.linearized_residuals, sizeof(double) * p.n, hipMemcpyHostToDevice, h->stream));
            }
            HIPCHECK(h, hipStreamSynchronize(h->stream));   // hd / x0 are stack buffers
        }
    }
    hipLaunchKernelGGL(k_prior_prep, dim3(B), dim3(VB_NT), VILF_PRIOR_PREP_LDS, h->stream, h->batch, h->d[D_PH].as<double>(), h->d[D_PG].as<double>(), (unsigned)VILF_PRIOR_PREP_LDS, (const int *)nullptr);
    HIPCHECK(h, hipGetLastError());
    for (int w = 0; w < B; w++) h->prior_dirty[w] = 0;
    h->prior_slots_valid = std::max(h->prior_slots_valid, B);
    return VILF_OK;
}

extern "C" int vilf_batch_upload(vilf_handle *h, int B, const vilf_window_in *wins) {
    if (!h || B <= 0 || !wins) return VILF_ERR_INVALID_ARGUMENT;
    const bool timing = std::getenv("VILF_DEBUG_TIMING") != nullptr;
    auto tnow = []() { return std::chrono::steady_clock::now(); };
    auto t_a = tnow();
    auto lap = [&](const char *what) { if (timing) { auto t = tnow(); fprintf(stderr, "[vilf_batch_upload] %-28s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_a).count()); t_a = t; } };
    HIPCHECK(h, hipSetDevice(h->device));
    if (h->upload_inflight) { HIPCHECK(h, hipStreamSynchronize(h->stream)); h->upload_inflight = false; }      // vilf_set_async_upload: the last upload's copies may still read the staging
    // the device-resident priors of slots 0..B-1 survive this call unless the slot range grows (buffers may be re-allocated)
    const bool keep_priors = h->resident && B <= h->prior_slots_valid;
    if (!keep_priors) { int rcp = pull_device_priors(h); if (rcp != VILF_OK) return rcp; }
    // validation, the class plan of every window (kept for the packer) and the batch-wide maxima, on the host threads: one thread took 4.3 of the 16.5 ms
    // a 2048-window upload cost
    int Fmax = 4, Omax = 4, FACmax = 4, Mcap = 2;
    const int nthr = B >= 64 ? (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency())) : 1;
    h->plans.resize(B);
    {
        struct Local { int Fmax = 4, Omax = 4, FACmax = 4, Mcap = 2, rc = VILF_OK; const char *err = nullptr; };
        std::vector<Local> loc(nthr);
        auto check = [&](int t) {
            Local &L = loc[t];
            auto fail = [&](const char *msg, int rc) { if (L.rc == VILF_OK) { L.rc = rc; L.err = msg; } };
            for (int w = t; w < B; w += nthr) {
                const vilf_window_in &in = wins[w];
                if (in.n_frames != VB_NF) { fail("n_frames must be window_size + 1 = 11", VILF_ERR_INVALID_ARGUMENT); continue; }
                if (in.n_features < 0 || in.n_features > VILF_MAX_FEATURES) { fail("n_features out of range", VILF_ERR_INVALID_ARGUMENT); continue; }
                if (!in.para_pose || !in.para_speed_bias || !in.imu || (in.n_features && (!in.para_feature || !in.feature_const || !in.feature_start_frame || !in.feature_obs_offset || !in.obs_point))) {
                    fail("null input array", VILF_ERR_INVALID_ARGUMENT); continue;
                }
                if (h->opts.use_lidar_const && !in.lidar) { fail("lidar constraints missing (use_lidar_const = 1)", VILF_ERR_INVALID_ARGUMENT); continue; }
                if (h->opts.estimate_td && in.n_features && (!in.obs_velocity || !in.obs_cur_td || !in.obs_row)) { fail("estimate_td needs obs_velocity / obs_cur_td / obs_row", VILF_ERR_INVALID_ARGUMENT); continue; }
                // the observation CSR must be exactly [0 .. n_obs): the packer indexes obs_point / the factor arrays through it
                if (in.n_obs < 0 || (in.n_features && (in.feature_obs_offset[0] != 0 || in.feature_obs_offset[in.n_features] != in.n_obs)) || (!in.n_features && in.n_obs != 0)) {
                    fail("feature_obs_offset must start at 0 and end at n_obs", VILF_ERR_INVALID_ARGUMENT); continue;
                }
                bool ok = true;
                int mf = 0;
                for (int f = 0; f < in.n_features; f++) {
                    const int s = in.feature_start_frame[f], n = in.feature_obs_offset[f + 1] - in.feature_obs_offset[f];     // n >= 2 also makes the offsets increasing
                    if (s < 0 || n < 2 || s + n > VB_NF) { ok = false; break; }
                    if (s == 0) mf++;
                }
                if (!ok) { fail("feature track outside the window", VILF_ERR_INVALID_ARGUMENT); continue; }
                WindowPlan &pl = h->plans[w];
                pl.nslot = class_plan(in, pl.cls, pl.cstart, pl.ccount);
                L.FACmax = std::max(L.FACmax, pl.nslot); L.Fmax = std::max(L.Fmax, in.n_features); L.Omax = std::max(L.Omax, in.n_obs); L.Mcap = std::max(L.Mcap, MG_MD + mf + 1);
            }
        };
        if (nthr <= 1) check(0);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthr; t++) pool.emplace_back(check, t);
            for (std::thread &th : pool) th.join();
        }
        for (const Local &L : loc) {
            if (L.rc != VILF_OK) { h->err = L.err; return L.rc; }
            Fmax = std::max(Fmax, L.Fmax); Omax = std::max(Omax, L.Omax); FACmax = std::max(FACmax, L.FACmax); Mcap = std::max(Mcap, L.Mcap);
        }
    }
    Fmax = (Fmax + 3) & ~3; FACmax = (FACmax + 63) & ~63;
    h->B = B;
    h->resident = false;
    if ((int)h->priors.size() < B) { vilf_prior z; std::memset(&z, 0, sizeof(z)); h->priors.resize(B, z); }
    if (!keep_priors) { h->prior_dirty.assign(h->priors.size(), 1); h->prior_dev_newer.assign(h->priors.size(), 0); h->prior_slots_valid = 0; }
    h->prior_dirty.resize(h->priors.size(), 1); h->prior_dev_newer.resize(h->priors.size(), 0);
    h->h_mflag.resize(B);
    for (int w = 0; w < B; w++) h->h_mflag[w] = wins[w].marginalization_flag;
    Mcap = (Mcap + 1) & ~1;
    h->mg_Mcap = Mcap;
    const size_t sB = B, sF = Fmax, sO = Omax, sC = FACmax;
    struct Req { int id; size_t bytes; };
    const Req reqs[] = {
        {D_NFEAT, sB * 4}, {D_NFAC, sB * 4}, {D_POSE, sB * 77 * 8}, {D_SB, sB * 99 * 8}, {D_FEAT, sB * sF * 8}, {D_CPOSE, sB * 77 * 8}, {D_CSB, sB * 99 * 8},
        {D_CFEAT, sB * sF * 8}, {D_POSE0, sB * 77 * 8}, {D_SB0, sB * 99 * 8}, {D_FEAT0, sB * sF * 8}, {D_EX, sB * 7 * 8}, {D_GR0, sB * 9 * 8}, {D_GP0, sB * 3 * 8},
        {D_FSTART, sB * sF * 4}, {D_FNOBS, sB * sF * 4}, {D_FOBS0, sB * sF * 4}, {D_FFAC0, sB * sF * 4}, {D_FCONST, sB * sF}, {D_OBS, sB * sO * 3 * 8},
        {D_PSFEAT, sB * sC * 4}, {D_PSOBS, sB * sC * 4}, {D_PSSLOT, sB * sC * 4}, {D_PAIROFF, sB * VB_PTAB * 4}, {D_IMU, sB * 10 * IMU_REC * 8},
        {D_LIDAR, sB * 10 * 7 * 8}, {D_PHDR, sB * VB_PRIOR_HDR * 4}, {D_PX0, sB * 24 * 9 * 8}, {D_PJ, sB * VB_PRIOR_LD * VB_PRIOR_LD * 8}, {D_PR, sB * VB_PRIOR_LD * 8},
        {D_PH, sB * VB_PRIOR_LD * VB_PRIOR_LD * 8}, {D_PG, sB * VB_PRIOR_LD * 8}, {D_FACW, sB * VB_FACW * sC * 8}, {D_HPP, 2 * sB * 66 * 36 * 8},      // 2 x: the two linearisation workspaces (VbState::ws)
        {D_W, 2 * sB * sF * VB_WLD * 8}, {D_HF, 2 * sB * sF * 8}, {D_GF, 2 * sB * sF * 8}, {D_IMUH, 2 * sB * 9000 * 8}, {D_IMUG, 2 * sB * 300 * 8}, {D_LIDH, 2 * sB * 1440 * 8},
        {D_LIDG, 2 * sB * 120 * 8}, {D_G, 2 * sB * VB_P * 8}, {D_DIAGH, 2 * sB * VB_P * 8}, {D_SCALE, sB * (VB_P + sF) * 8}, {D_DIAG, sB * (VB_P + sF) * 8}, {D_GRAD, sB * (VB_P + sF) * 8},
        {D_GN, sB * (VB_P + sF) * 8}, {D_ST, sB * sizeof(VbState)}, {D_OPS, sB * 33 * 8}, {D_ORS, sB * 99 * 8}, {D_OVS, sB * 33 * 8}, {D_OBAS, sB * 33 * 8},
        {D_OBGS, sB * 33 * 8}, {D_PAIRD, 2 * sB * VB_NPAIR * VB_PAIRD * 8}, {D_FACREC, sB * sC * 64}, {D_CF, sB * sF * 8}, {D_TD, sB * 8},
        {D_OBSV, h->opts.estimate_td ? sB * sO * 16 : 8}, {D_OBSTD, h->opts.estimate_td ? sB * sO * 8 : 8}, {D_OBSROW, h->opts.estimate_td ? sB * sO * 8 : 8}, {D_COV, sB * 10 * 225 * 8}, {D_WORK, sB * 10 * 450 * 8}, {D_MFLAG, sB * 4},
    };
    for (const Req &r : reqs) if (!h->d[r.id].ensure(r.bytes)) { h->err = "hipMalloc failed"; return VILF_ERR_DEVICE; }

    lap("validate + device buffers");
    // ---- pack on the host ---------------------------------------------------------------------------------------
    // persistent PINNED staging carved into typed spans: allocating and zero-filling ~250 MB of std::vectors per call was half of the upload time, and copies
    // from pageable memory are neither fast nor asynchronous. Every slice a kernel reads is rewritten by pack_one; padding is never read.
    const bool est_td0 = h->opts.estimate_td != 0;
    int *nfeat, *nfac, *fstart, *fnobs, *fobs0, *ffac0, *facfeat, *facobs, *pairoff, *psfeat, *psobs, *psslot;
    uint8_t *fconst;
    double *pose, *sb, *feat, *ex, *gR0, *gP0, *imu, *lidar, *cov, *facrec, *obsv = nullptr, *obstd = nullptr, *obsrow = nullptr, *td_img = nullptr;
    int *mflag_img = nullptr;
    auto carve_all = [&](Carve &cv) {          // the same sequence sizes the allocation (base = 0) and hands out the spans
        mflag_img = cv.take<int>(sB); td_img = cv.take<double>(sB);
        nfeat = cv.take<int>(sB); nfac = cv.take<int>(sB); fstart = cv.take<int>(sB * sF); fnobs = cv.take<int>(sB * sF); fobs0 = cv.take<int>(sB * sF); ffac0 = cv.take<int>(sB * sF);
        facfeat = cv.take<int>(sB * sC); facobs = cv.take<int>(sB * sC); pairoff = cv.take<int>(sB * VB_PTAB); psfeat = cv.take<int>(sB * sC); psobs = cv.take<int>(sB * sC); psslot = cv.take<int>(sB * sC);
        fconst = cv.take<uint8_t>(sB * sF);
        pose = cv.take<double>(sB * 77); sb = cv.take<double>(sB * 99); feat = cv.take<double>(sB * sF); ex = cv.take<double>(sB * 7); gR0 = cv.take<double>(sB * 9); gP0 = cv.take<double>(sB * 3);
        imu = cv.take<double>(sB * 10 * IMU_REC); lidar = cv.take<double>(sB * 10 * 7); cov = cv.take<double>(sB * 10 * 225); facrec = cv.take<double>(sB * sC * 8);
        if (est_td0) { obsv = cv.take<double>(sB * sO * 2); obstd = cv.take<double>(sB * sO); obsrow = cv.take<double>(sB * sO); }
    };
    size_t image_bytes = 0;
    { Carve sz{nullptr}; carve_all(sz); image_bytes = (sz.off + 63) & ~(size_t)63; if (!h->pin_up.ensure(sz.off + 64)) { h->err = "hipHostMalloc failed (upload staging)"; return VILF_ERR_DEVICE; } }
    { Carve cv{static_cast<char *>(h->pin_up.p)}; carve_all(cv); }
    h->h_nfeat.assign(B, 0); h->h_ex.assign(sB * 7, 0.0); h->h_td.assign(B, 0.0);
    const bool est_any = h->opts.estimate_extrinsic || h->opts.estimate_td, est_td = est_td0;
    h->own.clear();
    if (est_any) h->own.resize(B);
    lap("host vectors");
    auto pack_one = [&](int w) {        // every window writes its own slices only: packed by several host threads below
        const vilf_window_in &in = wins[w];
        const int F = in.n_features;
        nfeat[w] = F; h->h_nfeat[w] = F;
        std::memcpy(&pose[(size_t)w * 77], in.para_pose, 77 * 8);
        std::memcpy(&sb[(size_t)w * 99], in.para_speed_bias, 99 * 8);
        std::memcpy(&ex[(size_t)w * 7], in.para_ex_pose, 7 * 8);
        std::memcpy(&h->h_ex[(size_t)w * 7], in.para_ex_pose, 7 * 8);
        h->h_td[w] = in.para_td;
        if (in.gauge_R0) std::memcpy(&gR0[(size_t)w * 9], in.gauge_R0, 72); else quat_to_R(in.para_pose + 3, &gR0[(size_t)w * 9]);
        if (in.gauge_P0) std::memcpy(&gP0[(size_t)w * 3], in.gauge_P0, 24); else std::memcpy(&gP0[(size_t)w * 3], in.para_pose, 24);
        if (est_td && in.n_obs) {
            std::memcpy(&obsv[(size_t)w * sO * 2], in.obs_velocity, (size_t)in.n_obs * 16);
            std::memcpy(&obstd[(size_t)w * sO], in.obs_cur_td, (size_t)in.n_obs * 8);
            std::memcpy(&obsrow[(size_t)w * sO], in.obs_row, (size_t)in.n_obs * 8);
        }
        if (est_any) {                  // the batched solve of these options runs the general path per slot: keep the inputs
            OwnedWindow &o = h->own[w];
            o.in = in;
            o.pose.assign(in.para_pose, in.para_pose + 77); o.sb.assign(in.para_speed_bias, in.para_speed_bias + 99);
            o.feat.assign(in.para_feature, in.para_feature + F); o.fconst.assign(in.feature_const, in.feature_const + F);
            o.fstart.assign(in.feature_start_frame, in.feature_start_frame + F); o.foff.assign(in.feature_obs_offset, in.feature_obs_offset + F + 1);
            o.obs.assign(in.obs_point, in.obs_point + 3 * (size_t)in.n_obs);
            if (in.obs_velocity) o.vel.assign(in.obs_velocity, in.obs_velocity + 2 * (size_t)in.n_obs);
            if (in.obs_cur_td) o.ctd.assign(in.obs_cur_td, in.obs_cur_td + in.n_obs);
            if (in.obs_row) o.row.assign(in.obs_row, in.obs_row + in.n_obs);
            o.imu.assign(in.imu, in.imu + VB_NF);
            if (in.lidar) o.lidar.assign(in.lidar, in.lidar + VB_NF);
            if (in.gauge_R0) o.gR0.assign(in.gauge_R0, in.gauge_R0 + 9);
            if (in.gauge_P0) o.gP0.assign(in.gauge_P0, in.gauge_P0 + 3);
            o.in.para_pose = o.pose.data(); o.in.para_speed_bias = o.sb.data(); o.in.para_feature = o.feat.data(); o.in.feature_const = o.fconst.data();
            o.in.feature_start_frame = o.fstart.data(); o.in.feature_obs_offset = o.foff.data(); o.in.obs_point = o.obs.data();
            o.in.obs_velocity = o.vel.empty() ? nullptr : o.vel.data(); o.in.obs_cur_td = o.ctd.empty() ? nullptr : o.ctd.data(); o.in.obs_row = o.row.empty() ? nullptr : o.row.data();
            o.in.imu = o.imu.data(); o.in.lidar = o.lidar.empty() ? nullptr : o.lidar.data();
            o.in.gauge_R0 = o.gR0.empty() ? nullptr : o.gR0.data(); o.in.gauge_P0 = o.gP0.empty() ? nullptr : o.gP0.data();
        }
        int fac = 0;
        for (int f = 0; f < F; f++) {
            const int o0 = in.feature_obs_offset[f], o1 = in.feature_obs_offset[f + 1], s = in.feature_start_frame[f];
            feat[(size_t)w * sF + f] = in.para_feature[f];
            fconst[(size_t)w * sF + f] = in.feature_const[f] ? 1 : 0;
            fstart[(size_t)w * sF + f] = s; fnobs[(size_t)w * sF + f] = o1 - o0; fobs0[(size_t)w * sF + f] = o0; ffac0[(size_t)w * sF + f] = fac;
            for (int t = o0 + 1; t < o1; t++) { facfeat[(size_t)w * sC + fac] = f; facobs[(size_t)w * sC + fac] = t; fac++; }
        }
        const WindowPlan &pl = h->plans[w];
        const int *cls = pl.cls, *cstart = pl.cstart, *ccount = pl.ccount, nslot = pl.nslot;
        int cur[VB_NPAIR];
        nfac[w] = nslot;                                   // the kernels sweep slots; unused ones carry a null record
        int *po = &pairoff[(size_t)w * VB_PTAB];
        for (int p = 0; p < VB_NPAIR; p++) { po[2 * p] = cstart[p]; po[2 * p + 1] = ccount[p] | (cls[p] << 24); cur[p] = cstart[p]; }
        po[2 * VB_NPAIR] = 0; po[2 * VB_NPAIR + 1] = 0;
        {   // null records first (flag bit 17), then the factors at their slots
            const unsigned long long nul = 1ULL << 17;
            for (int g = 0; g < nslot; g++) { double *rec = &facrec[((size_t)w * sC + g) * 8]; for (int k = 0; k < 7; k++) rec[k] = 0.0; std::memcpy(&rec[7], &nul, 8); psslot[(size_t)w * sC + g] = 0; psobs[(size_t)w * sC + g] = 0; }
        }
        for (int q = 0; q < fac; q++) {
            const int f = facfeat[(size_t)w * sC + q], t = facobs[(size_t)w * sC + q];
            const int s = in.feature_start_frame[f], j = s + (t - in.feature_obs_offset[f]), p = j * (j - 1) / 2 + s;
            const int pos = VB_SLOT(cls[p], cur[p]); cur[p]++;
            psfeat[(size_t)w * sC + pos] = f; psobs[(size_t)w * sC + pos] = t; psslot[(size_t)w * sC + pos] = q;
            double *rec = &facrec[((size_t)w * sC + pos) * 8];
            const double *pi = in.obs_point + 3 * (size_t)in.feature_obs_offset[f], *pj = in.obs_point + 3 * (size_t)t;
            for (int k = 0; k < 3; k++) { rec[k] = pi[k]; rec[3 + k] = pj[k]; }
            const unsigned long long a = (unsigned long long)(unsigned)f | ((unsigned long long)(unsigned)q << 32);
            const unsigned long long b2 = (unsigned long long)s | ((unsigned long long)j << 8) | ((unsigned long long)(in.feature_const[f] ? 1 : 0) << 16);
            std::memcpy(&rec[6], &a, 8); std::memcpy(&rec[7], &b2, 8);
        }
        for (int k = 0; k < 10; k++) {
            const vilf_imu_preint &p = in.imu[k + 1];
            double *rec = &imu[((size_t)w * 10 + k) * IMU_REC];
            rec[0] = p.sum_dt;
            for (int i = 0; i < 3; i++) { rec[1 + i] = p.delta_p[i]; rec[8 + i] = p.delta_v[i]; rec[11 + i] = p.linearized_ba[i]; rec[14 + i] = p.linearized_bg[i]; }
            for (int i = 0; i < 4; i++) rec[4 + i] = p.delta_q[i];
            auto blk = [&](int off, int r0, int c0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rec[off + 3 * i + j] = p.jacobian[(r0 + i) * 15 + c0 + j]; };
            blk(17, 0, 9); blk(26, 0, 12); blk(35, 3, 12); blk(44, 6, 9); blk(53, 6, 12);
            rec[287] = (p.sum_dt > 10.0) ? 0.0 : 1.0;                           // estimator.cpp:745
            std::memcpy(&cov[((size_t)w * 10 + k) * 225], p.covariance, 225 * 8);
            if (in.lidar) { const vilf_lidar_constraint &c = in.lidar[k + 1]; double *l = &lidar[((size_t)w * 10 + k) * 7]; for (int i = 0; i < 4; i++) l[i] = c.q[i]; for (int i = 0; i < 3; i++) l[4 + i] = c.t[i]; }
            else { double *l = &lidar[((size_t)w * 10 + k) * 7]; for (int i = 0; i < 7; i++) l[i] = (i == 3) ? 1.0 : 0.0; }
        }
    };
    // Packing and copying overlap: the windows are packed in order by the host threads (an atomic cursor), and as soon as every window of a quarter of the batch
    // is done the main thread enqueues that part's slices of the per-window arrays — the DMA engine works while the threads pack the next part
    // (pack 5.1 ms + copy 6.6 ms one after the other before).
    auto upr = [&](int id, const void *src, size_t per_window_bytes, int w0, int w1) {
        return hipMemcpyAsync(static_cast<char *>(h->d[id].p) + (size_t)w0 * per_window_bytes, static_cast<const char *>(src) + (size_t)w0 * per_window_bytes, (size_t)(w1 - w0) * per_window_bytes, hipMemcpyHostToDevice, h->stream);
    };
    auto copy_range = [&](int w0, int w1) -> int {
        HIPCHECK(h, upr(D_POSE, pose, 77 * 8, w0, w1)); HIPCHECK(h, upr(D_POSE0, pose, 77 * 8, w0, w1));
        HIPCHECK(h, upr(D_SB, sb, 99 * 8, w0, w1)); HIPCHECK(h, upr(D_SB0, sb, 99 * 8, w0, w1));
        HIPCHECK(h, upr(D_FEAT, feat, sF * 8, w0, w1)); HIPCHECK(h, upr(D_FEAT0, feat, sF * 8, w0, w1));
        HIPCHECK(h, upr(D_FSTART, fstart, sF * 4, w0, w1)); HIPCHECK(h, upr(D_FNOBS, fnobs, sF * 4, w0, w1));
        HIPCHECK(h, upr(D_FFAC0, ffac0, sF * 4, w0, w1)); HIPCHECK(h, upr(D_FCONST, fconst, sF, w0, w1));
        HIPCHECK(h, upr(D_PSSLOT, psslot, sC * 4, w0, w1));                                   // (obs points and the factor -> feature / observation maps travel inside facrec; the
        if (est_td) { HIPCHECK(h, upr(D_FOBS0, fobs0, sF * 4, w0, w1)); HIPCHECK(h, upr(D_PSOBS, psobs, sC * 4, w0, w1)); }   //  observation indices are only needed by the td factors)
        HIPCHECK(h, upr(D_FACREC, facrec, sC * 64, w0, w1));
        HIPCHECK(h, upr(D_IMU, imu, 10 * IMU_REC * 8, w0, w1)); HIPCHECK(h, upr(D_LIDAR, lidar, 10 * 7 * 8, w0, w1));
        HIPCHECK(h, upr(D_COV, cov, 10 * 225 * 8, w0, w1));
        if (est_td) { HIPCHECK(h, upr(D_OBSV, obsv, sO * 16, w0, w1)); HIPCHECK(h, upr(D_OBSTD, obstd, sO * 8, w0, w1)); HIPCHECK(h, upr(D_OBSROW, obsrow, sO * 8, w0, w1)); }
        return VILF_OK;
    };
    // a small batch: one copy of the whole host image + k_copy_spans (see there)
    const bool staged = nthr <= 1 && image_bytes <= ((size_t)64 << 20) && !std::getenv("VILF_NO_STAGED_UPLOAD") && h->d[D_UPSTAGE].ensure(image_bytes + 4096);
    UpJobs jobs; jobs.n = 0;
    auto span = [&](int id, const void *src, size_t bytes) {          // a span of the image -> the whole of a library array
        UpJob &jb = jobs.j[jobs.n++];
        jb.src = h->d[D_UPSTAGE].as<char>() + (static_cast<const char *>(src) - static_cast<const char *>(h->pin_up.p)); jb.dst = static_cast<char *>(h->d[id].p); jb.bytes = bytes;
    };
    {
        const int nchunk = (nthr > 1 && B >= 256) ? 4 : 1, csz = (B + nchunk - 1) / nchunk;      // (eight parts: no better — 200 copy calls of the main thread compete with the packers)
        if (staged) {
            for (int w = 0; w < B; w++) pack_one(w);
            span(D_POSE, pose, sB * 77 * 8); span(D_POSE0, pose, sB * 77 * 8); span(D_SB, sb, sB * 99 * 8); span(D_SB0, sb, sB * 99 * 8);
            span(D_FEAT, feat, sB * sF * 8); span(D_FEAT0, feat, sB * sF * 8);
            span(D_FSTART, fstart, sB * sF * 4); span(D_FNOBS, fnobs, sB * sF * 4); span(D_FFAC0, ffac0, sB * sF * 4); span(D_FCONST, fconst, sB * sF);
            span(D_PSSLOT, psslot, sB * sC * 4);
            if (est_td) { span(D_FOBS0, fobs0, sB * sF * 4); span(D_PSOBS, psobs, sB * sC * 4); }
            span(D_FACREC, facrec, sB * sC * 64);
            span(D_IMU, imu, sB * 10 * IMU_REC * 8); span(D_LIDAR, lidar, sB * 10 * 7 * 8); span(D_COV, cov, sB * 10 * 225 * 8);
            if (est_td) { span(D_OBSV, obsv, sB * sO * 16); span(D_OBSTD, obstd, sB * sO * 8); span(D_OBSROW, obsrow, sB * sO * 8); }
        }
        else if (nthr <= 1) { for (int w = 0; w < B; w++) pack_one(w); const int rcc = copy_range(0, B); if (rcc != VILF_OK) return rcc; }
        else {
            std::atomic<int> next(0);
            std::vector<std::atomic<int>> done(nchunk);
            for (auto &d : done) d.store(0);
            std::vector<std::thread> pool;
            for (int t = 0; t < nthr; t++) pool.emplace_back([&]() { for (int w = next.fetch_add(1); w < B; w = next.fetch_add(1)) { pack_one(w); done[w / csz].fetch_add(1, std::memory_order_release); } });
            int rcc = VILF_OK;
            for (int c = 0; c < nchunk; c++) {
                const int w0 = c * csz, w1 = std::min(B, w0 + csz);
                while (done[c].load(std::memory_order_acquire) < w1 - w0) std::this_thread::yield();
                if (rcc == VILF_OK) rcc = copy_range(w0, w1);
            }
            for (std::thread &th : pool) th.join();
            if (rcc != VILF_OK) return rcc;
        }
    }
    lap("pack + per-window copies");
    auto up = [&](int id, const void *src, size_t bytes) { return hipMemcpyAsync(h->d[id].p, src, bytes, hipMemcpyHostToDevice, h->stream); };
    if (staged) {
        std::memcpy(mflag_img, h->h_mflag.data(), sB * 4); std::memcpy(td_img, h->h_td.data(), sB * 8);
        span(D_NFEAT, nfeat, sB * 4); span(D_NFAC, nfac, sB * 4); span(D_EX, ex, sB * 7 * 8); span(D_GR0, gR0, sB * 9 * 8); span(D_GP0, gP0, sB * 3 * 8);
        span(D_PAIROFF, pairoff, sB * VB_PTAB * 4); span(D_MFLAG, mflag_img, sB * 4); span(D_TD, td_img, sB * 8);
        HIPCHECK(h, hipMemcpyAsync(h->d[D_UPSTAGE].p, h->pin_up.p, image_bytes, hipMemcpyHostToDevice, h->stream));
        size_t big = 0; for (int k = 0; k < jobs.n; k++) big = std::max<size_t>(big, jobs.j[k].bytes);
        hipLaunchKernelGGL(k_copy_spans, dim3((unsigned)std::max<size_t>(1, std::min<size_t>(64, big / 65536 + 1)), (unsigned)jobs.n), dim3(256), 0, h->stream, jobs);
        HIPCHECK(h, hipGetLastError());
    } else {
    HIPCHECK(h, up(D_NFEAT, nfeat, sB * 4)); HIPCHECK(h, up(D_NFAC, nfac, sB * 4));
    HIPCHECK(h, up(D_EX, ex, sB * 7 * 8)); HIPCHECK(h, up(D_GR0, gR0, sB * 9 * 8)); HIPCHECK(h, up(D_GP0, gP0, sB * 3 * 8));
    HIPCHECK(h, up(D_PAIROFF, pairoff, sB * VB_PTAB * 4));
    std::memcpy(mflag_img, h->h_mflag.data(), sB * 4); std::memcpy(td_img, h->h_td.data(), sB * 8);      // (from the pinned image, like everything else: an asynchronous upload must not read pageable memory the next call rewrites)
    HIPCHECK(h, up(D_MFLAG, mflag_img, sB * 4));
    HIPCHECK(h, up(D_TD, td_img, sB * 8));
    if (!h->async_upload) HIPCHECK(h, hipStreamSynchronize(h->stream));
    }
    lap("small arrays + sync");

    // ---- batch descriptor -------------------------------------------------------------------------------------
    VbBatch &b = h->batch;
    std::memset(&b, 0, sizeof(b));
    const vilf_options &o = h->opts;
    b.B = B; b.w0 = 0; b.Fmax = Fmax; b.Omax = Omax; b.FACmax = FACmax;
    b.sqrt_info = o.focal_length / 1.5; b.cauchy_b = o.cauchy_a * o.cauchy_a;
    for (int i = 0; i < 3; i++) b.G[i] = o.G[i];
    {   // qil = Quaterniond(RIC*RCL), til = RIC*TCL + TIC (lidar_factor.h:28-29)
        double M[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = o.RIC[3 * i] * o.RCL[j] + o.RIC[3 * i + 1] * o.RCL[3 + j] + o.RIC[3 * i + 2] * o.RCL[6 + j];
        quat_from_R(M, b.qil);
        for (int i = 0; i < 3; i++) b.til[i] = o.RIC[3 * i] * o.TCL[0] + o.RIC[3 * i + 1] * o.TCL[1] + o.RIC[3 * i + 2] * o.TCL[2] + o.TIC[i];
    }
    b.use_lidar = o.use_lidar_const; b.max_iterations = o.max_num_iterations;
    b.min_relative_decrease = 1e-3; b.function_tolerance = 1e-6; b.gradient_tolerance = 1e-10; b.parameter_tolerance = 1e-8;
    b.min_radius = 1e-32; b.initial_radius = 1e4; b.min_lm_diagonal = 1e-6; b.max_lm_diagonal = 1e32;
    b.n_feat = h->d[D_NFEAT].as<int>(); b.n_fac = h->d[D_NFAC].as<int>();
    b.pose = h->d[D_POSE].as<double>(); b.sb = h->d[D_SB].as<double>(); b.feat = h->d[D_FEAT].as<double>();
    b.cand_pose = h->d[D_CPOSE].as<double>(); b.cand_sb = h->d[D_CSB].as<double>(); b.cand_feat = h->d[D_CFEAT].as<double>();
    b.pose_init = h->d[D_POSE0].as<double>(); b.sb_init = h->d[D_SB0].as<double>(); b.feat_init = h->d[D_FEAT0].as<double>();
    b.ex = h->d[D_EX].as<double>(); b.gauge_R0 = h->d[D_GR0].as<double>(); b.gauge_P0 = h->d[D_GP0].as<double>();
    b.f_start = h->d[D_FSTART].as<int>(); b.f_nobs = h->d[D_FNOBS].as<int>(); b.f_obs0 = h->d[D_FOBS0].as<int>(); b.f_fac0 = h->d[D_FFAC0].as<int>();
    b.f_const = h->d[D_FCONST].as<uint8_t>(); b.obs = h->d[D_OBS].as<double>();
    b.obs_vel = h->d[D_OBSV].as<double>(); b.obs_ctd = h->d[D_OBSTD].as<double>(); b.obs_row = h->d[D_OBSROW].as<double>(); b.td = h->d[D_TD].as<double>();
    b.est_td = o.estimate_td ? 1 : 0; b.tr_over_row = o.TR / o.ROW; b.row_half = o.ROW / 2;
    b.ps_feat = h->d[D_PSFEAT].as<int>(); b.ps_obs = h->d[D_PSOBS].as<int>(); b.ps_slot = h->d[D_PSSLOT].as<int>();
    b.pair_off = h->d[D_PAIROFF].as<int>(); b.facrec = h->d[D_FACREC].as<double>();
    b.imu = h->d[D_IMU].as<double>(); b.lidar = h->d[D_LIDAR].as<double>();
    bind_prior_pointers(h);
    b.facw = h->d[D_FACW].as<double>(); b.Hpp = h->d[D_HPP].as<double>(); b.W = h->d[D_W].as<double>(); b.hf = h->d[D_HF].as<double>(); b.gf = h->d[D_GF].as<double>();
    b.imuH = h->d[D_IMUH].as<double>(); b.imug = h->d[D_IMUG].as<double>(); b.lidH = h->d[D_LIDH].as<double>(); b.lidg = h->d[D_LIDG].as<double>(); b.g = h->d[D_G].as<double>(); b.diagH = h->d[D_DIAGH].as<double>(); b.pairD = h->d[D_PAIRD].as<double>();
    b.cf = h->d[D_CF].as<double>();
    b.scale = h->d[D_SCALE].as<double>(); b.diag = h->d[D_DIAG].as<double>(); b.grad = h->d[D_GRAD].as<double>(); b.gn = h->d[D_GN].as<double>();
    b.st = h->d[D_ST].as<VbState>();
    b.out_Ps = h->d[D_OPS].as<double>(); b.out_Rs = h->d[D_ORS].as<double>(); b.out_Vs = h->d[D_OVS].as<double>();
    b.out_Bas = h->d[D_OBAS].as<double>(); b.out_Bgs = h->d[D_OBGS].as<double>();
    b.dbg = nullptr;
    if (getenv("VILF_DEBUG_STAMPS")) { if (!h->d[D_DBG].ensure(3 * 32 * 8)) return VILF_ERR_DEVICE; hipMemsetAsync(h->d[D_DBG].p, 0, 3 * 32 * 8, h->stream); b.dbg = h->d[D_DBG].as<long long>(); }

    HIPCHECK(h, hipMemsetAsync(h->d[D_W].p, 0, 2 * sB * sF * VB_WLD * 8, h->stream));   // W rows are zero outside the rewritten ranges
    if (!h->luts_ready) {   // static scatter tables of the tile assembly (same for every window): source element -> LDS offset, -1 = not stored
        auto perm = [](int a, int l) { return l < 6 ? 6 * a + l : 66 + 9 * a + (l - 6); };
        // packed entry: bits 0..14 = LDS offset + 1 (0: element not stored, upper block triangle), bits 15..22 = row, bits 23..30 = column
        auto off1 = [](int r, int c) { const int tr = r >> 4, tc = c >> 4; if (tr < tc) return 0; return (tr * (tr + 1) / 2 + tc) * 256 + 16 * (r & 15) + ((c & 15) ^ (r & 15)) + 1; };
        auto off = [&](int r, int c) { return off1(r, c) | (r << 15) | (c << 23); };
        std::vector<int> li(9000), ll(1440), lv(2 * 2376);
        for (int k = 0; k < 10; k++) {
            for (int e = 0; e < 900; e++) { const int p = e / 30, q = e % 30; li[900 * k + e] = off(perm(k + p / 15, p % 15), perm(k + q / 15, q % 15)); }
            for (int e = 0; e < 144; e++) { const int p = e / 12, q = e % 12; ll[144 * k + e] = off(perm(k + p / 6, p % 6), perm(k + q / 6, q % 6)); }
        }
        for (int t = 0; t < 2376; t++) {
            const int blk = t / 36, e = t % 36, l = e / 6, m = e % 6;
            int a = 0; while ((a + 1) * (a + 2) / 2 <= blk) a++;
            const int bb = blk - a * (a + 1) / 2, r = 6 * a + l, c = 6 * bb + m;
            lv[2 * t] = off(r, c);
            lv[2 * t + 1] = (a != bb && (r >> 4) == (c >> 4)) ? off1(c, r) : 0;
        }
        if (!h->d[D_LUTI].ensure(li.size() * 4) || !h->d[D_LUTL].ensure(ll.size() * 4) || !h->d[D_LUTV].ensure(lv.size() * 4)) return VILF_ERR_DEVICE;
        HIPCHECK(h, vilf_copy_sync(h, h->d[D_LUTI].p, li.data(), li.size() * 4, hipMemcpyHostToDevice));
        HIPCHECK(h, vilf_copy_sync(h, h->d[D_LUTL].p, ll.data(), ll.size() * 4, hipMemcpyHostToDevice));
        HIPCHECK(h, vilf_copy_sync(h, h->d[D_LUTV].p, lv.data(), lv.size() * 4, hipMemcpyHostToDevice));
        h->luts_ready = true;
    }
    h->batch.lut_imu = h->d[D_LUTI].as<int>(); h->batch.lut_lid = h->d[D_LUTL].as<int>(); h->batch.lut_vis = h->d[D_LUTV].as<int>();
    h->batch.sb_tab = h->d[D_SBTAB].as<int>();
    const int nimu = B * 10;
    hipLaunchKernelGGL(k_imu_prep, dim3((nimu + 3) / 4), dim3(64), 0, h->stream, nimu, h->d[D_COV].as<double>(), h->d[D_WORK].as<double>(), h->d[D_IMU].as<double>());
    HIPCHECK(h, hipGetLastError());
    int rc = upload_priors(h);
    if (rc != VILF_OK) return rc;
    hipLaunchKernelGGL(k_reset, dim3(B), dim3(VB_NT), 0, h->stream, h->batch, 0);
    HIPCHECK(h, hipGetLastError());
    if (h->async_upload || (h->defer_upload_sync && staged)) h->upload_inflight = true;          // the next upload of this handle waits before it touches the staging (also when
                                                                                                 // vilf_window_solve returns early on an error, before its own wait for the stream)
    else HIPCHECK(h, hipStreamSynchronize(h->stream));
    h->resident = true;
    return VILF_OK;
}

// the library's radix sort (vilf_sort.hip) through host buffers — a test hook (tests/test_scan2map.py compares it with a stable host sort); not part of include/vilfusion.h.
// keys: n values of 4 (key64 == 0) or 8 bytes; the low `bits` bits are sorted, equal keys keep their order.
extern "C" int vilf_debug_sort_pairs(vilf_handle *h, const void *keys, const int *vals, size_t n, int bits, int key64, void *keys_out, int *vals_out) {
    if (!h || !keys || !vals || !keys_out || !vals_out || n == 0) return VILF_ERR_INVALID_ARGUMENT;
    const size_t kb = key64 ? 8 : 4, tb = vilf_sort_temp_bytes(n, kb);
    void *dk = nullptr, *dk2 = nullptr, *dv = nullptr, *dv2 = nullptr, *dt = nullptr;
    int rc = VILF_OK;
    if (hipMalloc(&dk, n * kb) != hipSuccess || hipMalloc(&dk2, n * kb) != hipSuccess || hipMalloc(&dv, n * 4) != hipSuccess || hipMalloc(&dv2, n * 4) != hipSuccess || hipMalloc(&dt, tb) != hipSuccess) rc = VILF_ERR_DEVICE;
    if (rc == VILF_OK && (hipMemcpyAsync(dk, keys, n * kb, hipMemcpyHostToDevice, h->stream) != hipSuccess || hipMemcpyAsync(dv, vals, n * 4, hipMemcpyHostToDevice, h->stream) != hipSuccess)) rc = VILF_ERR_DEVICE;
    if (rc == VILF_OK) {
        const int r = key64 ? vilf_sort_pairs_u64(h->stream, dt, tb, (const unsigned long long *)dk, (unsigned long long *)dk2, (const int *)dv, (int *)dv2, n, bits)
                            : vilf_sort_pairs_u32(h->stream, dt, tb, (const unsigned *)dk, (unsigned *)dk2, (const int *)dv, (int *)dv2, n, bits);
        if (r != 0) rc = VILF_ERR_DEVICE;
    }
    if (rc == VILF_OK && (hipMemcpyAsync(keys_out, dk2, n * kb, hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipMemcpyAsync(vals_out, dv2, n * 4, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                          hipStreamSynchronize(h->stream) != hipSuccess)) rc = VILF_ERR_DEVICE;
    hipFree(dk); hipFree(dk2); hipFree(dv); hipFree(dv2); hipFree(dt);
    return rc;
}
// dynamic LDS the window kernels are launched with (bytes): [0] k_linearize / k_linearize_split / k_linearize_last, [1] k_solve_sb, [2] k_iter. Diagnostic (bench.py
// quotes them beside the registers it reads from the code objects); not part of include/vilfusion.h.
extern "C" int vilf_debug_lds_bytes(vilf_handle *h, int out3[3]) {
    if (!h || !out3) return VILF_ERR_INVALID_ARGUMENT;
    out3[0] = (int)h->lin_lds; out3[1] = (int)h->solve_sb_lds; out3[2] = (int)std::max(h->lin_lds, h->solve_sb_lds);
    return VILF_OK;
}
extern "C" int vilf_debug_stamps(vilf_handle *h, long long *out96) {
    if (!h || !h->batch.dbg) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, vilf_copy_sync(h, out96, h->batch.dbg, 96 * 8, hipMemcpyDeviceToHost));
    return VILF_OK;
}

extern "C" int vilf_batch_rewind(vilf_handle *h) {
    if (!h || !h->resident) return VILF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(k_reset, dim3(h->B), dim3(VB_NT), 0, h->stream, h->batch, 1);
    HIPCHECK(h, hipGetLastError());
    if ((h->opts.estimate_extrinsic || h->opts.estimate_td) && (int)h->own.size() == h->B) {      // Ex_Pose / td are variables then: their uploaded values are part of the state
        for (int w = 0; w < h->B; w++) { std::memcpy(&h->h_ex[(size_t)w * 7], h->own[w].in.para_ex_pose, 56); h->h_td[w] = h->own[w].in.para_td; }
        HIPCHECK(h, hipMemcpyAsync(h->d[D_EX].p, h->h_ex.data(), (size_t)h->B * 56, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipMemcpyAsync(h->d[D_TD].p, h->h_td.data(), (size_t)h->B * 8, hipMemcpyHostToDevice, h->stream));
    }
    if (h->prior_restore_needed && h->prior_backup_valid) {          // a marginalization replaced the priors: the set as uploaded becomes the live one again (swap, no copy)
        const int live[6] = {D_PHDR, D_PX0, D_PJ, D_PR, D_PH, D_PG}, bak[6] = {D_PHDR0, D_PX00, D_PJ0, D_PR0, D_PH0, D_PG0};
        for (int k = 0; k < 6; k++) std::swap(h->d[live[k]], h->d[bak[k]]);
        bind_prior_pointers(h);
        // the device now holds the authoritative priors; the host mirror may have seen the marginalized ones through an export
        for (int w = 0; w < h->B; w++) { h->prior_dev_newer[w] = 1; h->prior_dirty[w] = 0; }
        // (the restored set is the uploaded one: prior_dense / solve_dense_fallback describe exactly it)
        h->prior_restore_needed = false;
    }
    return VILF_OK;
}

// usec_solve of an asynchronous solve: its two events are read once the stream has been waited for (summaries / download_states)
static void solve_time_resolve(vilf_handle *h) {
    if (!h->solve_time_pending) return;
    float ms = 0;
    if (hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess) h->last_solve_usec = ms * 1000.0;
    h->solve_time_pending = false;
}
static bool tlim_disabled(vilf_handle *h) { return !(h->opts.max_solver_time > 0); }
extern "C" int vilf_batch_solve(vilf_handle *h, int sync) {
    if (!h || !h->resident) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    if (h->opts.estimate_extrinsic || h->opts.estimate_td) {
        // Ex_Pose / td as variables (estimator.cpp:701-717; both off in the KITTI configuration): every slot through the general single-window path
        // (vilf_lw.hip: ProjectionTdFactor / Ex_Pose Jacobians, the slot's device-resident prior), one after the other. State, gauge-fixed outputs and the
        // summary are written back to the slot, so download / summaries / marginalization continue as after the batched kernels.
        if ((int)h->own.size() != h->B) { h->err = "batch inputs not retained"; return VILF_ERR_INVALID_ARGUMENT; }
        bool dirty0 = false;
        for (int w = 0; w < h->B; w++) if (h->prior_dirty[w]) dirty0 = true;
        if (dirty0) { int rc = upload_priors(h); if (rc != VILF_OK) return rc; }
        const auto t0 = std::chrono::steady_clock::now();
        // all slots as ONE group: a single chain of launches solves them side by side (vilf_lw_group_solve), priors in and states / summaries back in bulk copies
        const size_t B = h->B;
        std::vector<double> bufP(B * 77), bufS(B * 99), bufF(B * (h->batch.Fmax + 4)), Ps(B * 33), Rs(B * 99), Vs(B * 33), Bas(B * 33), Bgs(B * 33);
        std::vector<vilf_window_out> outv(B);
        std::vector<const vilf_window_in *> inp(B);
        std::vector<vilf_window_out *> outp(B);
        std::vector<int> slot1(B);
        for (size_t w = 0; w < B; w++) {
            vilf_window_out &out = outv[w];
            std::memset(&out, 0, sizeof(out));
            out.para_pose = &bufP[w * 77]; out.para_speed_bias = &bufS[w * 99]; out.para_feature = &bufF[w * (h->batch.Fmax + 4)];
            out.Ps = &Ps[w * 33]; out.Rs = &Rs[w * 99]; out.Vs = &Vs[w * 33]; out.Bas = &Bas[w * 33]; out.Bgs = &Bgs[w * 33];
            inp[w] = &h->own[w].in; outp[w] = &out; slot1[w] = (int)w + 1;
        }
        const int rc = vilf_lw_group_solve(h, h->B, inp.data(), outp.data(), slot1.data());
        if (rc < 0) return rc;
        h->last_solve_usec = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        h->solve_time_pending = false;
        (void)sync;                             // the general path reads its results back: always synchronous
        return VILF_OK;
    }
    bool dirty = false;
    for (int w = 0; w < h->B; w++) if (h->prior_dirty[w]) dirty = true;
    if (dirty) { int rc = upload_priors(h); if (rc != VILF_OK) return rc; }
    const dim3 grid(h->B), block(VB_NT);
    // k_solve_sb eliminates SpeedBias[1..10] as a block-tridiagonal chain: valid while the priors hold no speed-bias block but SpeedBias[0] (all the
    // reference ever produces, estimator.cpp:960-971); VILF_SOLVE_DENSE=1 forces the dense-Cholesky kernel (tests compare the two)
    // (solve_dense_fallback describes the priors as uploaded; after a device marginalization the live priors hold SpeedBias[0] only — prior_restore_needed — until a rewind)
    const bool dense = (h->solve_dense_fallback && !h->prior_restore_needed) || std::getenv("VILF_SOLVE_DENSE") != nullptr;
    const bool prof = h->profiling != 0;
    if (prof && h->prof_used.size() > 4096) { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }      // asynchronous calls without a reader: bound the pool
    std::vector<int> kinds;
    int ne = 0;
    std::vector<hipEvent_t> pev;
    auto mark = [&](int kind) { if (prof) { pev.push_back(vilf_prof_event(h)); ne++; kinds.push_back(kind); } };
    hipEventRecord(h->ev0, h->stream);
    h->batch.w0 = 0;
    // Experiment (VILF_SOLVE_SPLIT=n): the batch as n parts on n streams, each part running its own chain of launches. The two kernels of an iteration then meet on the
    // chip (one part's k_solve_sb beside another part's k_linearize) instead of the whole chip running one kernel at a time.
    if (const char *es = std::getenv("VILF_SOLVE_SPLIT")) {
        const int np = std::max(2, std::min(8, std::atoi(es)));
        if (!dense && tlim_disabled(h) && h->B >= 2 * np) {
            if ((int)h->split_streams.size() < np) { h->split_streams.resize(np, nullptr); for (auto &st : h->split_streams) if (!st) HIPCHECK(h, hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); }
            if (!h->split_ev) HIPCHECK(h, hipEventCreateWithFlags(&h->split_ev, hipEventDisableTiming));
            HIPCHECK(h, hipEventRecord(h->split_ev, h->stream));
            for (int p = 0; p < np; p++) {
                hipStream_t st = h->split_streams[p];
                HIPCHECK(h, hipStreamWaitEvent(st, h->split_ev, 0));
                VbBatch bb = h->batch;
                bb.w0 = (int)((long long)h->B * p / np);
                const dim3 g2((unsigned)((long long)h->B * (p + 1) / np - bb.w0));
                hipLaunchKernelGGL(k_reset, g2, block, 0, st, bb, 0);
                hipLaunchKernelGGL(k_linearize, g2, block, h->lin_lds, st, bb, 1);
                for (int it = 0; it < h->opts.max_num_iterations; it++) {
                    hipLaunchKernelGGL(k_solve_sb, g2, dim3(256), h->solve_sb_lds, st, bb);
                    if (it + 1 == h->opts.max_num_iterations) hipLaunchKernelGGL(k_linearize_last, g2, block, h->lin_lds, st, bb);
                    else hipLaunchKernelGGL(k_linearize, g2, block, h->lin_lds, st, bb, 0);
                }
                hipLaunchKernelGGL(k_finalize, g2, dim3(64), 0, st, bb);
            }
            for (int p = 0; p < np; p++) {                       // join: the handle's stream continues after every part
                HIPCHECK(h, hipEventRecord(h->split_ev, h->split_streams[p]));
                HIPCHECK(h, hipStreamWaitEvent(h->stream, h->split_ev, 0));
            }
            hipEventRecord(h->ev1, h->stream);
            HIPCHECK(h, hipGetLastError());
            h->solve_time_pending = true;
            if (sync) { HIPCHECK(h, hipStreamSynchronize(h->stream)); solve_time_resolve(h); }
            return VILF_OK;
        }
    }
    // live-window lists (vilf_batch.hpp): the launches of iteration i address their windows through the list k_linearize of iteration i - 1 left, once something has stopped
    const int max_it = h->opts.max_num_iterations;
    const bool use_live = !std::getenv("VILF_NO_LIVE_LIST") && max_it + 2 <= 64 && h->d[D_LIVE].ensure(((size_t)2 * h->B + 128) * sizeof(int));
    // small batches: every window's linearisation over several workgroups (k_linearize_split: one per factor chunk + two for the IMU / LiDAR / prior parts)
    const int nch = h->batch.FACmax / VB_CHUNK;
    const bool split = !dense && h->B <= VB_SPLIT_MAXB && nch >= 1 && nch <= VB_SPLIT_MAXCH && !std::getenv("VILF_NO_LIN_SPLIT") &&
                       h->d[D_SPLITC].ensure((size_t)h->B * VB_SPLIT_CTL * sizeof(int)) && h->d[D_SPLITB].ensure((size_t)h->B * VB_SPLIT_DBL * sizeof(double));
    if (split) HIPCHECK(h, hipMemsetAsync(h->d[D_SPLITC].p, 0, (size_t)h->B * VB_SPLIT_CTL * sizeof(int), h->stream));
    auto with_lists = [&](int it) {
        VbBatch bb = h->batch;
        if (use_live && !split) { bb.live_ctl = h->d[D_LIVE].as<int>(); bb.live_buf = bb.live_ctl + 128; bb.live_it = it; }
        return bb;
    };
    auto linearize = [&](const VbBatch &bb0, int iteration_zero) {
        if (split) {
            VbBatch bb = bb0;
            bb.split_nr = nch + 2; bb.split_ctl = h->d[D_SPLITC].as<int>(); bb.split_buf = h->d[D_SPLITB].as<double>();
            if (++h->split_gen <= 0) h->split_gen = 1;            // the launch's generation: what its hand-over flags carry (never 0 = the cleared state)
            bb.split_gen = h->split_gen;
            bb.split_fault = std::getenv("VILF_SPLIT_FAULT") != nullptr;
            hipLaunchKernelGGL(k_linearize_split, dim3((unsigned)(h->B * (nch + 2))), block, h->lin_lds, h->stream, bb, iteration_zero);
        } else hipLaunchKernelGGL(k_linearize, grid, block, h->lin_lds, h->stream, bb0, iteration_zero);
    };
    // One launch per iteration (k_iter: step + linearisation at the candidate + accept / reject + reduce + solve in the same persistent workgroup, the hand-over in the
    // workgroup's own scratch slot): whenever the speed-bias-first solve applies, the batch is not split over workgroups and no host clock runs between the iterations.
    // (tests compare it with the two-kernel sequence to the bit)
    // MEASURED (round 5, same box, 4096 windows, ms per 8-iteration solve): two kernels 15.6, k_iter on the windows' own workspaces 16.2, k_iter on slots 17.8 — the
    // hand-over through HBM is not what the iteration waits for (HISTORY.md). k_iter therefore stays an experiment: VILF_FUSED=1 selects it, VILF_NO_SLOTS=1 its
    // per-window-workspace form; the default is the two-kernel sequence.
    const bool fused = !dense && !split && !(h->opts.max_solver_time > 0) && std::getenv("VILF_FUSED") && !std::getenv("VILF_NO_FUSED") && h->d[D_ITERQ].ensure(128 * sizeof(int));
    if (fused) {
        const size_t iter_lds = std::max(h->lin_lds, h->solve_sb_lds);
        // batches of more than 1024 windows work in 1024 scratch slots (the first 1024 workspaces of set 0) handed out per XCD by a bitmap: [0..31] free bits, [32] error
        const bool slots = h->B > 1024 && !std::getenv("VILF_NO_SLOTS");
        unsigned *bm = slots ? h->d[D_ITERQ].as<unsigned>() : nullptr;
        if (slots) { HIPCHECK(h, hipMemsetAsync(h->d[D_ITERQ].p, 0xff, 32 * sizeof(int), h->stream)); HIPCHECK(h, hipMemsetAsync(h->d[D_ITERQ].as<int>() + 32, 0, 4 * sizeof(int), h->stream)); }
        mark(3);
        hipLaunchKernelGGL(k_reset, grid, block, 0, h->stream, h->batch, 0);
        for (int it = 0; it < max_it; it++) {
            mark(1);             // kind 1 with no kind-0 launches beside it = the iteration kernel (bench.py names it k_iter)
            hipLaunchKernelGGL(k_iter, grid, block, iter_lds, h->stream, h->batch, it == 0 ? 1 : 0, bm, h->d[D_ITERQ].as<int>() + 32);
        }
        mark(2);
        hipLaunchKernelGGL(k_linearize_last, grid, block, h->lin_lds, h->stream, h->batch);
        mark(3);
        hipLaunchKernelGGL(k_finalize, grid, dim3(64), 0, h->stream, h->batch);
        if (prof) {
            pev.push_back(vilf_prof_event(h)); ne++;
            for (size_t i = 0; i < kinds.size(); i++) vilf_prof_span(h, pev[i], pev[i + 1], &h->kernel_ms[kinds[i]], &h->kernel_launches[kinds[i]]);
        }
        hipEventRecord(h->ev1, h->stream);
        HIPCHECK(h, hipGetLastError());
        h->solve_time_pending = true;
        if (sync) {
            HIPCHECK(h, hipStreamSynchronize(h->stream));
            solve_time_resolve(h);
            if (prof) { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
            if (slots) {                                 // a workgroup that found no free slot left its window untouched and said so
                int e = 0;
                HIPCHECK(h, hipMemcpy(&e, h->d[D_ITERQ].as<int>() + 32, sizeof(int), hipMemcpyDeviceToHost));
                if (e) { h->err = "k_iter: more workgroups resident than workspace slots"; return VILF_ERR_DEVICE; }
            }
        }
        return VILF_OK;
    }
    mark(3);
    hipLaunchKernelGGL(k_reset, grid, block, 0, h->stream, with_lists(0), 0);
    mark(0);
    linearize(h->batch, 1);
    // options.max_solver_time (estimator.cpp:847-850: SOLVER_TIME, x 4/5 when the oldest frame is marginalized): Ceres tests the wall clock at the top of
    // every iteration. The iterations of a batch run in lockstep, so the host waits for the stream before each one (only when the limit is on) and
    // stops the windows whose limit has passed (termination NO_CONVERGENCE, like Ceres' "maximum solver time reached").
    const double tlim = h->opts.max_solver_time;
    const auto t_begin = std::chrono::steady_clock::now();
    bool stopped_old = false;
    for (int it = 0; it < h->opts.max_num_iterations; it++) {
        if (tlim > 0) {
            HIPCHECK(h, hipStreamSynchronize(h->stream));
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
            if (el >= tlim) { hipLaunchKernelGGL(k_time_limit, grid, dim3(64), 0, h->stream, h->batch, h->d[D_MFLAG].as<int>(), 0); break; }
            if (el >= tlim * 4.0 / 5.0 && !stopped_old) { hipLaunchKernelGGL(k_time_limit, grid, dim3(64), 0, h->stream, h->batch, h->d[D_MFLAG].as<int>(), 1); stopped_old = true; }
        }
        mark(1);
        const bool last = it + 1 == h->opts.max_num_iterations;
        const VbBatch bs = with_lists(it + 1), bl = bs;
        if (dense) hipLaunchKernelGGL(k_solve, grid, dim3(512), h->solve_lds, h->stream, bs);
        else hipLaunchKernelGGL(k_solve_sb, grid, dim3(256), h->solve_sb_lds, h->stream, bs);
        mark(last ? 2 : 0);      // kind 2 ("k_step" in the bench line): the step-only launch that ends a solve
        // the step of the last iteration needs no linearisation behind it (nothing solves with it): residuals only
        if (last) hipLaunchKernelGGL(k_linearize_last, grid, block, h->lin_lds, h->stream, bl);
        else linearize(bl, 0);
    }
    mark(3);
    hipLaunchKernelGGL(k_finalize, grid, dim3(64), 0, h->stream, h->batch);
    if (prof) {
        pev.push_back(vilf_prof_event(h)); ne++;
        for (size_t i = 0; i < kinds.size(); i++) vilf_prof_span(h, pev[i], pev[i + 1], &h->kernel_ms[kinds[i]], &h->kernel_launches[kinds[i]]);
    }
    hipEventRecord(h->ev1, h->stream);
    HIPCHECK(h, hipGetLastError());
    h->solve_time_pending = true;
    if (sync) {
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        solve_time_resolve(h);
        if (prof) { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    }                                           // sync == 0 with profiling on: the spans stay pending (read by the next call that waits for the stream)
    return VILF_OK;
}

// per-kernel timing with HIP events on the handle's stream (bench.py roofline). kind: 0 linearize, 1 solve, 2 step, 3 other
extern "C" int vilf_set_profiling(vilf_handle *h, int on) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    // hipEventCreate is slow enough to starve the stream when it happens between launches (measured: 1.9 ms per frame for ~50 events): the pool is filled here
    if (on) { HIPCHECK(h, hipSetDevice(h->device)); while (h->prof_free.size() < 1024) { hipEvent_t e; HIPCHECK(h, hipEventCreate(&e)); h->prof_free.push_back(e); } }
    h->profiling = on;
    for (int i = 0; i < 4; i++) { h->kernel_ms[i] = 0; h->kernel_launches[i] = 0; }
    for (int i = 0; i < 8; i++) { h->s2m_ms[i] = 0; h->s2m_launches[i] = 0; }
    for (int i = 0; i < 4; i++) { h->marg_ms[i] = 0; h->marg_launches[i] = 0; }
    return VILF_OK;
}
extern "C" int vilf_get_profile(vilf_handle *h, double ms_out[4], long launches_out[4]) {
    if (!h || !ms_out || !launches_out) return VILF_ERR_INVALID_ARGUMENT;
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    for (int i = 0; i < 4; i++) { ms_out[i] = h->kernel_ms[i]; launches_out[i] = h->kernel_launches[i]; }
    return VILF_OK;
}

extern "C" int vilf_get_profile_marginalize(vilf_handle *h, double ms_out[4], long launches_out[4]) {
    if (!h || !ms_out || !launches_out) return VILF_ERR_INVALID_ARGUMENT;
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    for (int i = 0; i < 4; i++) { ms_out[i] = h->marg_ms[i]; launches_out[i] = h->marg_launches[i]; }
    return VILF_OK;
}

// a window whose kernel gave up a bounded device-side wait (VbState::dev_error): no numbers are handed out for it
static int vb_check_dev_error(vilf_handle *h, const VbState *st, int first, int n) {
    for (int i = 0; i < n; i++) if (st[i].dev_error) {
        h->err = "window " + std::to_string(first + i) + ": a workgroup of k_linearize_split gave up waiting for another one's hand-over (bounded wait; the results of this solve are not valid)";
        return VILF_ERR_DEVICE;
    }
    return VILF_OK;
}
extern "C" int vilf_batch_summaries(vilf_handle *h, int first, int n, vilf_summary *sums) {
    if (!h || !h->resident || first < 0 || n < 0 || first + n > h->B || !sums) return VILF_ERR_INVALID_ARGUMENT;
    std::vector<VbState> st(n);
    HIPCHECK(h, hipMemcpyAsync(st.data(), h->batch.st + first, sizeof(VbState) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    solve_time_resolve(h);
    { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }     // the stream is idle: pending profile spans cost nothing to read now
    { const int rce = vb_check_dev_error(h, st.data(), first, n); if (rce != VILF_OK) return rce; }
    for (int i = 0; i < n; i++) {
        sums[i].num_iterations = st[i].iteration;
        sums[i].num_successful_steps = st[i].num_successful;
        sums[i].num_linear_solves = st[i].num_linear_solves;
        sums[i].termination = st[i].termination;
        sums[i].initial_cost = st[i].initial_cost;
        sums[i].final_cost = st[i].x_cost;
        sums[i].final_radius = st[i].radius;
        sums[i].usec_solve = h->last_solve_usec;
    }
    return VILF_OK;
}

extern "C" int vilf_batch_download(vilf_handle *h, int first, int n, vilf_window_out *outs) {
    if (!h || !h->resident || first < 0 || n < 0 || first + n > h->B || !outs) return VILF_ERR_INVALID_ARGUMENT;
    const size_t sF = h->batch.Fmax;
    // a few windows (the single-window entry point): the nine result arrays are gathered on the device (k_copy_spans) and come back in ONE copy through pinned memory
    const size_t sn0 = n, per0 = 77 + 99 + sF + 33 + 99 + 33 + 33 + 33, img0 = ((sn0 * per0 * 8 + 63) & ~(size_t)63) + sn0 * sizeof(VbState);
    if (n <= 64 && !std::getenv("VILF_NO_STAGED_UPLOAD") && h->d[D_DNSTAGE].ensure(img0 + 256) && h->pin_down.ensure(img0 + 256)) {
        UpJobs jobs; jobs.n = 0;
        char *dst = h->d[D_DNSTAGE].as<char>();
        size_t off = 0;
        auto gather = [&](const void *src, size_t bytes) { UpJob &jb = jobs.j[jobs.n++]; jb.src = static_cast<const char *>(src); jb.dst = dst + off; jb.bytes = bytes; const size_t at = off; off += (bytes + 15) & ~(size_t)15; return at; };
        const size_t o_pose = gather(h->batch.pose + (size_t)first * 77, sn0 * 77 * 8), o_sb = gather(h->batch.sb + (size_t)first * 99, sn0 * 99 * 8), o_feat = gather(h->batch.feat + (size_t)first * sF, sn0 * sF * 8);
        const size_t o_ps = gather(h->batch.out_Ps + (size_t)first * 33, sn0 * 33 * 8), o_rs = gather(h->batch.out_Rs + (size_t)first * 99, sn0 * 99 * 8), o_vs = gather(h->batch.out_Vs + (size_t)first * 33, sn0 * 33 * 8);
        const size_t o_ba = gather(h->batch.out_Bas + (size_t)first * 33, sn0 * 33 * 8), o_bg = gather(h->batch.out_Bgs + (size_t)first * 33, sn0 * 33 * 8), o_st = gather(h->batch.st + first, sn0 * sizeof(VbState));
        hipLaunchKernelGGL(k_copy_spans, dim3(1, (unsigned)jobs.n), dim3(256), 0, h->stream, jobs);
        HIPCHECK(h, hipGetLastError());
        HIPCHECK(h, hipMemcpyAsync(h->pin_down.p, dst, off, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        solve_time_resolve(h);
        { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
        const char *img = static_cast<const char *>(h->pin_down.p);
        auto dd = [&](size_t at) { return reinterpret_cast<const double *>(img + at); };
        const VbState *st = reinterpret_cast<const VbState *>(img + o_st);
        { const int rce = vb_check_dev_error(h, st, first, n); if (rce != VILF_OK) return rce; }
        for (int i = 0; i < n; i++) {
            vilf_window_out &o = outs[i];
            const int F = h->h_nfeat[first + i];
            if (o.para_pose) std::memcpy(o.para_pose, dd(o_pose) + (size_t)i * 77, 77 * 8);
            if (o.para_speed_bias) std::memcpy(o.para_speed_bias, dd(o_sb) + (size_t)i * 99, 99 * 8);
            if (o.para_feature && F) std::memcpy(o.para_feature, dd(o_feat) + (size_t)i * sF, (size_t)F * 8);
            if (o.Ps) std::memcpy(o.Ps, dd(o_ps) + (size_t)i * 33, 33 * 8);
            if (o.Rs) std::memcpy(o.Rs, dd(o_rs) + (size_t)i * 99, 99 * 8);
            if (o.Vs) std::memcpy(o.Vs, dd(o_vs) + (size_t)i * 33, 33 * 8);
            if (o.Bas) std::memcpy(o.Bas, dd(o_ba) + (size_t)i * 33, 33 * 8);
            if (o.Bgs) std::memcpy(o.Bgs, dd(o_bg) + (size_t)i * 33, 33 * 8);
            const double *exw = &h->h_ex[(size_t)(first + i) * 7];
            for (int k = 0; k < 3; k++) o.tic[k] = exw[k];
            quat_to_R(exw + 3, o.ric);
            o.td = h->h_td[first + i];
            o.summary.num_iterations = st[i].iteration; o.summary.num_successful_steps = st[i].num_successful; o.summary.num_linear_solves = st[i].num_linear_solves;
            o.summary.termination = st[i].termination; o.summary.initial_cost = st[i].initial_cost; o.summary.final_cost = st[i].x_cost; o.summary.final_radius = st[i].radius;
            o.summary.usec_solve = h->last_solve_usec;
        }
        return VILF_OK;
    }
    std::vector<double> pose((size_t)n * 77), sb((size_t)n * 99), feat((size_t)n * sF), Ps((size_t)n * 33), Rs((size_t)n * 99), Vs((size_t)n * 33), Bas((size_t)n * 33), Bgs((size_t)n * 33);
    auto dn = [&](void *dst, const double *src, size_t cnt) { return hipMemcpyAsync(dst, src, cnt * 8, hipMemcpyDeviceToHost, h->stream); };
    HIPCHECK(h, dn(pose.data(), h->batch.pose + (size_t)first * 77, (size_t)n * 77));
    HIPCHECK(h, dn(sb.data(), h->batch.sb + (size_t)first * 99, (size_t)n * 99));
    HIPCHECK(h, dn(feat.data(), h->batch.feat + (size_t)first * sF, (size_t)n * sF));
    HIPCHECK(h, dn(Ps.data(), h->batch.out_Ps + (size_t)first * 33, (size_t)n * 33));
    HIPCHECK(h, dn(Rs.data(), h->batch.out_Rs + (size_t)first * 99, (size_t)n * 99));
    HIPCHECK(h, dn(Vs.data(), h->batch.out_Vs + (size_t)first * 33, (size_t)n * 33));
    HIPCHECK(h, dn(Bas.data(), h->batch.out_Bas + (size_t)first * 33, (size_t)n * 33));
    HIPCHECK(h, dn(Bgs.data(), h->batch.out_Bgs + (size_t)first * 33, (size_t)n * 33));
    std::vector<vilf_summary> sums(n);
    int rc = vilf_batch_summaries(h, first, n, sums.data());   // synchronises the stream
    if (rc != VILF_OK) return rc;
    for (int i = 0; i < n; i++) {
        vilf_window_out &o = outs[i];
        const int F = h->h_nfeat[first + i];
        if (o.para_pose) std::memcpy(o.para_pose, &pose[(size_t)i * 77], 77 * 8);
        if (o.para_speed_bias) std::memcpy(o.para_speed_bias, &sb[(size_t)i * 99], 99 * 8);
        if (o.para_feature && F) std::memcpy(o.para_feature, &feat[(size_t)i * sF], (size_t)F * 8);
        if (o.Ps) std::memcpy(o.Ps, &Ps[(size_t)i * 33], 33 * 8);
        if (o.Rs) std::memcpy(o.Rs, &Rs[(size_t)i * 99], 99 * 8);
        if (o.Vs) std::memcpy(o.Vs, &Vs[(size_t)i * 33], 33 * 8);
        if (o.Bas) std::memcpy(o.Bas, &Bas[(size_t)i * 33], 33 * 8);
        if (o.Bgs) std::memcpy(o.Bgs, &Bgs[(size_t)i * 33], 33 * 8);
        const double *exw = &h->h_ex[(size_t)(first + i) * 7];
        for (int k = 0; k < 3; k++) o.tic[k] = exw[k];
        quat_to_R(exw + 3, o.ric);
        o.td = h->h_td[first + i];                      // kept current by the general path (estimate_td)
        o.summary = sums[i];
    }
    return VILF_OK;
}

// The estimator's outputs only (double2vector(): Ps / Rs / Vs / Bas / Bgs, estimator.cpp:549-638) plus the summaries, into caller-owned contiguous arrays
// [n][33] / [n][99] / ...: five device-to-host copies through pinned staging and one wait — what a per-frame caller needs back; the parameter arrays
// (para_Pose, para_Feature ...) stay on the device for the marginalization. Any pointer may be NULL.
extern "C" int vilf_batch_download_states(vilf_handle *h, int first, int n, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, vilf_summary *sums) {
    if (!h || !h->resident || first < 0 || n < 0 || first + n > h->B) return VILF_ERR_INVALID_ARGUMENT;
    if (n == 0) return VILF_OK;
    const size_t sn = n, per = 33 + 99 + 33 + 33 + 33;
    if (!h->pin_down.ensure(sn * per * 8 + sn * sizeof(VbState) + 256)) { h->err = "hipHostMalloc failed (download staging)"; return VILF_ERR_DEVICE; }
    double *st = static_cast<double *>(h->pin_down.p);
    double *pP = st, *pR = pP + sn * 33, *pV = pR + sn * 99, *pA = pV + sn * 33, *pG = pA + sn * 33;
    VbState *pS = reinterpret_cast<VbState *>(pG + sn * 33);
    auto dn = [&](void *dst, const void *src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream); };
    if (Ps) HIPCHECK(h, dn(pP, h->batch.out_Ps + (size_t)first * 33, sn * 33 * 8));
    if (Rs) HIPCHECK(h, dn(pR, h->batch.out_Rs + (size_t)first * 99, sn * 99 * 8));
    if (Vs) HIPCHECK(h, dn(pV, h->batch.out_Vs + (size_t)first * 33, sn * 33 * 8));
    if (Bas) HIPCHECK(h, dn(pA, h->batch.out_Bas + (size_t)first * 33, sn * 33 * 8));
    if (Bgs) HIPCHECK(h, dn(pG, h->batch.out_Bgs + (size_t)first * 33, sn * 33 * 8));
    if (sums) HIPCHECK(h, dn(pS, h->batch.st + first, sn * sizeof(VbState)));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    solve_time_resolve(h);
    if (Ps) std::memcpy(Ps, pP, sn * 33 * 8);
    if (Rs) std::memcpy(Rs, pR, sn * 99 * 8);
    if (Vs) std::memcpy(Vs, pV, sn * 33 * 8);
    if (Bas) std::memcpy(Bas, pA, sn * 33 * 8);
    if (Bgs) std::memcpy(Bgs, pG, sn * 33 * 8);
    if (sums) { const int rce = vb_check_dev_error(h, pS, first, n); if (rce != VILF_OK) return rce; }
    if (sums) for (int i = 0; i < n; i++) {
        sums[i].num_iterations = pS[i].iteration; sums[i].num_successful_steps = pS[i].num_successful; sums[i].num_linear_solves = pS[i].num_linear_solves;
        sums[i].termination = pS[i].termination; sums[i].initial_cost = pS[i].initial_cost; sums[i].final_cost = pS[i].x_cost; sums[i].final_radius = pS[i].radius;
        sums[i].usec_solve = h->last_solve_usec;
    }
    return VILF_OK;
}

// n independent windows of sizes other than the reference's WINDOW_SIZE + 1 = 11 frames, solved side by side in one chain of launches (vilf_lw.hip). 11-frame
// windows belong to the vilf_batch_* entry points (LDS kernels, priors, marginalization).
extern "C" int vilf_window_solve_group(vilf_handle *h, int n, const vilf_window_in *in, vilf_window_out *out) {
    if (!h || n < 1 || !in || !out) return VILF_ERR_INVALID_ARGUMENT;
    std::vector<const vilf_window_in *> inp(n);
    std::vector<vilf_window_out *> outp(n);
    for (int i = 0; i < n; i++) {
        if (in[i].n_frames == VB_NF) { h->err = "vilf_window_solve_group: 11-frame windows go through vilf_batch_upload / vilf_batch_solve"; return VILF_ERR_UNSUPPORTED; }
        if (!out[i].Ps || !out[i].Rs || !out[i].Vs || !out[i].Bas || !out[i].Bgs) return VILF_ERR_INVALID_ARGUMENT;
        inp[i] = &in[i]; outp[i] = &out[i];
    }
    return vilf_lw_group_solve(h, n, inp.data(), outp.data(), nullptr);
}
extern "C" int vilf_window_solve(vilf_handle *h, const vilf_window_in *in, vilf_window_out *out) {
    if (!h || !in || !out) return VILF_ERR_INVALID_ARGUMENT;
    if (in->n_frames != VB_NF) {          // not the reference's WINDOW_SIZE = 10: the general path (one window spread over the device, no prior)
        if (in->n_frames != h->opts.window_size + 1) { h->err = "n_frames must be options.window_size + 1"; return VILF_ERR_INVALID_ARGUMENT; }
        return vilf_lw_window_solve(h, in, out, 0);
    }
    auto t0 = std::chrono::steady_clock::now();
    h->defer_upload_sync = true;
    int rc = vilf_batch_upload(h, 1, in);
    h->defer_upload_sync = false;
    if (rc != VILF_OK) return rc;
    if (h->opts.estimate_extrinsic || h->opts.estimate_td)      // Ex_Pose / td as variables (estimator.cpp:701-717): the general single-window path, with the slot-0 prior
        return vilf_lw_window_solve(h, in, out, 1);      // slot 0
    rc = vilf_batch_solve(h, 1);
    if (rc != VILF_OK) return rc;
    rc = vilf_batch_download(h, 0, 1, out);
    if (rc != VILF_OK) return rc;
    out->summary.usec_solve = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    return (out->summary.termination == VILF_TERM_FAILURE) ? VILF_SOLVER_ABNORMAL : VILF_OK;
}

extern "C" int vilf_prior_import(vilf_handle *h, int slot, const vilf_prior *prior) {
    if (!h || slot < 0 || !prior) return VILF_ERR_INVALID_ARGUMENT;
    if (prior->valid && (prior->n <= 0 || prior->n > VILF_PRIOR_MAX_DIM || prior->n_blocks <= 0 || prior->n_blocks > VILF_PRIOR_MAX_BLOCKS)) return VILF_ERR_INVALID_ARGUMENT;
    if (prior->valid) {          // block table: global sizes 7 (pose), 9 (speed-bias) or 1 (td), local columns inside [0, n) — the kernels index LDS vectors with it
        for (int i = 0; i < prior->n_blocks; i++) {
            const int gs = prior->block_size[i], ls = gs == 7 ? 6 : gs, idx = prior->block_idx[i];
            if ((gs != 7 && gs != 9 && gs != 1) || prior->block_id[i] < 0 || idx < 0 || idx + ls > prior->n) { h->err = "prior block table out of range"; return VILF_ERR_INVALID_ARGUMENT; }
        }
    }
    if ((int)h->priors.size() <= slot) { vilf_prior z; std::memset(&z, 0, sizeof(z)); h->priors.resize(slot + 1, z); h->prior_dirty.resize(slot + 1, 1); }
    h->priors[slot] = *prior;
    h->prior_dirty[slot] = 1;
    if ((int)h->prior_dev_newer.size() <= slot) h->prior_dev_newer.resize(slot + 1, 0);
    h->prior_dev_newer[slot] = 0;
    return VILF_OK;
}

extern "C" int vilf_prior_export(vilf_handle *h, int slot, vilf_prior *out) {
    if (!h || slot < 0 || slot >= (int)h->priors.size() || !out) return VILF_ERR_INVALID_ARGUMENT;
    { int rcp = pull_device_priors(h); if (rcp != VILF_OK) return rcp; }
    *out = h->priors[slot];
    return VILF_OK;
}

extern "C" int vilf_batch_marginalize(vilf_handle *h, int sync) {
    if (!h || !h->resident) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    { bool dirty = false; for (int w = 0; w < h->B; w++) if (h->prior_dirty[w]) dirty = true; if (dirty) { int rc = upload_priors(h); if (rc != VILF_OK) return rc; } }
    const size_t sB = h->B, sF = h->batch.Fmax, sC = h->batch.FACmax, M = h->mg_Mcap;
    // the exact (Jacobi) fallback's workspace — rotation log, Amm, X, eigenvalues — is a pool of slots, not one per window: the log alone is 24 (M - 1) M doubles
    // (20 MB at 300 dropped features). The pool is as large as 8 GB allow; flagged windows beyond it are taken by further launches of the exact pass.
    const size_t slot_bytes = (MG_SWEEPS * (M - 1) * M + (M > MG_MLDS ? M * M : 0) + M * (MG_NK + 1) + M) * 8;
    size_t sPool = std::max<size_t>(1, std::min<size_t>(sB, ((size_t)8 << 30) / std::max<size_t>(slot_bytes, 1)));
    if (const char *e = std::getenv("VILF_MARG_POOL")) sPool = std::max<size_t>(1, std::min<size_t>(sB, (size_t)std::atoi(e)));      // test hook: several rounds on a small batch
    struct Req { int id; size_t bytes; };
    const Req reqs[] = {
        {D_MINFO, sB * MG_INFO * 4}, {D_MF0, sB * sF * 4}, {D_MSTP, sB * 77 * 8}, {D_MSTS, sB * 99 * 8}, {D_MSTF, sB * sF * 8}, {D_MSTE, sB * 7 * 8},
        {D_MBUF, sB * MG_MROW * sC * 8}, {D_MHD, sB * MG_ND * MG_ND * 8}, {D_MGD, sB * MG_ND * 8}, {D_MWF, sB * sF * MG_ND * 8}, {D_MHF, sB * sF * 8},
        {D_MGF, sB * sF * 8}, {D_MAMM, (M > MG_MLDS ? sPool * M * M * 8 : 8)}, {D_MX, sPool * M * (MG_NK + 1) * 8}, {D_MROT, sPool * MG_SWEEPS * (M - 1) * M * 8},
        {D_MLAM, sPool * M * 8}, {D_MAR, sB * MG_NK * MG_NK * 8}, {D_MBR, sB * MG_NK * 8},
        {D_QLV, sB * MG_NK * (MG_NK + 1) * 8}, {D_QLD, sB * 2 * (MG_NK + 2) * 8}, {D_QLLOG, sB * 2 * QL_RCAP * 8}, {D_QLIT, sB * QL_ICAP * 4}, {D_QLINFO, sB * 4 * 4},
    };
    for (const Req &r : reqs) if (!h->d[r.id].ensure(r.bytes)) { h->err = "hipMalloc failed (marginalization workspace)"; return VILF_ERR_DEVICE; }
    VbMarg &g = h->marg;
    g.Mcap = (int)M; g.init_depth = h->opts.init_depth;
    g.mflag = h->d[D_MFLAG].as<int>(); g.info = h->d[D_MINFO].as<int>(); g.f0rank = h->d[D_MF0].as<int>();
    g.st_pose = h->d[D_MSTP].as<double>(); g.st_sb = h->d[D_MSTS].as<double>(); g.st_feat = h->d[D_MSTF].as<double>(); g.st_ex = h->d[D_MSTE].as<double>();
    g.Mbuf = h->d[D_MBUF].as<double>(); g.Hd = h->d[D_MHD].as<double>(); g.gd = h->d[D_MGD].as<double>(); g.Wf = h->d[D_MWF].as<double>();
    g.hfm = h->d[D_MHF].as<double>(); g.gfm = h->d[D_MGF].as<double>(); g.Amm = h->d[D_MAMM].as<double>(); g.X = h->d[D_MX].as<double>();
    g.rot = h->d[D_MROT].as<double>(); g.lam = h->d[D_MLAM].as<double>(); g.Ar = h->d[D_MAR].as<double>(); g.br = h->d[D_MBR].as<double>();
    g.qlV = h->d[D_QLV].as<double>(); g.qlD = h->d[D_QLD].as<double>(); g.qlLog = h->d[D_QLLOG].as<double>(); g.qlIt = h->d[D_QLIT].as<int>(); g.qlInfo = h->d[D_QLINFO].as<int>();
    // Two sets of prior buffers. When the live set is still the one the windows were uploaded / rewound with, the new priors go to the other set and the sets swap
    // afterwards: the uploaded priors stay intact for vilf_batch_rewind at no cost (this used to be a 1.7 GB device copy per marginalization and another per rewind).
    // A second marginalization without a rewind in between writes in place, as before, so that the snapshot survives.
    const int live[6] = {D_PHDR, D_PX0, D_PJ, D_PR, D_PH, D_PG}, bak[6] = {D_PHDR0, D_PX00, D_PJ0, D_PR0, D_PH0, D_PG0};
    const size_t pbytes[6] = {sB * VB_PRIOR_HDR * 4, sB * 24 * 9 * 8, sB * VB_PRIOR_LD * VB_PRIOR_LD * 8, sB * VB_PRIOR_LD * 8, sB * VB_PRIOR_LD * VB_PRIOR_LD * 8, sB * VB_PRIOR_LD * 8};
    const bool to_other_set = !h->prior_restore_needed;
    if (to_other_set) for (int k = 0; k < 6; k++) if (!h->d[bak[k]].ensure(pbytes[k])) { h->err = "hipMalloc failed (second prior set)"; return VILF_ERR_DEVICE; }
    const int *oset = to_other_set ? bak : live;
    g.prior_hdr_out = h->d[oset[0]].as<int>(); g.prior_x0_out = h->d[oset[1]].as<double>(); g.prior_J_out = h->d[oset[2]].as<double>(); g.prior_r_out = h->d[oset[3]].as<double>();
    g.prior_H_out = h->d[oset[4]].as<double>(); g.prior_g_out = h->d[oset[5]].as<double>();
    const dim3 grid(h->B), block(VB_NT);
    const bool prof = h->profiling != 0;
    hipEvent_t mev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (prof) mev[0] = vilf_prof_event(h);
    if (h->batch.est_td) hipLaunchKernelGGL(k_marg_prepare_td, grid, block, 0, h->stream, h->batch, g);
    else hipLaunchKernelGGL(k_marg_prepare, grid, block, 0, h->stream, h->batch, g);
    if (prof) mev[1] = vilf_prof_event(h);
    hipLaunchKernelGGL(k_marg_schur, grid, block, (size_t)(MG_MD * MG_MD + MG_MD * (MG_NK + 1) + 1000 + VB_NT + MG_FCH * MG_RWP) * sizeof(double), h->stream, h->batch, g,
                       std::getenv("VILF_MARG_FORCE_EXACT") ? 2 : 0);      // test hook: exercise the Jacobi path on well-conditioned windows too
    g.pool = (int)sPool;
    for (int r = 0; r < (int)((sB + sPool - 1) / sPool); r++) {      // one launch unless the pool is smaller than the batch (large Mcap)
        g.pool_round = r;
        hipLaunchKernelGGL(k_marg_schur, grid, block, h->marg_lds_schur, h->stream, h->batch, g, 1);
    }
    g.pool_round = 0;
    if (prof) mev[2] = vilf_prof_event(h);
    // kept block: the Cholesky form of the new prior where the reference's 1e-8 truncation provably removes nothing (k_mf_chol; it marks the windows it has
    // finished, the launches below skip those), otherwise the
    // eigen-solver of the kept block in three launches (tred2 per workgroup, the QL recurrence of every window one lane each, rotation replay +
    // prior output per workgroup); k_marg_finish (everything in one workgroup) only takes windows whose rotation log overflowed
    // The kept-block kernels come in two LDS sizes by dimension class (n < 78: three workgroups per CU) — for a large batch. A small one (the single window of the real-time
    // case) does not fill the device either way and takes ONE launch of each with the full LDS: four launches less in a chain of mostly empty ones (~5 us each).
    const bool one_class = h->B <= 64;
    const size_t lds_small = (size_t)77 * 77 * sizeof(double);
    {
        const int no_chol = std::getenv("VILF_MARG_NO_CHOL") ? 1 : 0;         // test hook: the eigen-solver for every window
        // n <= 75 (every prior the reference produces without td): the augmented factorisation on the matrix cores (k_mf_chol_tiles, 34 KB of LDS); wider kept blocks:
        // the column-by-column kernel. VILF_MARG_CHOL_COLUMNS=1: the column kernel for every size (tests compare the two forms).
        const bool tiles = !std::getenv("VILF_MARG_CHOL_COLUMNS");
        const int lo = tiles ? SB_ND + 1 : 0;
        if (tiles) hipLaunchKernelGGL(k_mf_chol_tiles, grid, block, (size_t)(SB_NR * (SB_NR + 1) / 2 + 2 * 4 * 160 + 16) * sizeof(double), h->stream, h->batch, g, no_chol);
        if (one_class) hipLaunchKernelGGL(k_mf_chol, grid, block, h->marg_lds_finish, h->stream, h->batch, g, lo, 1 << 30, no_chol);
        else {
            hipLaunchKernelGGL(k_mf_chol, grid, block, lds_small, h->stream, h->batch, g, lo, 78, no_chol);
            hipLaunchKernelGGL(k_mf_chol, grid, block, h->marg_lds_finish, h->stream, h->batch, g, 78, 1 << 30, no_chol);
        }
    }
    if (one_class) hipLaunchKernelGGL(k_mf_tridiag, grid, block, h->marg_lds_finish, h->stream, h->batch, g, 0, 1 << 30);
    else {
        hipLaunchKernelGGL(k_mf_tridiag, grid, block, lds_small, h->stream, h->batch, g, 0, 78);
        hipLaunchKernelGGL(k_mf_tridiag, grid, block, h->marg_lds_finish, h->stream, h->batch, g, 78, 1 << 30);
    }
    hipLaunchKernelGGL(k_mf_ql, dim3((h->B + QL_LPW - 1) / QL_LPW), dim3(64), (size_t)2 * (MG_NK + 2) * QL_LPW * sizeof(double), h->stream, h->batch, g,
                       std::getenv("VILF_MARG_FORCE_QL_FALLBACK") ? 1 : 0);        // test hook
    if (one_class) {
        hipLaunchKernelGGL(k_mf_apply, grid, block, h->marg_lds_finish + VILF_MFA_LDS_EXTRA, h->stream, h->batch, g, 0, 1 << 30);
        hipLaunchKernelGGL(k_marg_finish, grid, block, h->marg_lds_finish, h->stream, h->batch, g, 0, 1 << 30, 1);
    } else {
        hipLaunchKernelGGL(k_mf_apply, grid, block, lds_small + VILF_MFA_LDS_EXTRA, h->stream, h->batch, g, 0, 78);
        hipLaunchKernelGGL(k_mf_apply, grid, block, h->marg_lds_finish + VILF_MFA_LDS_EXTRA, h->stream, h->batch, g, 78, 1 << 30);
        hipLaunchKernelGGL(k_marg_finish, grid, block, lds_small, h->stream, h->batch, g, 0, 78, 1);
        hipLaunchKernelGGL(k_marg_finish, grid, block, h->marg_lds_finish, h->stream, h->batch, g, 78, 1 << 30, 1);
    }
    if (to_other_set) {
        hipLaunchKernelGGL(k_prior_keep, grid, block, 0, h->stream, h->batch, g);
        for (int k = 0; k < 6; k++) std::swap(h->d[live[k]], h->d[bak[k]]);
        bind_prior_pointers(h);
        h->prior_backup_valid = true;           // the other set now holds the priors as uploaded
    }
    if (prof) mev[3] = vilf_prof_event(h);
    hipLaunchKernelGGL(k_prior_prep, grid, block, VILF_PRIOR_PREP_LDS, h->stream, h->batch, h->d[D_PH].as<double>(), h->d[D_PG].as<double>(), (unsigned)VILF_PRIOR_PREP_LDS, (const int *)g.qlInfo);
    if (prof) mev[4] = vilf_prof_event(h);
    HIPCHECK(h, hipGetLastError());
    if (prof) {
        for (int k = 0; k < 4; k++) vilf_prof_span(h, mev[k], mev[k + 1], &h->marg_ms[k], &h->marg_launches[k]);
        if (sync) { const int rcf = vilf_prof_flush(h); if (rcf != VILF_OK) return rcf; }
    }
    for (int w = 0; w < h->B; w++) { h->prior_dev_newer[w] = 1; h->prior_dirty[w] = 0; }
    h->prior_restore_needed = true;
    if (sync) {
        std::vector<int> info(sB * MG_INFO);
        HIPCHECK(h, hipMemcpyAsync(info.data(), g.info, info.size() * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(h, hipStreamSynchronize(h->stream));
        for (int w = 0; w < h->B; w++) {
            const int st = info[(size_t)w * MG_INFO];
            if (st == 2) h->prior_dev_newer[w] = 0;     // SECOND_NEW without Pose[WINDOW_SIZE-1] in the prior: prior unchanged (estimator.cpp:982-983)
            if (st == 3) { h->err = "marginalization: block table / dimension outside the supported range"; return VILF_ERR_UNSUPPORTED; }
        }
    }
    return VILF_OK;
}
extern "C" int vilf_window_marginalize(vilf_handle *h) { return vilf_batch_marginalize(h, 1); }

// which path every window of the last vilf_batch_marginalize took: counts[0] windows that produced a new prior; [1] of those, dropped block Amm by the arrow
// Cholesky (the rest: Jacobi eigen-decomposition with the 1e-8 pseudo-inverse); [2] kept block by Cholesky (J0 = L^T; the rest: tred2 / tql2 eigen-solver);
// [3] windows whose prior was left unchanged (SECOND_NEW without Pose[WINDOW_SIZE - 1] in it) or unsupported
extern "C" int vilf_batch_marginalize_stats(vilf_handle *h, int counts[4]) {
    if (!h || !counts || !h->resident || !h->marg.info || !h->marg.qlInfo) return VILF_ERR_INVALID_ARGUMENT;
    HIPCHECK(h, hipSetDevice(h->device));
    const size_t sB = h->B;
    std::vector<int> info(sB * MG_INFO), qi(sB * 4);
    HIPCHECK(h, hipMemcpyAsync(info.data(), h->marg.info, info.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipMemcpyAsync(qi.data(), h->marg.qlInfo, qi.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    counts[0] = counts[1] = counts[2] = counts[3] = 0;
    for (size_t w = 0; w < sB; w++) {
        if (info[w * MG_INFO] != 0) { counts[3]++; continue; }
        counts[0]++;
        if (info[w * MG_INFO + 7] == 0) counts[1]++;
        if (qi[w * 4 + 3] == 1) counts[2]++;
    }
    return VILF_OK;
}

// ---- Ceres-layout hooks --------------------------------------------------------------------------------------
static int hook_buf(vilf_handle *h, size_t doubles) { return h->d[D_HOOK].ensure(doubles * 8) ? VILF_OK : VILF_ERR_DEVICE; }

extern "C" int vilf_eval_projection(vilf_handle *h, const double *const *p, const double pts_i[3], const double pts_j[3], double *residuals, double **jac) {
    if (!h || !p || !residuals) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 256) != VILF_OK) return VILF_ERR_DEVICE;
    double in[32];
    std::memcpy(in, p[0], 56); std::memcpy(in + 7, p[1], 56); std::memcpy(in + 14, p[2], 56);
    std::memcpy(in + 21, pts_i, 24); std::memcpy(in + 24, pts_j, 24);
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_projection, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 14, p[3][0], d + 21, d + 24, h->opts.focal_length / 1.5, d + 32);
    double out[28];
    HIPCHECK(h, hipMemcpyAsync(out, d + 32, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    residuals[0] = out[0]; residuals[1] = out[1];
    if (jac) {
        for (int blk = 0; blk < 2; blk++)
            if (jac[blk]) for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) jac[blk][7 * r + c] = out[2 + 12 * blk + 6 * r + c]; jac[blk][7 * r + 6] = 0; }
        if (jac[2]) {                               // Ex_Pose block (projection_factor.cpp:97-104): the device routine of the general path (ProjectionTdFactor with td = td_i = td_j, zero velocity
                                                    // = ProjectionFactor); its residual and other blocks are the same function of the same inputs
            double in2[40] = {0};
            std::memcpy(in2, in, 27 * 8);
            in2[31] = p[3][0]; in2[37] = 0.0; in2[38] = h->opts.focal_length / 1.5;
            HIPCHECK(h, hipMemcpyAsync(d + 64, in2, sizeof(in2), hipMemcpyHostToDevice, h->stream));
            hipLaunchKernelGGL(k_hook_projection_td, dim3(1), dim3(64), 0, h->stream, d + 64, d + 64 + 40);
            double out2[42];
            HIPCHECK(h, hipMemcpyAsync(out2, d + 64 + 40, sizeof(out2), hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(h, hipStreamSynchronize(h->stream));
            for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) jac[2][7 * r + c] = out2[26 + 6 * r + c]; jac[2][7 * r + 6] = 0; }
        }
        if (jac[3]) { jac[3][0] = out[26]; jac[3][1] = out[27]; }
    }
    return VILF_OK;
}

extern "C" int vilf_eval_projection_td(vilf_handle *h, const double *const *p, const double pts_i[3], const double pts_j[3], const double vel_i[2], const double vel_j[2],
                                       double td_i, double td_j, double row_i, double row_j, double *residuals, double **jac) {
    if (!h || !p || !residuals || !pts_i || !pts_j || !vel_i || !vel_j) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 128) != VILF_OK) return VILF_ERR_DEVICE;
    double in[40];
    std::memcpy(in, p[0], 56); std::memcpy(in + 7, p[1], 56); std::memcpy(in + 14, p[2], 56);
    std::memcpy(in + 21, pts_i, 24); std::memcpy(in + 24, pts_j, 24); std::memcpy(in + 27, vel_i, 16); std::memcpy(in + 29, vel_j, 16);
    const double ROW = h->opts.ROW;
    in[31] = p[3][0]; in[32] = p[4][0]; in[33] = td_i; in[34] = td_j; in[35] = row_i - ROW / 2; in[36] = row_j - ROW / 2;      // projection_td_factor.cpp:19-20
    in[37] = h->opts.TR / ROW; in[38] = h->opts.focal_length / 1.5;
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_projection_td, dim3(1), dim3(64), 0, h->stream, d, d + 40);
    double out[42];
    HIPCHECK(h, hipMemcpyAsync(out, d + 40, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    residuals[0] = out[0]; residuals[1] = out[1];
    if (jac) {
        const int off[3] = {2, 14, 26};
        for (int blk = 0; blk < 3; blk++)
            if (jac[blk]) for (int r = 0; r < 2; r++) { for (int c = 0; c < 6; c++) jac[blk][7 * r + c] = out[off[blk] + 6 * r + c]; jac[blk][7 * r + 6] = 0; }
        if (jac[3]) { jac[3][0] = out[38]; jac[3][1] = out[39]; }
        if (jac[4]) { jac[4][0] = out[40]; jac[4][1] = out[41]; }
    }
    return VILF_OK;
}

static void pack_imu_rec(const vilf_imu_preint *p, double *rec) {
    std::memset(rec, 0, IMU_REC * 8);
    rec[0] = p->sum_dt;
    for (int i = 0; i < 3; i++) { rec[1 + i] = p->delta_p[i]; rec[8 + i] = p->delta_v[i]; rec[11 + i] = p->linearized_ba[i]; rec[14 + i] = p->linearized_bg[i]; }
    for (int i = 0; i < 4; i++) rec[4 + i] = p->delta_q[i];
    auto blk = [&](int off, int r0, int c0) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) rec[off + 3 * i + j] = p->jacobian[(r0 + i) * 15 + c0 + j]; };
    blk(17, 0, 9); blk(26, 0, 12); blk(35, 3, 12); blk(44, 6, 9); blk(53, 6, 12);
    rec[287] = 1.0;
}

extern "C" int vilf_eval_imu(vilf_handle *h, const double *const *p, const vilf_imu_preint *pre, double *residuals, double **jac) {
    if (!h || !p || !pre || !residuals) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 4096) != VILF_OK) return VILF_ERR_DEVICE;
    double *d = h->d[D_HOOK].as<double>();
    std::vector<double> in(40 + IMU_REC + 225, 0.0);
    std::memcpy(&in[0], p[0], 56); std::memcpy(&in[7], p[1], 72); std::memcpy(&in[16], p[2], 56); std::memcpy(&in[23], p[3], 72);
    for (int i = 0; i < 3; i++) in[32 + i] = h->opts.G[i];
    pack_imu_rec(pre, &in[40]);
    std::memcpy(&in[40 + IMU_REC], pre->covariance, 225 * 8);
    HIPCHECK(h, hipMemcpyAsync(d, in.data(), in.size() * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_imu_prep, dim3(1), dim3(64), 0, h->stream, 1, d + 40 + IMU_REC, d + 1024, d + 40);
    hipLaunchKernelGGL(k_hook_imu, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 16, d + 23, d + 40, d + 32, d + 2048, d + 3072);
    std::vector<double> out(15 + 450);
    HIPCHECK(h, hipMemcpyAsync(out.data(), d + 2048, out.size() * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 15; i++) residuals[i] = out[i];
    if (jac) {
        const int off[4] = {0, 6, 15, 21}, loc[4] = {6, 9, 6, 9}, glob[4] = {7, 9, 7, 9};
        for (int blk = 0; blk < 4; blk++)
            if (jac[blk]) for (int r = 0; r < 15; r++) { for (int c = 0; c < glob[blk]; c++) jac[blk][glob[blk] * r + c] = (c < loc[blk]) ? out[15 + 30 * r + off[blk] + c] : 0.0; }
    }
    return VILF_OK;
}

// the parts of IMUFactor::Evaluate on their own: residual / Jacobians BEFORE the multiplication by sqrt_info (Ceres layout as above), and the 15 x 15 sqrt_info the device
// computes once per upload (k_imu_prep) — the tests compare each tightly instead of only their ill-conditioned product
extern "C" int vilf_eval_imu_raw(vilf_handle *h, const double *const *p, const vilf_imu_preint *pre, double *residuals, double **jac, double *sqrt_info_out) {
    if (!h || !p || !pre || !residuals) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 4096) != VILF_OK) return VILF_ERR_DEVICE;
    double *d = h->d[D_HOOK].as<double>();
    std::vector<double> in(40 + IMU_REC + 225, 0.0);
    std::memcpy(&in[0], p[0], 56); std::memcpy(&in[7], p[1], 72); std::memcpy(&in[16], p[2], 56); std::memcpy(&in[23], p[3], 72);
    for (int i = 0; i < 3; i++) in[32 + i] = h->opts.G[i];
    pack_imu_rec(pre, &in[40]);
    std::memcpy(&in[40 + IMU_REC], pre->covariance, 225 * 8);
    HIPCHECK(h, hipMemcpyAsync(d, in.data(), in.size() * 8, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_imu_prep, dim3(1), dim3(64), 0, h->stream, 1, d + 40 + IMU_REC, d + 1024, d + 40);
    hipLaunchKernelGGL(k_hook_imu, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 16, d + 23, d + 40, d + 32, d + 2048, d + 3072);
    std::vector<double> raw(450 + 15), S(225);
    HIPCHECK(h, hipMemcpyAsync(raw.data(), d + 3072, raw.size() * 8, hipMemcpyDeviceToHost, h->stream));       // scratch of k_hook_imu: J_raw [15 x 30], r_raw [15]
    HIPCHECK(h, hipMemcpyAsync(S.data(), d + 40 + IMU_SQRT, 225 * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 15; i++) residuals[i] = raw[450 + i];
    if (jac) {
        const int off[4] = {0, 6, 15, 21}, loc[4] = {6, 9, 6, 9}, glob[4] = {7, 9, 7, 9};
        for (int blk = 0; blk < 4; blk++)
            if (jac[blk]) for (int r = 0; r < 15; r++) { for (int c = 0; c < glob[blk]; c++) jac[blk][glob[blk] * r + c] = (c < loc[blk]) ? raw[30 * r + off[blk] + c] : 0.0; }
    }
    if (sqrt_info_out) std::memcpy(sqrt_info_out, S.data(), 225 * 8);
    return VILF_OK;
}

extern "C" int vilf_eval_lidar_between(vilf_handle *h, const double *const *p, const vilf_lidar_constraint *c, double *residuals, double **jac) {
    if (!h || !p || !c || !residuals) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 256) != VILF_OK) return VILF_ERR_DEVICE;
    if (!h->resident) {   // qil / til are derived at upload; derive here too
        const vilf_options &o = h->opts; VbBatch &b = h->batch;
        double M[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[3 * i + j] = o.RIC[3 * i] * o.RCL[j] + o.RIC[3 * i + 1] * o.RCL[3 + j] + o.RIC[3 * i + 2] * o.RCL[6 + j];
        quat_from_R(M, b.qil);
        for (int i = 0; i < 3; i++) b.til[i] = o.RIC[3 * i] * o.TCL[0] + o.RIC[3 * i + 1] * o.TCL[1] + o.RIC[3 * i + 2] * o.TCL[2] + o.TIC[i];
    }
    double in[32];
    std::memcpy(in, p[0], 56); std::memcpy(in + 7, p[1], 56);
    std::memcpy(in + 14, h->batch.qil, 32); std::memcpy(in + 18, h->batch.til, 24);
    for (int i = 0; i < 4; i++) in[21 + i] = c->q[i];
    for (int i = 0; i < 3; i++) in[25 + i] = c->t[i];
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_lidar, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 14, d + 18, d + 21, d + 32);
    double out[78];
    HIPCHECK(h, hipMemcpyAsync(out, d + 32, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 6; i++) residuals[i] = out[i];
    if (jac)
        for (int blk = 0; blk < 2; blk++)
            if (jac[blk]) for (int r = 0; r < 6; r++) { for (int cc = 0; cc < 6; cc++) jac[blk][7 * r + cc] = out[6 + 36 * blk + 6 * r + cc]; jac[blk][7 * r + 6] = 0; }
    return VILF_OK;
}

extern "C" int vilf_eval_edge(vilf_handle *h, const double pose[7], const double cp[3], const double a[3], const double bb[3], double r[3], double *J) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 128) != VILF_OK) return VILF_ERR_DEVICE;
    double in[16];
    std::memcpy(in, pose, 56); std::memcpy(in + 7, cp, 24); std::memcpy(in + 10, a, 24); std::memcpy(in + 13, bb, 24);
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_edge, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 10, d + 13, d + 16);
    double out[21];
    HIPCHECK(h, hipMemcpyAsync(out, d + 16, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 3; i++) r[i] = out[i];
    if (J) for (int i = 0; i < 3; i++) { for (int c = 0; c < 6; c++) J[7 * i + c] = out[3 + 6 * i + c]; J[7 * i + 6] = 0; }
    return VILF_OK;
}

extern "C" int vilf_eval_surf(vilf_handle *h, const double pose[7], const double cp[3], const double n[3], double dd, double r[1], double *J) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 128) != VILF_OK) return VILF_ERR_DEVICE;
    double in[16];
    std::memcpy(in, pose, 56); std::memcpy(in + 7, cp, 24); std::memcpy(in + 10, n, 24);
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_surf, dim3(1), dim3(64), 0, h->stream, d, d + 7, d + 10, dd, d + 16);
    double out[7];
    HIPCHECK(h, hipMemcpyAsync(out, d + 16, sizeof(out), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    r[0] = out[0];
    if (J) { for (int c = 0; c < 6; c++) J[c] = out[1 + c]; J[6] = 0; }
    return VILF_OK;
}

static int plus_hook(vilf_handle *h, const double x[7], const double dl[6], double xp[7], int kind) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    if (hook_buf(h, 64) != VILF_OK) return VILF_ERR_DEVICE;
    double in[13];
    std::memcpy(in, x, 56); std::memcpy(in + 7, dl, 48);
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in, sizeof(in), hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_hook_plus, dim3(1), dim3(64), 0, h->stream, d, d + 7, kind, d + 16);
    HIPCHECK(h, hipMemcpyAsync(xp, d + 16, 56, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}
extern "C" int vilf_pose_plus(vilf_handle *h, const double x[7], const double d[6], double xp[7]) { return plus_hook(h, x, d, xp, 0); }
extern "C" int vilf_se3_plus(vilf_handle *h, const double x[7], const double d[6], double xp[7]) { return plus_hook(h, x, d, xp, 1); }

extern "C" __global__ void k_hook_prior(const int *hdr, const double *x0, const double *x, const double *J0, const double *r0, double *out);
extern "C" int vilf_eval_prior(vilf_handle *h, const vilf_prior *p, const double *const *params, double *residuals, double **jac) {
    if (!h || !p || !params || !residuals || !p->valid || p->n < 1 || p->n > VILF_PRIOR_MAX_DIM || p->n_blocks < 1 || p->n_blocks > VILF_PRIOR_MAX_BLOCKS) return VILF_ERR_INVALID_ARGUMENT;
    const int n = p->n, nb = p->n_blocks;
    const size_t doubles = 32 + 2 * 24 * 9 + (size_t)n * n + 2 * (size_t)n + 16;
    if (hook_buf(h, doubles) != VILF_OK) return VILF_ERR_DEVICE;
    std::vector<double> in(32 + 2 * 24 * 9 + (size_t)n * n + n, 0.0);
    int *hdr = reinterpret_cast<int *>(in.data());             // 64 ints in the first 32 doubles
    hdr[0] = n; hdr[1] = nb;
    for (int i = 0; i < nb; i++) {
        hdr[2 + i] = p->block_size[i]; hdr[26 + i] = p->block_idx[i];
        if (!params[i]) return VILF_ERR_INVALID_ARGUMENT;
        for (int k = 0; k < p->block_size[i] && k < 9; k++) { in[32 + 9 * i + k] = p->block_x0[i][k]; in[32 + 216 + 9 * i + k] = params[i][k]; }
    }
    std::memcpy(&in[32 + 432], p->linearized_jacobians, sizeof(double) * n * n);
    std::memcpy(&in[32 + 432 + (size_t)n * n], p->linearized_residuals, sizeof(double) * n);
    double *d = h->d[D_HOOK].as<double>();
    HIPCHECK(h, hipMemcpyAsync(d, in.data(), in.size() * 8, hipMemcpyHostToDevice, h->stream));
    double *d_out = d + in.size();
    hipLaunchKernelGGL(k_hook_prior, dim3(1), dim3(256), 0, h->stream, reinterpret_cast<const int *>(d), d + 32, d + 32 + 216, d + 32 + 432, d + 32 + 432 + (size_t)n * n, d_out);
    HIPCHECK(h, hipMemcpyAsync(residuals, d_out, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    if (jac)                                                   // jacobians[i] = J0[:, idx : idx + local] in global size (pose: 7th column 0), :364-376
        for (int i = 0; i < nb; i++) {
            if (!jac[i]) continue;
            const int gs = p->block_size[i], ls = gs == 7 ? 6 : gs, idx = p->block_idx[i];
            for (int r = 0; r < n; r++) for (int c = 0; c < gs; c++) jac[i][(size_t)r * gs + c] = c < ls ? p->linearized_jacobians[(size_t)r * n + idx + c] : 0.0;
        }
    return VILF_OK;
}

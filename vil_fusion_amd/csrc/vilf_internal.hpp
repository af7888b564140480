// vilf_internal.hpp — library-internal definitions shared by the host translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include <cstring>
#include "../../include/vilfusion.h"
#include "vilf_batch.hpp"

struct DBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 8 + 256;
        if (hipMalloc(&p, want) != hipSuccess) return false;
        cap = want;
        return true;
    }
    void release() { if (p) hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() { return reinterpret_cast<T *>(p); }
};

enum {
    D_NFEAT, D_NFAC, D_POSE, D_SB, D_FEAT, D_CPOSE, D_CSB, D_CFEAT, D_POSE0, D_SB0, D_FEAT0, D_EX, D_GR0, D_GP0,
    D_FSTART, D_FNOBS, D_FOBS0, D_FFAC0, D_FCONST, D_OBS, D_PSFEAT, D_PSOBS, D_PSSLOT, D_PAIROFF, D_IMU, D_LIDAR,
    D_PHDR, D_PX0, D_PJ, D_PR, D_PH, D_PG, D_FACW, D_HPP, D_W, D_HF, D_GF, D_IMUH, D_IMUG, D_LIDH, D_LIDG, D_G, D_DIAGH,
    D_SCALE, D_DIAG, D_GRAD, D_GN, D_ST, D_OPS, D_ORS, D_OVS, D_OBAS, D_OBGS, D_COV, D_WORK, D_HOOK, D_DBG, D_LUTI, D_LUTL, D_LUTV,
    D_MFLAG, D_MINFO, D_MF0, D_MSTP, D_MSTS, D_MSTF, D_MSTE, D_MBUF, D_MHD, D_MGD, D_MWF, D_MHF, D_MGF, D_MAMM, D_MX, D_MROT, D_MLAM, D_MAR, D_MBR, D_QLV, D_QLD, D_QLLOG, D_QLIT, D_QLINFO,
    D_PAIRD, D_FACREC, D_CF, D_STAMPS, D_LIVE, D_ITERQ, D_SBTAB, D_SPLITC, D_SPLITB, D_UPSTAGE, D_DNSTAGE, D_OBSV, D_OBSTD, D_OBSROW, D_TD, D_PHDR0, D_PX00, D_PJ0, D_PR0, D_PH0, D_PG0,      // priors as uploaded (restored by vilf_batch_rewind after a marginalization)
    D_COUNT
};

// pinned host staging kept across calls (vilf_batch_upload / download): no allocation, no zero fill, truly asynchronous copies
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    bool ensure(size_t bytes) {
        if (bytes <= cap) return true;
        if (p) hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        if (hipHostMalloc(&p, want, hipHostMallocDefault) != hipSuccess) return false;
        std::memset(p, 0, want);
        cap = want;
        return true;
    }
    void release() { if (p) hipHostFree(p); p = nullptr; cap = 0; }
};

struct S2B;
struct FeatCtx;
struct PgCtx;
struct LwCtx;

// deep copy of a window snapshot (estimate_extrinsic / estimate_td: the batched entry points run the general single-window solve per slot and need the inputs again)
struct OwnedWindow {
    vilf_window_in in;
    std::vector<double> pose, sb, feat, obs, vel, ctd, row, gR0, gP0;
    std::vector<uint8_t> fconst;
    std::vector<int32_t> fstart, foff;
    std::vector<vilf_imu_preint> imu;
    std::vector<vilf_lidar_constraint> lidar;
};

// class plan of one window's visual factors (class_plan, vilf_api.hip): computed by the upload's validation pass, used by its packer
struct WindowPlan { int cls[VB_NPAIR], cstart[VB_NPAIR], ccount[VB_NPAIR], nslot; };
struct vilf_handle {
    vilf_options opts;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    DBuf d[D_COUNT];
    VbBatch batch;
    int B = 0;
    bool resident = false;
    bool async_upload = false, upload_inflight = false;   // vilf_set_async_upload: vilf_batch_upload does not wait for its copies; the next upload of the handle does, before it touches the staging
    bool defer_upload_sync = false;   // vilf_window_solve: upload, solve and download are one call — the host waits once, at the download
    std::vector<vilf_prior> priors;          // per slot (host mirror)
    std::vector<char> prior_dirty;
    std::vector<char> prior_dense;           // slot's live prior (imported from the host) holds a speed-bias block other than SpeedBias[0]
    int prior_dense_count = 0;
    std::vector<char> prior_dev_newer;       // slot's prior was produced on the device (marginalize) and not yet mirrored
    std::vector<int> h_mflag;
    int prior_slots_valid = 0;               // the device prior arrays hold slots 0 .. prior_slots_valid-1 (survive a re-upload of the windows)
    bool prior_backup_valid = false;         // D_P*0 hold the priors as last uploaded
    bool prior_restore_needed = false;       // a marginalization has overwritten the device priors since that backup
    int mg_Mcap = 0;
    VbMarg marg;
    size_t marg_lds_schur = 0, marg_lds_finish = 0;
    std::vector<int> h_nfeat, h_nframes;
    std::vector<double> h_ex, h_td;
    std::vector<OwnedWindow> own;            // only with estimate_extrinsic / estimate_td
    PinBuf pin_up, pin_down;                 // host staging of the batched upload / download
    bool luts_ready = false;                 // static scatter / gather tables uploaded
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int profiling = 0;                       // per-kernel HIP-event timing of the solve launches
    // deferred profile spans: an asynchronous call (sync == 0) records its events and leaves them here; they are read at the next point that waits for the
    // stream anyway (a synchronous call, vilf_batch_summaries, vilf_get_profile*) — per-kernel times without a host round trip between the stages of a frame
    struct ProfSpan { hipEvent_t a, b; double *ms; long *cnt; };
    std::vector<ProfSpan> prof_pending;
    std::vector<hipEvent_t> prof_used, prof_free;
    std::vector<hipStream_t> split_streams; hipEvent_t split_ev = nullptr;    // VILF_SOLVE_SPLIT experiment: the batch in parts on their own streams
    hipEvent_t wait_ev = nullptr;            // vilf_wait_for
    void *stamp_pinned = nullptr; size_t stamp_cap = 0; hipEvent_t stamp_ev = nullptr;   // vilf_batch_newest_poses_device: pinned staging of the caller's stamps
    double kernel_ms[4] = {0, 0, 0, 0};      // linearize, solve, step, other (accumulated since last reset)
    long kernel_launches[4] = {0, 0, 0, 0};
    std::vector<hipEvent_t> s2m_ev;         // scan-to-map profiling (same switch): group of launches -> ms
    std::vector<int> s2m_groups;
    double marg_ms[4] = {0, 0, 0, 0};        // k_marg_prepare, k_marg_schur, k_marg_finish, k_prior_prep
    long marg_launches[4] = {0, 0, 0, 0};
    double s2m_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long s2m_launches[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<WindowPlan> plans;
    double last_solve_usec = 0;
    bool solve_time_pending = false;         // the last solve was enqueued with sync == 0: ev0 / ev1 are read by the next call that waits for the stream
    size_t solve_lds = 0, lin_lds = 0, solve_sb_lds = 0;
    int split_gen = 0;      // generation counter of the k_linearize_split launches of this handle
    int iter_slots = 512;   // resident workgroups of k_iter = workspace slots it uses (two per CU)
    bool solve_dense_fallback = false;       // a prior imported from the host holds a speed-bias block other than SpeedBias[0]: k_solve (dense) instead of k_solve_sb
    FeatCtx *feat = nullptr;                 // LiDAR feature extraction workspace (vilf_feat.hip)
    S2B *s2m = nullptr, *s2b = nullptr;      // scan-to-map state: single stream / batched streams (vilf_s2m.hip)
    PgCtx *pg = nullptr;                     // pose-graph workspace (vilf_pg.hip)
    LwCtx *lw = nullptr;                     // large-window solve workspace (vilf_lw.hip)
};

// a copy ordered on the handle's stream and waited for: the library's streams are non-blocking (they do not synchronise with the legacy default stream), so a plain
// hipMemcpy would neither wait for the kernels before it nor hold back the ones after it
static inline hipError_t vilf_copy_sync(vilf_handle *h, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, h->stream);
    return e != hipSuccess ? e : hipStreamSynchronize(h->stream);
}
#define HIPCHECK(h, call)                                                                                        \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) {                                                                                  \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                        \
            return VILF_ERR_DEVICE;                                                                              \
        }                                                                                                        \
    } while (0)


hipEvent_t vilf_prof_event(vilf_handle *h);                                 // an event recorded on the handle's stream now (from the pool)
void vilf_prof_span(vilf_handle *h, hipEvent_t a, hipEvent_t b, double *ms, long *cnt);   // elapsed(a, b) is added to *ms at the next flush
int vilf_prof_flush(vilf_handle *h);                                        // waits for the stream, reads every pending span
void vilf_s2m_release(vilf_handle *h);
void vilf_feat_release(vilf_handle *h);
void vilf_pg_release(vilf_handle *h);
void vilf_lw_release(vilf_handle *h);
int vilf_lw_chol_max_n();                                                      // largest n vilf_lw_chol_solve takes (its back substitution keeps the solution in LDS)
int vilf_lw_chol_solve(vilf_handle *h, int n, double *S, double *y, int *info);   // blocked Cholesky solve on the handle's stream (vilf_lw.hip): S = [(n + 1) x n] row-major, row n = rhs
int vilf_lw_group_solve(vilf_handle *h, int G, const vilf_window_in *const *ins, vilf_window_out *const *outs, const int *slot1);   // G windows in one chain of launches; slot1[g] = resident batch slot + 1 (0 / nullptr: not resident)
int vilf_lw_window_solve(vilf_handle *h, const vilf_window_in *in, vilf_window_out *out, int batch_slot1);   // batch_slot1 = resident slot + 1 (0: not resident)   // window sizes other than 10, estimate_extrinsic / estimate_td (vilf_lw.hip)

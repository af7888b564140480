// vilf_comm.hip — the only collective of the path: an RCCL all-gather of newest-frame poses [stamp x y z qx qy qz qw] (64 B per solved window) that feeds
// the global_fusion pose graph (src/global_fusion/poseGraphOptimization.cpp:116-121 consumes position + quaternion + stamp). One process per GPU, one
// communicator per process; windows are independent, so nothing else crosses xGMI. The message is a few KB: latency-bound, one ncclAllGather per step.
// RCCL is loaded on first use (dlopen): processes that never gather (single GPU, tests) do not pay for loading it, and the library has no link-time
// dependency on a particular librccl (a host program that already carries one — PyTorch does — keeps using its own).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <string>
#include "vilf_internal.hpp"
#include "vilf_device.hpp"

namespace {
typedef struct { char internal[128]; } rcclUniqueId_t;      // ncclUniqueId (rccl.h:43, NCCL_UNIQUE_ID_BYTES = 128)
typedef void *rcclComm_t;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(rcclUniqueId_t *) = nullptr;
    int (*CommInitRank)(rcclComm_t *, int, rcclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(rcclComm_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, rcclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*CommCount)(rcclComm_t, int *) = nullptr;          // optional: vilf_comm_ranks falls back to what vilf_comm_create was told
    int (*CommUserRank)(rcclComm_t, int *) = nullptr;
    bool ok = false;
    std::string err;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {                       // first callers from several threads: one of them fills the table, the others wait
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (r.lib) break; }
        if (!r.lib) { r.err = std::string("librccl not found: ") + dlerror(); return; }
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
        r.CommCount = (decltype(r.CommCount))dlsym(r.lib, "ncclCommCount");
        r.CommUserRank = (decltype(r.CommUserRank))dlsym(r.lib, "ncclCommUserRank");
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
        if (!r.ok) r.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather";
    });
    return r;
}
thread_local std::string g_comm_err;
}  // namespace

struct vilf_comm {
    rcclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" const char *vilf_comm_last_error(void) { return g_comm_err.c_str(); }

extern "C" int vilf_comm_unique_id(unsigned char id[VILF_COMM_ID_BYTES]) {
    if (!id) return VILF_ERR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    if (!r.ok) { g_comm_err = r.err; return VILF_ERR_DEVICE; }
    rcclUniqueId_t u;
    const int rc = r.GetUniqueId(&u);
    if (rc != 0) { g_comm_err = std::string("ncclGetUniqueId: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); return VILF_ERR_DEVICE; }
    static_assert(sizeof(u) == VILF_COMM_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id, &u, sizeof(u));
    return VILF_OK;
}

extern "C" int vilf_comm_create(const unsigned char id[VILF_COMM_ID_BYTES], int world_size, int rank, int device, vilf_comm **out) {
    if (!id || !out || world_size < 1 || rank < 0 || rank >= world_size) return VILF_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return VILF_ERR_NO_GPU;
    if (device < 0 || device >= ndev) return VILF_ERR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    if (!r.ok) { g_comm_err = r.err; return VILF_ERR_DEVICE; }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);                  // the caller's current device is restored: a library call must not change it
    if (hipSetDevice(device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return VILF_ERR_DEVICE; }
    rcclUniqueId_t u;
    std::memcpy(&u, id, sizeof(u));
    vilf_comm *c = new vilf_comm();
    c->world = world_size; c->rank = rank; c->device = device;
    const int rc = r.CommInitRank(&c->comm, world_size, u, rank);
    if (prev_dev >= 0 && prev_dev != device) (void)hipSetDevice(prev_dev);
    if (rc != 0) { g_comm_err = std::string("ncclCommInitRank: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); delete c; return VILF_ERR_DEVICE; }
    *out = c;
    return VILF_OK;
}

extern "C" int vilf_comm_destroy(vilf_comm *c) {
    if (!c) return VILF_ERR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    if (r.ok && c->comm) r.CommDestroy(c->comm);
    delete c;
    return VILF_OK;
}

// all ranks: n_local rows of 8 doubles in, world_size * n_local rows out (rank-major = global unit order under contiguous sharding), both DEVICE buffers.
// Enqueued on hip_stream (NULL: the default stream); the caller synchronises.
extern "C" int vilf_gather_poses(vilf_comm *c, void *hip_stream, const double *local_dev8, int n_local, double *out_dev8) {
    if (!c || !local_dev8 || !out_dev8 || n_local < 0) return VILF_ERR_INVALID_ARGUMENT;
    if (n_local == 0) return VILF_OK;
    Rccl &r = rccl();
    if (!r.ok) { g_comm_err = r.err; return VILF_ERR_DEVICE; }
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    if (prev_dev != c->device && hipSetDevice(c->device) != hipSuccess) { g_comm_err = "hipSetDevice failed"; return VILF_ERR_DEVICE; }
    const int rc = r.AllGather(local_dev8, out_dev8, (size_t)n_local * 8, 8 /* ncclFloat64 (rccl.h:467) */, c->comm, (hipStream_t)hip_stream);
    if (prev_dev >= 0 && prev_dev != c->device) (void)hipSetDevice(prev_dev);
    if (rc != 0) { g_comm_err = std::string("ncclAllGather: ") + (r.GetErrorString ? r.GetErrorString(rc) : "error"); return VILF_ERR_DEVICE; }
    return VILF_OK;
}

// the gather on the stream the handle works on (the library's own streams are non-blocking: the NULL stream is NOT ordered behind them)
extern "C" int vilf_gather_poses_handle(vilf_comm *c, vilf_handle *h, const double *local_dev8, int n_local, double *out_dev8) {
    if (!h) return VILF_ERR_INVALID_ARGUMENT;
    return vilf_gather_poses(c, (void *)h->stream, local_dev8, n_local, out_dev8);
}
extern "C" int vilf_get_stream(vilf_handle *h, void **hip_stream_out) {
    if (!h || !hip_stream_out) return VILF_ERR_INVALID_ARGUMENT;
    *hip_stream_out = (void *)h->stream;
    return VILF_OK;
}
extern "C" int vilf_comm_ranks(vilf_comm *c, int *world_size_out, int *rank_out) {
    if (!c) return VILF_ERR_INVALID_ARGUMENT;
    Rccl &r = rccl();
    int n = c->world, me = c->rank;
    if (r.ok && c->comm && r.CommCount && r.CommUserRank) {
        if (r.CommCount(c->comm, &n) != 0 || r.CommUserRank(c->comm, &me) != 0) { g_comm_err = "ncclCommCount / ncclCommUserRank failed"; return VILF_ERR_DEVICE; }
    }
    if (world_size_out) *world_size_out = n;
    if (rank_out) *rank_out = me;
    return VILF_OK;
}

// newest-frame pose rows on the device: no host round trip between the solve and the gather
extern "C" __global__ void k_newest_poses(int B, const double *out_Ps, const double *out_Rs, const double *stamps, double *out8) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= B) return;
    const double *P = out_Ps + (size_t)w * 33 + 30, *R = out_Rs + (size_t)w * 99 + 90;
    const vd::Q q = vd::q_fromR(R);
    double *o = out8 + (size_t)w * 8;
    o[0] = stamps ? stamps[w] : (double)w;
    o[1] = P[0]; o[2] = P[1]; o[3] = P[2];
    o[4] = q.x; o[5] = q.y; o[6] = q.z; o[7] = q.w;
}

extern "C" int vilf_batch_newest_poses_device(vilf_handle *h, const double *stamps_host, void *device_out8) {
    if (!h || !h->resident || !device_out8) return VILF_ERR_INVALID_ARGUMENT;
    const int B = h->B;
    const double *stamps_dev = nullptr;
    if (stamps_host) {
        // The caller's stamps go through a pinned staging buffer the handle owns, so the call returns without waiting for the stream (the gather path has no host
        // wait). The buffer is re-used by the next call: only then — normally long after the copy has run — the event of the previous copy is waited for.
        if (!h->d[D_STAMPS].ensure((size_t)B * 8)) return VILF_ERR_DEVICE;
        if (h->stamp_cap < (size_t)B * 8) {
            if (h->stamp_pinned) { HIPCHECK(h, hipStreamSynchronize(h->stream)); (void)hipHostFree(h->stamp_pinned); h->stamp_pinned = nullptr; h->stamp_cap = 0; }
            HIPCHECK(h, hipHostMalloc(&h->stamp_pinned, (size_t)B * 8, hipHostMallocDefault));
            h->stamp_cap = (size_t)B * 8;
        }
        if (!h->stamp_ev) HIPCHECK(h, hipEventCreateWithFlags(&h->stamp_ev, hipEventDisableTiming));
        else HIPCHECK(h, hipEventSynchronize(h->stamp_ev));
        std::memcpy(h->stamp_pinned, stamps_host, (size_t)B * 8);
        HIPCHECK(h, hipMemcpyAsync(h->d[D_STAMPS].p, h->stamp_pinned, (size_t)B * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(h, hipEventRecord(h->stamp_ev, h->stream));
        stamps_dev = h->d[D_STAMPS].as<double>();
    }
    hipLaunchKernelGGL(k_newest_poses, dim3((B + 127) / 128), dim3(128), 0, h->stream, B, h->batch.out_Ps, h->batch.out_Rs, stamps_dev, (double *)device_out8);
    HIPCHECK(h, hipGetLastError());
    return VILF_OK;
}

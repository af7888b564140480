// vilf_host.hip — IMU pre-integration ≙ IntegrationBase::{push_back,propagate,midPointIntegration} (factor/integration_base.h:30-158)
//   vilf_imu_preintegrate        one interval on the host: the reference integrates per IMU message on the estimator's host thread
//                                (estimator.cpp:103-137) and hands the result to the IMU factor
//   vilf_imu_preintegrate_batch  many intervals on the device: one lane per interval runs the SAME routine
//                                (preintegrate_core is compiled for host and device), e.g. to re-integrate every interval of a batch of
//                                windows after a bias update, or to prepare replayed windows without a host loop
#include <hip/hip_runtime.h>
#include <cstring>
#include <cmath>
#include "vilf_internal.hpp"

#define VPI __host__ __device__ inline
namespace {
struct M3 { double m[9]; };
VPI M3 mul(const M3 &a, const M3 &b) { M3 c; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j]; return c; }
VPI M3 add(const M3 &a, const M3 &b) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] + b.m[i]; return c; }
VPI M3 scl(const M3 &a, double s) { M3 c; for (int i = 0; i < 9; i++) c.m[i] = a.m[i] * s; return c; }
VPI M3 eye() { M3 c; for (int i = 0; i < 9; i++) c.m[i] = 0; c.m[0] = c.m[4] = c.m[8] = 1; return c; }
VPI M3 skew(const double *v) { M3 s; s.m[0] = 0; s.m[1] = -v[2]; s.m[2] = v[1]; s.m[3] = v[2]; s.m[4] = 0; s.m[5] = -v[0]; s.m[6] = -v[1]; s.m[7] = v[0]; s.m[8] = 0; return s; }
VPI void qmul(const double *a, const double *b, double *o) {  // x y z w
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
}
VPI M3 toR(const double *q) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3 R; R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz; R.m[2] = txz + twy; R.m[3] = txy + twz; R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
    R.m[6] = txz - twy; R.m[7] = tyz + twx; R.m[8] = 1 - (txx + tyy);
    return R;
}
VPI void qrot(const double *q, const double *v, double *o) {
    double ux = 2 * (q[1] * v[2] - q[2] * v[1]), uy = 2 * (q[2] * v[0] - q[0] * v[2]), uz = 2 * (q[0] * v[1] - q[1] * v[0]);
    o[0] = v[0] + q[3] * ux + (q[1] * uz - q[2] * uy);
    o[1] = v[1] + q[3] * uy + (q[2] * ux - q[0] * uz);
    o[2] = v[2] + q[3] * uz + (q[0] * uy - q[1] * ux);
}
VPI void setb(double *M, int ld, int r0, int c0, const M3 &b) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) M[(r0 + i) * ld + c0 + j] = b.m[3 * i + j]; }

VPI void preintegrate_core(const vilf_imu_noise *nz, const double acc_0_[3], const double gyr_0_[3], const double ba[3], const double bg[3],
                           int n, const double *dts, const double *accs, const double *gyrs, vilf_imu_preint *out) {
    double acc_0[3] = {acc_0_[0], acc_0_[1], acc_0_[2]}, gyr_0[3] = {gyr_0_[0], gyr_0_[1], gyr_0_[2]};
    double dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1}, sum_dt = 0;
    double J[225], P[225], noise[18];
    for (int i = 0; i < 225; i++) { J[i] = 0; P[i] = 0; }
    for (int i = 0; i < 15; i++) J[16 * i] = 1.0;
    for (int i = 0; i < 3; i++) {
        noise[i] = nz->acc_n * nz->acc_n; noise[3 + i] = nz->gyr_n * nz->gyr_n; noise[6 + i] = nz->acc_n * nz->acc_n;
        noise[9 + i] = nz->gyr_n * nz->gyr_n; noise[12 + i] = nz->acc_w * nz->acc_w; noise[15 + i] = nz->gyr_w * nz->gyr_w;
    }
    for (int s = 0; s < n; s++) {
        const double dt = dts[s];
        const double *a1 = accs + 3 * s, *g1 = gyrs + 3 * s;
        double a0b[3], a1b[3], w[3];
        for (int k = 0; k < 3; k++) { a0b[k] = acc_0[k] - ba[k]; a1b[k] = a1[k] - ba[k]; w[k] = 0.5 * (gyr_0[k] + g1[k]) - bg[k]; }
        double un0[3], un1[3], un[3], rq[4];
        qrot(dq, a0b, un0);
        const double dstep[4] = {w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0};
        qmul(dq, dstep, rq);
        qrot(rq, a1b, un1);
        for (int k = 0; k < 3; k++) un[k] = 0.5 * (un0[k] + un1[k]);
        double rp[3], rv[3];
        for (int k = 0; k < 3; k++) { rp[k] = dp[k] + dv[k] * dt + 0.5 * un[k] * dt * dt; rv[k] = dv[k] + un[k] * dt; }
        const M3 Rw = skew(w), Ra0 = skew(a0b), Ra1 = skew(a1b), I3 = eye(), Rd = toR(dq), Rr = toR(rq);
        const M3 ImRw = add(I3, scl(Rw, -dt));
        double F[225], V[270];
        for (int i = 0; i < 225; i++) F[i] = 0;
        for (int i = 0; i < 270; i++) V[i] = 0;
        setb(F, 15, 0, 0, I3);
        setb(F, 15, 0, 3, add(scl(mul(Rd, Ra0), -0.25 * dt * dt), scl(mul(mul(Rr, Ra1), ImRw), -0.25 * dt * dt)));
        setb(F, 15, 0, 6, scl(I3, dt));
        setb(F, 15, 0, 9, scl(add(Rd, Rr), -0.25 * dt * dt));
        setb(F, 15, 0, 12, scl(mul(Rr, Ra1), -0.25 * dt * dt * -dt));
        setb(F, 15, 3, 3, ImRw);
        setb(F, 15, 3, 12, scl(I3, -dt));
        setb(F, 15, 6, 3, add(scl(mul(Rd, Ra0), -0.5 * dt), scl(mul(mul(Rr, Ra1), ImRw), -0.5 * dt)));
        setb(F, 15, 6, 6, I3);
        setb(F, 15, 6, 9, scl(add(Rd, Rr), -0.5 * dt));
        setb(F, 15, 6, 12, scl(mul(Rr, Ra1), -0.5 * dt * -dt));
        setb(F, 15, 9, 9, I3);
        setb(F, 15, 12, 12, I3);
        const M3 V03 = scl(mul(Rr, Ra1), -0.25 * dt * dt * 0.5 * dt), V63 = scl(mul(Rr, Ra1), -0.5 * dt * 0.5 * dt);
        setb(V, 18, 0, 0, scl(Rd, 0.25 * dt * dt)); setb(V, 18, 0, 3, V03); setb(V, 18, 0, 6, scl(Rr, 0.25 * dt * dt)); setb(V, 18, 0, 9, V03);
        setb(V, 18, 3, 3, scl(I3, 0.5 * dt)); setb(V, 18, 3, 9, scl(I3, 0.5 * dt));
        setb(V, 18, 6, 0, scl(Rd, 0.5 * dt)); setb(V, 18, 6, 3, V63); setb(V, 18, 6, 6, scl(Rr, 0.5 * dt)); setb(V, 18, 6, 9, V63);
        setb(V, 18, 9, 12, scl(I3, dt)); setb(V, 18, 12, 15, scl(I3, dt));
        double FJ[225], FP[225], NP[225];
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) { double a = 0, b2 = 0; for (int k = 0; k < 15; k++) { a += F[15 * i + k] * J[15 * k + j]; b2 += F[15 * i + k] * P[15 * k + j]; } FJ[15 * i + j] = a; FP[15 * i + j] = b2; }
        for (int i = 0; i < 15; i++) for (int j = 0; j < 15; j++) {
            double a = 0; for (int k = 0; k < 15; k++) a += FP[15 * i + k] * F[15 * j + k];
            double c = 0; for (int k = 0; k < 18; k++) c += V[18 * i + k] * noise[k] * V[18 * j + k];
            NP[15 * i + j] = a + c;
        }
        for (int i = 0; i < 225; i++) { J[i] = FJ[i]; P[i] = NP[i]; }
        const double nq = sqrt(rq[0] * rq[0] + rq[1] * rq[1] + rq[2] * rq[2] + rq[3] * rq[3]);
        for (int k = 0; k < 4; k++) dq[k] = rq[k] / nq;
        for (int k = 0; k < 3; k++) { dp[k] = rp[k]; dv[k] = rv[k]; acc_0[k] = a1[k]; gyr_0[k] = g1[k]; }
        sum_dt += dt;
    }
    out->sum_dt = sum_dt;
    for (int k = 0; k < 3; k++) { out->delta_p[k] = dp[k]; out->delta_v[k] = dv[k]; out->linearized_ba[k] = ba[k]; out->linearized_bg[k] = bg[k]; }
    for (int k = 0; k < 4; k++) out->delta_q[k] = dq[k];
    for (int i = 0; i < 225; i++) { out->jacobian[i] = J[i]; out->covariance[i] = P[i]; }
}
}  // namespace

extern "C" int vilf_imu_preintegrate(const vilf_imu_noise *nz, const double acc_0[3], const double gyr_0[3], const double ba[3], const double bg[3],
                                     int n, const double *dts, const double *accs, const double *gyrs, vilf_imu_preint *out) {
    if (!nz || !acc_0 || !gyr_0 || !ba || !bg || n < 0 || !out) return VILF_ERR_INVALID_ARGUMENT;
    preintegrate_core(nz, acc_0, gyr_0, ba, bg, n, dts, accs, gyrs, out);
    return VILF_OK;
}

// one lane per interval; inputs are [n][...] arrays with max_samples slots per interval
__global__ void k_preintegrate(int n, vilf_imu_noise nz, const double *acc0, const double *gyr0, const double *ba, const double *bg, const int *n_samples, int max_samples,
                               const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    preintegrate_core(&nz, acc0 + 3 * i, gyr0 + 3 * i, ba + 3 * i, bg + 3 * i, n_samples[i], dt + (size_t)i * max_samples, acc + (size_t)i * max_samples * 3, gyr + (size_t)i * max_samples * 3, out + i);
}

extern "C" int vilf_imu_preintegrate_batch(vilf_handle *h, int n, const vilf_imu_noise *nz, const double *acc_0, const double *gyr_0, const double *ba, const double *bg,
                                           const int *n_samples, int max_samples, const double *dt, const double *acc, const double *gyr, vilf_imu_preint *out) {
    if (!h || n < 0 || max_samples < 0 || !nz || (n && (!acc_0 || !gyr_0 || !ba || !bg || !n_samples || !out)) || (n && max_samples && (!dt || !acc || !gyr))) return VILF_ERR_INVALID_ARGUMENT;
    if (n == 0) return VILF_OK;
    for (int i = 0; i < n; i++) if (n_samples[i] < 0 || n_samples[i] > max_samples) { h->err = "n_samples out of range"; return VILF_ERR_INVALID_ARGUMENT; }
    HIPCHECK(h, hipSetDevice(h->device));
    const size_t sn = n, sm = (size_t)n * std::max(max_samples, 1);
    const size_t off_a0 = 0, off_g0 = off_a0 + 3 * sn, off_ba = off_g0 + 3 * sn, off_bg = off_ba + 3 * sn, off_dt = off_bg + 3 * sn, off_acc = off_dt + sm, off_gyr = off_acc + 3 * sm, off_ns = off_gyr + 3 * sm;
    const size_t in_doubles = off_ns + (sn + 1) / 2 + 1;
    if (!h->d[D_HOOK].ensure(in_doubles * 8 + sn * sizeof(vilf_imu_preint) + 64)) { h->err = "hipMalloc failed (pre-integration)"; return VILF_ERR_DEVICE; }
    double *d = h->d[D_HOOK].as<double>();
    auto up = [&](size_t off, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(d + off, src, bytes, hipMemcpyHostToDevice, h->stream) : hipSuccess; };
    HIPCHECK(h, up(off_a0, acc_0, 24 * sn)); HIPCHECK(h, up(off_g0, gyr_0, 24 * sn)); HIPCHECK(h, up(off_ba, ba, 24 * sn)); HIPCHECK(h, up(off_bg, bg, 24 * sn));
    HIPCHECK(h, up(off_dt, dt, 8 * (size_t)n * max_samples)); HIPCHECK(h, up(off_acc, acc, 24 * (size_t)n * max_samples)); HIPCHECK(h, up(off_gyr, gyr, 24 * (size_t)n * max_samples));
    HIPCHECK(h, up(off_ns, n_samples, 4 * sn));
    vilf_imu_preint *d_out = reinterpret_cast<vilf_imu_preint *>(d + in_doubles);
    hipLaunchKernelGGL(k_preintegrate, dim3((n + 63) / 64), dim3(64), 0, h->stream, n, *nz, d + off_a0, d + off_g0, d + off_ba, d + off_bg, reinterpret_cast<const int *>(d + off_ns), max_samples,
                       d + off_dt, d + off_acc, d + off_gyr, d_out);
    HIPCHECK(h, hipGetLastError());
    HIPCHECK(h, hipMemcpyAsync(out, d_out, sn * sizeof(vilf_imu_preint), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(h, hipStreamSynchronize(h->stream));
    return VILF_OK;
}


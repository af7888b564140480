// Issue rate and dependent latency of fp64 on gfx950, measured: s_memtime (core clock cycles) against s_memrealtime (constant 100 MHz) around chains of
// v_fma_f64 / v_mfma_f64_16x16x4_f64, ONE workgroup of 64 .. 1024 threads (1 .. 4 waves per SIMD) with 1 .. 8 independent chains per lane.
// Build + run: bash tools/micro/clock_probe.sh (on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int ILP>
__global__ void probe_fma(double *out, long long *stamps, int n) {
    double x[ILP];
    for (int k = 0; k < ILP; k++) x[k] = out[(threadIdx.x + k) & 7];
    const double y = 1.0000001;
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) x[k] = fma(x[k], y, 1e-9);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < ILP; k++) s += x[k];
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&stamps[0], (unsigned long long)(c1 - c0)); atomicMax((unsigned long long *)&stamps[1], (unsigned long long)(w1 - w0)); }     // the slowest wave: the arbiter favours the oldest
}
template <int ILP>
__global__ void probe_mfma(double *out, long long *stamps, int n) {
    d4 acc[ILP];
    for (int k = 0; k < ILP; k++) acc[k] = d4{0, 0, 0, 0};
    const double a = out[threadIdx.x & 7], b = out[(threadIdx.x + 1) & 7];
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int k = 0; k < ILP; k++) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int k = 0; k < ILP; k++) s += acc[k][0] + acc[k][3];
    if (s == 12345.678) out[0] = s;
    if ((threadIdx.x & 63) == 0) { atomicMax((unsigned long long *)&stamps[0], (unsigned long long)(c1 - c0)); atomicMax((unsigned long long *)&stamps[1], (unsigned long long)(w1 - w0)); }
}
template <class K>
static void run(const char *what, K kern, int threads, int ilp, double *out, long long *st) {
    long long h[2];
    const int n = 20000;
    hipMemset(st, 0, 16);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, st, n);
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("%-6s threads %4d (%d wave(s) per SIMD)  chains per lane %d : %7.2f cycles per instruction of a chain, %6.2f cycles per instruction issued per SIMD, %.0f MHz\n", what, threads,
           (threads + 255) / 256, ilp, (double)h[0] / n, (double)h[0] / n / ilp / ((threads + 255) / 256), 100.0 * h[0] / h[1]);
}
int main() {
    double *out; long long *st;
    hipMalloc(&out, 64); hipMemset(out, 0, 64); hipMalloc(&st, 16);
    for (int threads : {64, 256, 512, 1024}) {
        run("fma64", probe_fma<1>, threads, 1, out, st); run("fma64", probe_fma<2>, threads, 2, out, st); run("fma64", probe_fma<4>, threads, 4, out, st); run("fma64", probe_fma<8>, threads, 8, out, st);
    }
    for (int threads : {256, 512, 1024}) {
        run("mfma64", probe_mfma<1>, threads, 1, out, st); run("mfma64", probe_mfma<2>, threads, 2, out, st); run("mfma64", probe_mfma<4>, threads, 4, out, st);
    }
    return 0;
}

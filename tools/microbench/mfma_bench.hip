#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void k(double *out, long long *cyc, int iters, double a0, double b0) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = double4_t{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
// fp64 FMA rate
__global__ void kf(double *out, long long *cyc, int iters, double a0, double b0) {
    double x[8];
    for (int i = 0; i < 8; i++) x[i] = a0 + i + threadIdx.x;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = fma(x[i], b0, a0);
    }
    long long t1 = __builtin_readcyclecounter();
    double s = 0; for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double *out; long long *cyc, h;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
    const int iters = 1000;
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %d: 1 dependent acc : %.1f cycles / mfma\n", threads, (double)h / iters);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %d: 4 independent acc: %.1f cycles / mfma\n", threads, (double)h / iters / 4);
        hipLaunchKernelGGL(k<8>, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 2.0); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %d: 8 independent acc: %.1f cycles / mfma\n", threads, (double)h / iters / 8);
        hipLaunchKernelGGL(kf, dim3(1), dim3(threads), 0, 0, out, cyc, iters, 1.0, 0.5); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("threads %d: v_fma_f64 8 chains: %.1f cycles / fma instr\n", threads, (double)h / iters / 8);
    }
    return 0;
}

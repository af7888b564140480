// micro check: raw buffer loads (SGPR descriptor + 32-bit offset) against plain loads; out-of-range offsets must return 0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int uint2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bload(__amdgpu_buffer_rsrc_t r, int elem) {
    const uint2_t v = __builtin_amdgcn_raw_buffer_load_b64(r, elem < 0 ? 0x7ffffff0u : 8u * (unsigned)elem, 0, 0);
    return __hiloint2double((int)v.y, (int)v.x);
}
__global__ void k(const double *a, int n, double *o) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)a, 0, n * 8, 0x00027000);
    const int t = threadIdx.x;
    o[t] = bload(r, (t * 37) % n);
    o[256 + t] = bload(r, -1);
    o[512 + t] = bload(r, n + t);      // just past the end
}
int main() {
    const int n = 9000;
    std::vector<double> h(n); for (int i = 0; i < n; i++) h[i] = 1.0 + i * 0.5;
    double *a, *o; hipMalloc(&a, n * 8 + 4096); hipMalloc(&o, 768 * 8);
    hipMemset(a, 0xff, n * 8 + 4096);
    hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<1, 256>>>(a, n, o);
    std::vector<double> r(768); hipMemcpy(r.data(), o, 768 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; t++) { if (r[t] != h[(t * 37) % n]) bad++; if (r[256 + t] != 0.0) bad++; if (r[512 + t] != 0.0) bad++; }
    printf("bufload: %d mismatches; sample %g %g %g\n", bad, r[5], r[256 + 5], r[512 + 5]);
    return bad != 0;
}

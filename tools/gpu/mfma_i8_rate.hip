#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void rate(int *out, int iters, long long *cyc)
{
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)threadIdx.x, 7};
    v16i c = {0};
    long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < 8; s++) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    }
    long long t1 = clock64();
    int s = 0; for (int g = 0; g < 16; g++) s += c[g];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main()
{
    int *d; long long *dc, hc;
    hipMalloc(&d, 4 * 1024 * 1024); hipMalloc(&dc, 8);
    for (int wpb : {64, 256, 512}) {
        hipLaunchKernelGGL(rate, dim3(256), dim3(wpb), 0, 0, d, 1000, dc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); hipLaunchKernelGGL(rate, dim3(256), dim3(wpb), 0, 0, d, 1000, dc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&hc, dc, 8, hipMemcpyDeviceToHost);
        printf("threads/block %d: 8000 dependent i8 32x32x32 MFMAs per wave: %lld s_memtime ticks (%.1f per MFMA), kernel %.3f ms -> %.1f ns per MFMA per wave\n", wpb, hc, hc / 8000.0, ms, ms * 1e6 / 8000);
    }
}

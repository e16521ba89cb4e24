// mfma_i8_probe.hip -- operand and result lane maps of v_mfma_i32_32x32x32_i8 on gfx950, checked with exact
// integer data (cdna_hip_programming.md: "other dtypes: check the map with exact integer data before relying on it").
// Assumed: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 16 h + j] and B[k = 16 h + j][col r] in byte j = 0..15
// of its 128-bit fragment; C/D: col = l & 31, row = (reg & 3) + 8 (reg >> 2) + 4 h, reg in [0, 16).
// Build + run on the GPU box:  hipcc -O2 --offload-arch=gfx950 tools/gpu/mfma_i8_probe.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void probe(const int8_t *A, const int8_t *B, int *C)      // A: 32 x 32 (row-major [i][k]), B: 32 x 32 ([k][j]), C: 32 x 32 ([i][j])
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    int8_t *pa = (int8_t *)&a, *pb = (int8_t *)&b;
    for (int j = 0; j < 16; j++) { pa[j] = A[r * 32 + 16 * h + j]; pb[j] = B[(16 * h + j) * 32 + r]; }
    v16i c = {0};
    c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
    for (int reg = 0; reg < 16; reg++) C[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = c[reg];
}

int main()
{
    int8_t hA[1024], hB[1024];
    int hC[1024], ref[1024];
    srand(7);
    for (int i = 0; i < 1024; i++) { hA[i] = (int8_t)(rand() % 255 - 127); hB[i] = (int8_t)(rand() % 255 - 127); }
    for (int i = 0; i < 32; i++)
        for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += (int)hA[i * 32 + k] * (int)hB[k * 32 + j]; ref[i * 32 + j] = s; }
    int8_t *dA, *dB; int *dC;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
    hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 1024; i++) bad += hC[i] != ref[i];
    printf("mfma_i32_32x32x32_i8 with the assumed lane maps: %d of 1024 results differ from the reference product\n", bad);
    return bad != 0;
}

// valu_roof.hip -- measured issue rate of the vector instructions lk_kernel is made of, on gfx950.
//
// For every instruction: a stream of INDEPENDENT chains (8 accumulators per lane, each instruction
// depends only on the one 8 slots earlier) at 1 / 2 / 4 / 8 waves per SIMD, every CU busy, plus a
// fully DEPENDENT stream (1 accumulator) for the latency.  Reported: wave-instructions per cycle per
// SIMD (cycles = s_memtime span of the workgroup, i.e. shader clocks) and the wall-clock rate.
// The mixed stream `lk_mix` reproduces the instruction mix of lk_kernel's per-iteration pixel work
// (22 dot2 : 11 perm : 4 alignbyte : 4 pk_sub : 3 pk_shift per slot).
//
// Build:  hipcc -O2 --offload-arch=gfx950 tools/gpu/valu_roof.hip -o /tmp/valu_roof
// Output: one line per (instruction, waves/SIMD); profiles/r02_valu_roof.txt is a captured run.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kIters = 256;           // loop trips
constexpr int kUnroll = 16;           // groups of 8 per trip -> 128 instructions per trip

// one instruction on accumulator A with loop-invariant operands X, Y
#define OP_dot2(A, X, Y)      asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_dot4(A, X, Y)      asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_perm(A, X, Y)      asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_alignbyte(A, X, Y) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_pk_sub(A, X, Y)    asm volatile("v_pk_sub_u16 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_pk_add(A, X, Y)    asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_pk_mul(A, X, Y)    asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_pk_mad(A, X, Y)    asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_pk_lshr(A, X, Y)   asm volatile("v_pk_lshrrev_b16 %0, 1, %0 op_sel_hi:[0,1]" : "+v"(A))
#define OP_add_u32(A, X, Y)   asm volatile("v_add_u32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_add3(A, X, Y)      asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_and_or(A, X, Y)    asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_lshl_add(A, X, Y)  asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(A) : "v"(X))
#define OP_ashr(A, X, Y)      asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(A))
#define OP_mul_lo(A, X, Y)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mad_u24(A, X, Y)   asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_fma_f32(A, X, Y)   asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_mul_f32(A, X, Y)   asm volatile("v_mul_f32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_cvt_i2f(A, X, Y)   asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(A))
#define OP_rndne(A, X, Y)     asm volatile("v_rndne_f32 %0, %0" : "+v"(A))
#define OP_dpp_add(A, X, Y)   asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(A))
#define OP_mov_dpp(A, X, Y)   asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(A) : "v"(X))
#define OP_cndmask(A, X, Y)   asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(X))
#define OP_bcnt(A, X, Y)      asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(A) : "v"(X))
#define OP_min3(A, X, Y)      asm volatile("v_min3_u32 %0, %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_sqrt(A, X, Y)      asm volatile("v_sqrt_f32 %0, %0" : "+v"(A))
#define OP_rcp(A, X, Y)       asm volatile("v_rcp_f32 %0, %0" : "+v"(A))
#define OP_fma_f64(A, X, Y)   asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(A) : "v"(X))


// ---- second batch: VOP2 / VOP1 / VOPC / SDWA / cross-lane forms -------------------------------
#define OP_dot2c(A, X, Y)     asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_dot4c(A, X, Y)     asm volatile("v_dot4c_i32_i8 %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_fmac_f32(A, X, Y)  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_add_f32(A, X, Y)   asm volatile("v_add_f32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mul_i24(A, X, Y)   asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mul_u24(A, X, Y)   asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mad_i24(A, X, Y)   asm volatile("v_mad_i32_i24 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_and(A, X, Y)       asm volatile("v_and_b32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_or(A, X, Y)        asm volatile("v_or_b32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_xor(A, X, Y)       asm volatile("v_xor_b32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_lshl(A, X, Y)      asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(A))
#define OP_lshr(A, X, Y)      asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(A))
#define OP_sub_u32(A, X, Y)   asm volatile("v_sub_u32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_max_i32(A, X, Y)   asm volatile("v_max_i32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_min_u32(A, X, Y)   asm volatile("v_min_u32 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mov(A, X, Y)       asm volatile("v_mov_b32 %0, %1" : "+v"(A) : "v"(X))
#define OP_mov_sgpr(A, X, Y)  asm volatile("v_mov_b32 %0, s20" : "+v"(A))
#define OP_add_sgpr(A, X, Y)  asm volatile("v_add_u32 %0, s20, %0" : "+v"(A))
#define OP_dot2_sgpr(A, X, Y) asm volatile("v_dot2_i32_i16 %0, %1, %2, s20" : "+v"(A) : "v"(X), "v"(Y))
#define OP_cvt_f2i(A, X, Y)   asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(A))
#define OP_floor(A, X, Y)     asm volatile("v_floor_f32 %0, %0" : "+v"(A))
#define OP_cvt_ub0(A, X, Y)   asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(A))
#define OP_cmp_vcc(A, X, Y)   asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(A), "v"(X) : "vcc")
#define OP_cmp_sgpr(A, X, Y)  asm volatile("v_cmp_lt_i32 s[20:21], %0, %1" : : "v"(A), "v"(X) : "s20", "s21")
#define OP_cnd_vcc0(A, X, Y)  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(A) : "v"(X))
#define OP_cnd_s(A, X, Y)     asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(A) : "v"(X))
#define OP_readlane(A, X, Y)  asm volatile("v_readlane_b32 s22, %0, 3" : : "v"(A) : "s22")
#define OP_readfl(A, X, Y)    asm volatile("v_readfirstlane_b32 s22, %0" : : "v"(A) : "s22")
#define OP_add_sdwa(A, X, Y)  asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(A) : "v"(X))
#define OP_mul_sdwa(A, X, Y)  asm volatile("v_mul_u32_u24_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:BYTE_2" : "+v"(A) : "v"(X))
#define OP_add_u16(A, X, Y)   asm volatile("v_add_u16 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mul_lo_u16(A, X, Y) asm volatile("v_mul_lo_u16 %0, %0, %1" : "+v"(A) : "v"(X))
#define OP_mad_u16(A, X, Y)   asm volatile("v_mad_u32_u16 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_lshl_or(A, X, Y)   asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(A) : "v"(X))
#define OP_bfe(A, X, Y)       asm volatile("v_bfe_u32 %0, %0, 1, 20" : "+v"(A))
#define OP_bfi(A, X, Y)       asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(A) : "v"(X), "v"(Y))
#define OP_alignbit(A, X, Y)  asm volatile("v_alignbit_b32 %0, %0, %1, 9" : "+v"(A) : "v"(X))
#define OP_pk_fma_f32(A, X, Y) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(A) : "v"(X))
#define OP_pk_fma_f16(A, X, Y) asm volatile("v_pk_fma_f16 %0, %1, %2, %0" : "+v"(A) : "v"(X), "v"(Y))
#define OP_swap16(A, X, Y)    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(A), "+v"(X))
#define OP_swap32(A, X, Y)    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(A), "+v"(X))
#define OP_ds_read(A, X, Y)   asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(4)" : "+v"(A) : "v"(X))
// alternating 4-cycle and 2-cycle classes: do the costs add?
#define OP_dot2_add(A, X, Y)  asm volatile("v_dot2_i32_i16 %0, %1, %2, %0\n v_add_u32 %0, %0, %1" : "+v"(A) : "v"(X), "v"(Y))

#define GROUP8(OP) OP(a0, x, y); OP(a1, x, y); OP(a2, x, y); OP(a3, x, y); OP(a4, x, y); OP(a5, x, y); OP(a6, x, y); OP(a7, x, y);
#define GROUP1(OP) OP(a0, x, y); OP(a0, x, y); OP(a0, x, y); OP(a0, x, y); OP(a0, x, y); OP(a0, x, y); OP(a0, x, y); OP(a0, x, y);

#define DEFINE_KERNEL(NAME, OP, T, GROUP)                                                          \
    __global__ __launch_bounds__(1024) void k_##NAME(uint64_t *stamps, T *sink, T xin, T yin)     \
    {                                                                                              \
        T a0 = (T)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5,     \
          a6 = a0 + 6, a7 = a0 + 7;                                                                \
        T x = xin, y = yin;                                                                        \
        __syncthreads();                                                                           \
        const uint64_t t0 = __builtin_readcyclecounter();                                          \
        _Pragma("nounroll") for (int it = 0; it < kIters; it++) {                                  \
            _Pragma("unroll") for (int u = 0; u < kUnroll; u++) { GROUP(OP) }                      \
        }                                                                                          \
        const uint64_t t1 = __builtin_readcyclecounter();                                          \
        if ((threadIdx.x & 63) == 0) {                                                             \
            const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);                    \
            stamps[2 * wv] = t0; stamps[2 * wv + 1] = t1;                                          \
        }                                                                                          \
        T r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                               \
        if (r == (T)0x7f123456) sink[0] = r;                                                       \
    }

#define BOTH(NAME, OP, T) DEFINE_KERNEL(NAME, OP, T, GROUP8) DEFINE_KERNEL(NAME##_dep, OP, T, GROUP1)

BOTH(dot2, OP_dot2, uint32_t)
BOTH(dot4, OP_dot4, uint32_t)
BOTH(perm, OP_perm, uint32_t)
BOTH(alignbyte, OP_alignbyte, uint32_t)
BOTH(pk_sub, OP_pk_sub, uint32_t)
BOTH(pk_add, OP_pk_add, uint32_t)
BOTH(pk_mul, OP_pk_mul, uint32_t)
BOTH(pk_mad, OP_pk_mad, uint32_t)
BOTH(pk_lshr, OP_pk_lshr, uint32_t)
BOTH(add_u32, OP_add_u32, uint32_t)
BOTH(add3, OP_add3, uint32_t)
BOTH(and_or, OP_and_or, uint32_t)
BOTH(lshl_add, OP_lshl_add, uint32_t)
BOTH(ashr, OP_ashr, uint32_t)
BOTH(mul_lo, OP_mul_lo, uint32_t)
BOTH(mad_u24, OP_mad_u24, uint32_t)
BOTH(fma_f32, OP_fma_f32, float)
BOTH(mul_f32, OP_mul_f32, float)
BOTH(cvt_i2f, OP_cvt_i2f, uint32_t)
BOTH(rndne, OP_rndne, float)
BOTH(dpp_add, OP_dpp_add, uint32_t)
BOTH(mov_dpp, OP_mov_dpp, uint32_t)
BOTH(cndmask, OP_cndmask, uint32_t)
BOTH(bcnt, OP_bcnt, uint32_t)
BOTH(min3, OP_min3, uint32_t)
BOTH(sqrt_f32, OP_sqrt, float)
BOTH(rcp_f32, OP_rcp, float)
BOTH(fma_f64, OP_fma_f64, double)


BOTH(dot2c, OP_dot2c, uint32_t)
BOTH(dot4c, OP_dot4c, uint32_t)
BOTH(fmac_f32, OP_fmac_f32, float)
BOTH(add_f32, OP_add_f32, float)
BOTH(mul_i24, OP_mul_i24, uint32_t)
BOTH(mul_u24, OP_mul_u24, uint32_t)
BOTH(mad_i24, OP_mad_i24, uint32_t)
BOTH(and, OP_and, uint32_t)
BOTH(or, OP_or, uint32_t)
BOTH(xor, OP_xor, uint32_t)
BOTH(lshl, OP_lshl, uint32_t)
BOTH(lshr, OP_lshr, uint32_t)
BOTH(sub_u32, OP_sub_u32, uint32_t)
BOTH(max_i32, OP_max_i32, uint32_t)
BOTH(min_u32, OP_min_u32, uint32_t)
BOTH(mov, OP_mov, uint32_t)
BOTH(mov_sgpr, OP_mov_sgpr, uint32_t)
BOTH(add_sgpr, OP_add_sgpr, uint32_t)
BOTH(dot2_sgpr, OP_dot2_sgpr, uint32_t)
BOTH(cvt_f2i, OP_cvt_f2i, uint32_t)
BOTH(floor, OP_floor, float)
BOTH(cvt_ub0, OP_cvt_ub0, uint32_t)
BOTH(cmp_vcc, OP_cmp_vcc, uint32_t)
BOTH(cmp_sgpr, OP_cmp_sgpr, uint32_t)
BOTH(cnd_vcc0, OP_cnd_vcc0, uint32_t)
BOTH(cnd_s, OP_cnd_s, uint32_t)
BOTH(readlane, OP_readlane, uint32_t)
BOTH(readfl, OP_readfl, uint32_t)
BOTH(add_sdwa, OP_add_sdwa, uint32_t)
BOTH(mul_sdwa, OP_mul_sdwa, uint32_t)
BOTH(add_u16, OP_add_u16, uint32_t)
BOTH(mul_lo_u16, OP_mul_lo_u16, uint32_t)
BOTH(mad_u16, OP_mad_u16, uint32_t)
BOTH(lshl_or, OP_lshl_or, uint32_t)
BOTH(bfe, OP_bfe, uint32_t)
BOTH(bfi, OP_bfi, uint32_t)
BOTH(alignbit, OP_alignbit, uint32_t)
BOTH(pk_fma_f32, OP_pk_fma_f32, double)
BOTH(pk_fma_f16, OP_pk_fma_f16, uint32_t)
BOTH(swap16, OP_swap16, uint32_t)
BOTH(swap32, OP_swap32, uint32_t)
BOTH(ds_read, OP_ds_read, uint32_t)
BOTH(dot2_add, OP_dot2_add, uint32_t)

// lk_kernel's pixel-work mix, 44 instructions per group (independent within the group of 8 accumulators)
#define LKMIX(OPD)                                                                                 \
    OP_dot2(a0, x, y); OP_dot2(a1, x, y); OP_perm(a2, x, y); OP_dot2(a3, x, y); OP_dot2(a4, x, y); \
    OP_perm(a5, x, y); OP_dot2(a6, x, y); OP_dot2(a7, x, y); OP_alignbyte(a0, x, y);               \
    OP_dot2(a1, x, y); OP_dot2(a2, x, y); OP_perm(a3, x, y); OP_dot2(a4, x, y); OP_dot2(a5, x, y); \
    OP_perm(a6, x, y); OP_pk_sub(a7, x, y); OP_dot2(a0, x, y); OP_dot2(a1, x, y); OP_perm(a2, x, y); \
    OP_dot2(a3, x, y); OP_dot2(a4, x, y); OP_perm(a5, x, y); OP_alignbyte(a6, x, y);               \
    OP_dot2(a7, x, y); OP_dot2(a0, x, y); OP_perm(a1, x, y); OP_pk_sub(a2, x, y); OP_dot2(a3, x, y); \
    OP_dot2(a4, x, y); OP_perm(a5, x, y); OP_pk_lshr(a6, x, y); OP_dot2(a7, x, y); OP_dot2(a0, x, y); \
    OP_perm(a1, x, y); OP_alignbyte(a2, x, y); OP_pk_sub(a3, x, y); OP_dot2(a4, x, y);             \
    OP_dot2(a5, x, y); OP_perm(a6, x, y); OP_pk_lshr(a7, x, y); OP_alignbyte(a0, x, y);            \
    OP_pk_sub(a1, x, y); OP_pk_lshr(a2, x, y); OP_dot2(a3, x, y);
constexpr int kMixLen = 44;
__global__ __launch_bounds__(1024) void k_lk_mix(uint64_t *stamps, uint32_t *sink, uint32_t xin, uint32_t yin)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t x = xin, y = yin;
    __syncthreads();
    const uint64_t t0 = __builtin_readcyclecounter();
#pragma nounroll
    for (int it = 0; it < kIters; it++) {
        LKMIX(0) LKMIX(0) LKMIX(0)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        stamps[2 * wv] = t0; stamps[2 * wv + 1] = t1;
    }
    uint32_t r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (r == 0x7f123456u) sink[0] = r;
}

struct Row { const char *name; void *fn; int instr_per_wave; int kind; };   // kind 0 u32, 1 f32, 2 f64

template <typename T, typename K>
static int run_one(const char *name, K kernel, int instr_per_wave, T x, T y, uint64_t *d_stamps, void *d_sink,
                   int n_cu, double *out_rate)
{
    std::vector<uint64_t> h;
    for (int wps : {1, 2, 4, 8}) {
        const int waves_per_cu = 4 * wps;
        const int threads = std::min(1024, 64 * waves_per_cu);
        const int blocks = n_cu * (64 * waves_per_cu / threads);
        const int n_waves = blocks * threads / 64;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) {            // second launch is the measured one
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), 0, 0, d_stamps, (T *)d_sink, x, y);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        h.resize(2 * (size_t)n_waves);
        CK(hipMemcpy(h.data(), d_stamps, sizeof(uint64_t) * 2 * n_waves, hipMemcpyDeviceToHost));
        // per workgroup: span from the first start to the last end; all its waves ran concurrently
        const int wpb = threads / 64;
        std::vector<double> spans;
        for (int b = 0; b < blocks; b++) {
            uint64_t s = ~0ull, e = 0;
            for (int w = 0; w < wpb; w++) { s = std::min(s, h[2 * (b * wpb + w)]); e = std::max(e, h[2 * (b * wpb + w) + 1]); }
            spans.push_back((double)(e - s));
        }
        std::sort(spans.begin(), spans.end());
        const double med = spans[spans.size() / 2];
        // waves of one workgroup sharing a SIMD: wpb / 4 (workgroups of >= 4 waves spread over the 4 SIMDs)
        const double share = wpb >= 4 ? wpb / 4.0 : 1.0;
        // with 8 waves per SIMD two workgroups share the CU: their streams interleave, so the per-SIMD
        // rate doubles relative to what one workgroup's span shows
        const double wg_per_cu = (double)(64 * waves_per_cu) / threads;
        const double rate = share * wg_per_cu * instr_per_wave / med;
        const double wall_rate = (double)n_waves * instr_per_wave / (ms * 1e-3) / (n_cu * 4.0);   // wave-instr / s / SIMD
        printf("%-14s waves/SIMD %d : %7.0f cycles  %.3f wave-instr/cycle/SIMD (%.2f cycles per instr)  wall %.3f ms = %.2f G wave-instr/s/SIMD\n",
               name, wps, med, rate, 1.0 / rate, ms, wall_rate * 1e-9);
        if (out_rate && wps == 4) *out_rate = rate;
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    return 0;
}

int main(int argc, char **argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 1;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d MHz; %d instructions per wave per stream\n", prop.gcnArchName, n_cu,
           prop.clockRate / 1000, kIters * kUnroll * 8);
    uint64_t *d_stamps; void *d_sink;
    CK(hipMalloc(&d_stamps, sizeof(uint64_t) * 2 * (size_t)n_cu * 32));
    CK(hipMalloc(&d_sink, 64));
    const int N = kIters * kUnroll * 8;
#define RUN(NAME, T, X, Y)                                                                   \
    if (run_one<T>(#NAME, k_##NAME, N, (T)(X), (T)(Y), d_stamps, d_sink, n_cu, nullptr)) return 1; \
    if (run_one<T>(#NAME "_dep", k_##NAME##_dep, N, (T)(X), (T)(Y), d_stamps, d_sink, n_cu, nullptr)) return 1;
    if (batch == 1) {
    RUN(dot2, uint32_t, 0x00030005u, 0x00070002u)
    RUN(dot4, uint32_t, 0x01030005u, 0x02070002u)
    RUN(perm, uint32_t, 0x12345678u, 0x05040100u)
    RUN(alignbyte, uint32_t, 0x12345678u, 1u)
    RUN(pk_sub, uint32_t, 0x00010001u, 0)
    RUN(pk_add, uint32_t, 0x00010001u, 0)
    RUN(pk_mul, uint32_t, 0x00030003u, 0)
    RUN(pk_mad, uint32_t, 0x00030003u, 0x00010001u)
    RUN(pk_lshr, uint32_t, 0, 0)
    RUN(add_u32, uint32_t, 3, 0)
    RUN(add3, uint32_t, 3, 5)
    RUN(and_or, uint32_t, 0x0ffffff0u, 5)
    RUN(lshl_add, uint32_t, 3, 0)
    RUN(ashr, uint32_t, 0, 0)
    RUN(mul_lo, uint32_t, 3, 0)
    RUN(mad_u24, uint32_t, 3, 5)
    RUN(fma_f32, float, 1.0001f, 0.5f)
    RUN(mul_f32, float, 1.0001f, 0)
    RUN(cvt_i2f, uint32_t, 0, 0)
    RUN(rndne, float, 0, 0)
    RUN(dpp_add, uint32_t, 0, 0)
    RUN(mov_dpp, uint32_t, 7, 0)
    RUN(cndmask, uint32_t, 7, 0)
    RUN(bcnt, uint32_t, 0xf0f0f0f0u, 0)
    RUN(min3, uint32_t, 9, 11)
    RUN(sqrt_f32, float, 0, 0)
    RUN(rcp_f32, float, 0, 0)
    RUN(fma_f64, double, 1.0001, 0)
    if (run_one<uint32_t>("lk_mix", k_lk_mix, kIters * 3 * kMixLen, 0x00030005u, 0x05040100u, d_stamps, d_sink, n_cu, nullptr)) return 1;
    } else {
    RUN(dot2c, uint32_t, 0x00030005u, 0x00070002u)
    RUN(dot4c, uint32_t, 0x01030005u, 0x02070002u)
    RUN(fmac_f32, float, 1.0001f, 0.5f)
    RUN(add_f32, float, 1.0f, 0)
    RUN(mul_i24, uint32_t, 3, 0)
    RUN(mul_u24, uint32_t, 3, 0)
    RUN(mad_i24, uint32_t, 3, 5)
    RUN(and, uint32_t, 0x0ffffff0u, 0)
    RUN(or, uint32_t, 0x10u, 0)
    RUN(xor, uint32_t, 0x10u, 0)
    RUN(lshl, uint32_t, 0, 0)
    RUN(lshr, uint32_t, 0, 0)
    RUN(sub_u32, uint32_t, 3, 0)
    RUN(max_i32, uint32_t, 3, 0)
    RUN(min_u32, uint32_t, 3, 0)
    RUN(mov, uint32_t, 3, 0)
    RUN(mov_sgpr, uint32_t, 3, 0)
    RUN(add_sgpr, uint32_t, 3, 0)
    RUN(dot2_sgpr, uint32_t, 0x00030005u, 0x00070002u)
    RUN(cvt_f2i, uint32_t, 0, 0)
    RUN(floor, float, 0, 0)
    RUN(cvt_ub0, uint32_t, 0, 0)
    RUN(cmp_vcc, uint32_t, 3, 0)
    RUN(cmp_sgpr, uint32_t, 3, 0)
    RUN(cnd_vcc0, uint32_t, 7, 0)
    RUN(cnd_s, uint32_t, 7, 0)
    RUN(readlane, uint32_t, 0, 0)
    RUN(readfl, uint32_t, 0, 0)
    RUN(add_sdwa, uint32_t, 0x01020304u, 0)
    RUN(mul_sdwa, uint32_t, 0x01020304u, 0)
    RUN(add_u16, uint32_t, 3, 0)
    RUN(mul_lo_u16, uint32_t, 3, 0)
    RUN(mad_u16, uint32_t, 3, 5)
    RUN(lshl_or, uint32_t, 1, 0)
    RUN(bfe, uint32_t, 0, 0)
    RUN(bfi, uint32_t, 0x00ff00ffu, 0x12345678u)
    RUN(alignbit, uint32_t, 0x12345678u, 0)
    RUN(pk_fma_f32, double, 1.0, 0)
    RUN(pk_fma_f16, uint32_t, 0x3c003c00u, 0x38003800u)
    RUN(swap16, uint32_t, 3, 0)
    RUN(swap32, uint32_t, 3, 0)
    RUN(ds_read, uint32_t, 64, 64)
    RUN(dot2_add, uint32_t, 0x00030005u, 0x00070002u)
    }
    return 0;
}

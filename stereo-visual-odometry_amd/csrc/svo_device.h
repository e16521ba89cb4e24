// svo_device.h -- shared device-side definitions for the gfx950 stereo-VO kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace svo {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kPad = 32;           // border kept around every pyramid level (>= LK win + tile slack)
constexpr int kMaxLevels = 4;      // LK maxLevel 3 (reference src/tracking.cpp:595)
constexpr int kWin = 21;           // LK window
constexpr int kLkMaxIter = 30;

// Geometry of one padded pyramid slot; identical for every image of a context.
struct PyrGeom {
    int nlevels;
    int w[kMaxLevels], h[kMaxLevels];
    int pitch[kMaxLevels];          // bytes per padded row (multiple of 64)
    int64_t origin[kMaxLevels];     // byte offset of pixel (0,0) of level l inside the slot
    int64_t slot_bytes;             // slot size (multiple of 256)
};

__host__ __device__ inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

// ---- wave64 integer reductions ------------------------------------------------------------
// DPP butterfly inside each row of 16 lanes, then the four row sums are combined through
// readlane (SALU).  The result is wave-uniform.
__device__ inline int row_allsum_i32(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror
    return v;
}

__device__ inline int wave_sum_i32(int v)
{
    v = row_allsum_i32(v);
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// Exact 64-lane sum of int32 partials whose total may exceed 32 bits: the low 16 bits and the
// (signed) high 16 bits are reduced separately and recombined in 64-bit.
__device__ inline long long wave_sum_i32_wide(int p)
{
    int lo = wave_sum_i32(p & 0xFFFF);
    int hi = wave_sum_i32(p >> 16);
    return ((long long)hi << 16) + (long long)lo;
}

// LDS traffic of ONE wave is ordered by the hardware queue; this only stops the compiler from
// moving LDS accesses across the hand-off between lanes of the same wave.
__device__ inline void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace svo

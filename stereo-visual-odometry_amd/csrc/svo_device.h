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

// Geometry of one ORB pyramid slot (8 levels, scale 1.2) + the per-level tables of ORBextractor.
constexpr int kOrbMaxLevels = 8;
struct OrbGeom {
    int nlevels;
    int w[kOrbMaxLevels], h[kOrbMaxLevels], pitch[kOrbMaxLevels];
    int64_t origin[kOrbMaxLevels];       // byte offset of pixel (0,0) of level l inside the slot
    int64_t blur_off[kOrbMaxLevels];     // offset of level l in the blurred images (rows bpitch[l] bytes apart, 4-byte aligned)
    int bpitch[kOrbMaxLevels];
    int64_t slot_bytes, blur_total;
    float scale[kOrbMaxLevels];          // mvScaleFactor
    int quota[kOrbMaxLevels];            // mnFeaturesPerLevel
    int umax[16];
    int gk[7];                           // integer 7-tap Gaussian, sigma 2, scale 256
    int nCols[kOrbMaxLevels], nRows[kOrbMaxLevels], wCell[kOrbMaxLevels], hCell[kOrbMaxLevels];
    int ncell[kOrbMaxLevels], cell_off[kOrbMaxLevels], cells_total;
    // the cell-FAST launch: a workgroup takes gcell[l] consecutive cells of a cell row (gcols[l] workgroups per row)
    int gcell[kOrbMaxLevels], gcols[kOrbMaxLevels], blk_off[kOrbMaxLevels], blks_total;
    int xtab_off[kOrbMaxLevels], ytab_off[kOrbMaxLevels], xtab_total, ytab_total;   // resize tables (levels >= 1)
    int rs_stream[kOrbMaxLevels];        // level l is produced by the row-streaming resize kernel (orb_resize_stream_levels)
    // the blur launch covers ALL levels: first block (y) of level l
    int blur_blk[kOrbMaxLevels + 1];
};

// Level 0 of the ORB pyramid read IN PLACE from the input images (the batched and online paths, whose frames stay put until the
// step's kernels are done): left / right images interleave into the image slots 2f, 2f + 1.  img == null: level 0 was copied
// into the slot (stage API, inputs that cannot be read as aligned dwords).
struct OrbL0 { const uint8_t *img, *img2; int pitch; int64_t stride; };
__device__ __forceinline__ const uint8_t *orb_level0(const OrbL0 &z, int b)
{
    return z.img2 ? ((b & 1) ? z.img2 : z.img) + (int64_t)(b >> 1) * z.stride : z.img + (int64_t)b * z.stride;
}

// inclusive prefix sum over the wave's lanes on the DPP network (row shifts inside the rows of 16, then the two row
// broadcasts: the sequence of LLVM's AMDGPUAtomicOptimizer for gfx9) -- six adds, no LDS round trips
__device__ __forceinline__ int wave_incl_scan(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 into rows 2 and 3
    return v;
}
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

// ---- several sums at once ---------------------------------------------------------------------
// Reducing k values with k independent butterflies costs k*log2(64) cross-lane adds.  Instead the
// first log2(k) steps exchange HALF of the values with the partner lane (each lane keeps one class
// of values and gives the other away), so afterwards every lane carries ONE value whose class is
// lane % k; the remaining steps use class-preserving moves: row_ror:4/8 inside a row of 16 and the
// CDNA4 v_permlane16_swap / v_permlane32_swap across rows.
__device__ inline int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
__device__ inline int dpp_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }
__device__ inline int dpp_ror4(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true); }
__device__ inline int dpp_ror8(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true); }

// every lane ends with the sum over all lanes congruent to it mod 4
__device__ inline int allsum_mod4(int v)
{
    v += dpp_ror4(v);
    v += dpp_ror8(v);
    auto a = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    v = (int)(a[0] + a[1]);
    auto b = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    return (int)(b[0] + b[1]);
}

// Reduce-scatter of 8 per-lane values v[2*slot + t] (slot 0..3, t 0..1) whose 64-lane totals may
// exceed 32 bits (each |v| < 2^28).  The SLOT is scattered over the four rows of 16 lanes with the
// CDNA4 row exchanges: v_permlane32_swap / v_permlane16_swap hand the partner's half of a value
// pair over in place, so a scatter step is swap + add -- no selects.  t is scattered over lane bit 0
// (two selects + one DPP add); the 8-lane sums, still exact in 32 bits, are then split into 16-bit
// halves, the half is scattered over lane bit 1, and the remaining two lane bits (the four quads
// of a row) are plain DPP adds.  On return EVERY quad of row `slot` holds that slot's
//   {t0.lo, t1.lo, t0.hi, t1.hi}       (lo: low 16 bits, hi: arithmetic high part of the 8-lane sums)
// 6 swaps + 16 other cross-lane/select ops (the bank-scattered version this replaces took 37).
__device__ inline int reduce_scatter8_rows(const int (&v)[8], int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    int a[2][2], c[2];
#pragma unroll
    for (int s0 = 0; s0 < 2; s0++)
#pragma unroll
        for (int t = 0; t < 2; t++) {                      // lanes 0-31 keep slots 0,1; lanes 32-63 slots 2,3
            auto x = __builtin_amdgcn_permlane32_swap((unsigned)v[2 * s0 + t], (unsigned)v[2 * (s0 + 2) + t], false, false);
            a[s0][t] = (int)(x[0] + x[1]);
        }
#pragma unroll
    for (int t = 0; t < 2; t++) {                          // even rows keep the even slot of their half
        auto x = __builtin_amdgcn_permlane16_swap((unsigned)a[0][t], (unsigned)a[1][t], false, false);
        c[t] = (int)(x[0] + x[1]);
    }
    const int keep = b0 ? c[1] : c[0], give = b0 ? c[0] : c[1];
    const int d = keep + dpp_xor1(give);                   // t = lane bit 0; sums of 8 lanes
    const int lo = d & 0xFFFF, hi = d >> 16;
    const int keep2 = b1 ? hi : lo, give2 = b1 ? lo : hi;
    int e = keep2 + dpp_xor2(give2);                       // half = lane bit 1
    e += dpp_ror4(e);
    e += dpp_ror8(e);
    return e;
}

// The same result layout for ONE slot's two values (each |v| < 2^28) summed over all 64 lanes: {t0.lo, t1.lo, t0.hi, t1.hi}
// in every quad of EVERY row.  A third of lk_kernel's wave-iterations serve a single slot (its three partners have
// converged): 13 cross-lane / select operations with two row swaps instead of 22 with six.
__device__ inline int reduce_pair_all(int v0, int v1, int lane)
{
    const bool b0 = lane & 1, b1 = lane & 2;
    const int keep = b0 ? v1 : v0, give = b0 ? v0 : v1;
    int d = keep + dpp_xor1(give);                         // t = lane bit 0; sums of 2 lanes
    d += dpp_xor2(d);                                      // sums of 4 lanes (< 2^30), the same in lanes i and i ^ 2
    const int e = b1 ? d >> 16 : d & 0xFFFF;               // half = lane bit 1
    return allsum_mod4(e);                                 // over the 16 quads: |lo| < 2^20, |hi| < 2^18
}

// broadcast lane q of every quad to the whole quad
template <int Q>
__device__ inline int quad_bcast(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, Q * 0x55, 0xF, 0xF, true);
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

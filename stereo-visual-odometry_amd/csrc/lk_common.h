// lk_common.h -- pieces shared by the two pyramidal-LK kernels (lk.hip: exact integer sums, the canonical
// recipe; lk_sse2.hip: upstream's x86 float accumulation order): tile geometry, the packed dot-product helpers,
// the bilinear weights and the tile staging.  See lk.hip for the mapping these serve.
#pragma once
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

constexpr int kSlots = 4;                                 // points per wave
// I tile: 24 rows x 28 bytes of the level, staged as ROW-PAIR COLUMN WORDS: Q[p][c] = byte c of tile
// row p | byte c of row p + 1 << 16 (23 pairs x 28 columns, one dword each).  The patch build wants
// exactly these words for the row pairs (r, r+1), (r+1, r+2), (r+2, r+3) of every lane's 10 columns;
// formed while staging (4 v_perm per staged dword pair, 12 per lane) they replace the 30 v_perm +
// 12 v_alignbyte every lane spent on its own copy, and the lane's reads become plain dword reads.
// Stored COLUMN-MAJOR with 29 words per column, like the J tile below and for the same reason: a lane's
// reads of its row pairs (row, row+1, row+2) of column offI + 7 seg + j then fall on banks 11 seg + row + d
// (mod 32) -- conflict-free -- where the row-major order was 2-way conflicted for every row stride below 53
// (half of all LDS-array cycles of the patch build; SQ_LDS_BANK_CONFLICT was 30 % of SQ_LDS_IDX_ACTIVE).
constexpr int kQPairs = 23, kQCols = 28, kQColDw = 29, kQTileDw = kQCols * kQColDw;   // 812 dwords per slot
// J tile: the same row-pair column words, 27 pairs x 28 columns around the window (3 spare on every side:
// a window drifts that far at one level only rarely, and then the tile is staged again), stored
// COLUMN-MAJOR with 29 words per column: lane (row, seg) reads column cx + 7 seg + k, pair cy + row, so
// the banks of a 32-lane half are 7 * 29 * seg + row = 11 seg + row (mod 32) -- conflict-free; the
// row-major order is 2-way conflicted for every stride below 53.
constexpr int kJPairs = 27, kJCols = 28, kJColDw = 29, kJTileDw = kJCols * kJColDw;   // 812 dwords per slot
constexpr int kJMargin = 3;
static_assert(kQTileDw == kJTileDw && kQColDw == kJColDw, "slot s's J tile takes over slot s's I tile");
constexpr int kLdsDwPerWave = kSlots * kJTileDw;                                         // 3248 dwords
constexpr int W_BITS = 14;

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32_unaligned __attribute__((aligned(1)));
typedef const uint32_t __attribute__((address_space(3))) lds_cu32;

// cvFloor: one instruction (floor + convert; the compiler's __float2int_rd is v_floor_f32 + v_cvt_i32_f32,
// and on gfx950 conversions issue at half the rate of plain 32-bit adds -- profiles/r02_valu_roof.txt)
__device__ __forceinline__ int cv_floor(float v)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
// first link of a dot chain: the rounding constant comes from an SGPR through the VOP3P encoding
// (the VOP2 v_dot2c form accumulates in place and would need a v_mov of the constant every time)
__device__ __forceinline__ int dot2_k(uint32_t a, uint32_t b, int k)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k));
    return r;
}
// first link of a chain that starts from 0: the inline constant instead of a zeroed accumulator (v_mov + v_dot2c)
__device__ __forceinline__ int dot2_0(uint32_t a, uint32_t b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// exact (float)(hi * 65536 + lo) with round-to-nearest-even: |hi|, |lo| < 2^24 convert exactly
// and one fused multiply-add rounds the exact sum once (f64 conversions issue at a fraction of
// the f32 rate, and this runs every iteration)
__device__ __forceinline__ float wide_to_f32(int hi, int lo)
{
    return __builtin_fmaf((float)hi, 65536.f, (float)lo);
}
// The four bilinear weights as the two packed operands of the column-word dot products:
//   Wa = iw00 | iw10 << 16   (column tap k: window rows A | B),   Wb = iw01 | iw11 << 16   (tap k + 1)
// iw00 = cvRound((1-a)(1-b) 2^14) ...: the 2^14 scale is folded into the b factors first (scaling by a
// power of two is exact, so every product rounds exactly as upstream's expression does).  cvRound of
// 0 <= x <= 2^14 is taken with the 1.5 * 2^23 trick: x + 12582912.f rounds x to the nearest-even
// integer and leaves it in the low mantissa bits (two full-rate adds instead of v_rndne + v_cvt);
// the low 16 bits of the sum's bit pattern ARE the weight, so the packing is one v_perm.
struct PackedWeights { uint32_t Wa, Wb; };
__device__ __forceinline__ PackedWeights bilinear_weights(float a, float b)
{
    const float magic = 12582912.f;                            // 0x4B400000
    const float a1 = 1.f - a, b1 = (1.f - b) * (float)(1 << W_BITS), b0 = b * (float)(1 << W_BITS);
    const uint32_t u00 = __float_as_uint(a1 * b1 + magic), u01 = __float_as_uint(a * b1 + magic),
                   u10 = __float_as_uint(a1 * b0 + magic);     // 0x4B400000 + iw
    const uint32_t w11 = (uint32_t)(1 << W_BITS) + 3u * 0x4B400000u - u00 - u01 - u10;
    PackedWeights w;
    w.Wa = perm_b32(u10, u00, 0x05040100u);
    w.Wb = perm_b32(w11, u01, 0x05040100u);
    return w;
}

// "ix < -win || ix >= w || iy < -win || iy >= h" with two unsigned compares
__device__ __forceinline__ bool window_oob(int ix, int iy, int w, int h)
{
    return (unsigned)(ix + kWin) >= (unsigned)(w + kWin) || (unsigned)(iy + kWin) >= (unsigned)(h + kWin);
}

// ---- one iteration's pixel work for one slot --------------------------------------------------
// Column words C_j = (J[r0][j] | J[r1][j] << 16) pair the two window rows, so a bilinear sample is
//   val_k = dot2(C_k, (w00 | w10 << 16)) + dot2(C_k+1, (w01 | w11 << 16)) + 2^8
// (signed 16-bit weights: w11 == -1 needs no special case).
__device__ __forceinline__ int dot2_v(uint32_t a, uint32_t b, int c)      // a . b + c, c in a VGPR that stays live
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int dot2_sv(uint32_t a, uint32_t b_uniform, int c)   // b wave-uniform (SGPR), c in a VGPR
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

// A tile's source dwords for this lane: items lane + 64 t = (row pair, dword column), rows `rowA` (upper) and
// `rowB` = rowA + pitch, at 32-bit offsets from the wave-uniform base (global_load with an SGPR base)
__device__ __forceinline__ void tile_loads(uint32_t (&r)[3][2], const uint8_t *rowA, const uint8_t *rowB, uint32_t s_off,
                                           const uint32_t (&q_src)[3], int lane, int n_items)
{
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (lane + 64 * t < n_items) {
            const uint32_t o = s_off + q_src[t];
            r[t][0] = *(const u32_unaligned *)(rowA + o); r[t][1] = *(const u32_unaligned *)(rowB + o);
        }
    }
}
// ... and their four column words each into a column-major J tile
__device__ __forceinline__ void tile_store_j(uint32_t *tile, const uint32_t (&r)[3][2], const int (&jq_dst)[3], int lane)
{
#pragma unroll
    for (int t = 0; t < 3; t++) {
        if (lane + 64 * t < kJPairs * 7) {
            const uint32_t top = r[t][0], bot = r[t][1];
            uint32_t *d = tile + jq_dst[t];
            // samples are stored as pixel << 7 (byte into the high byte of its half, one packed shift): the
            // bilinear sums then come out scaled by 2^7 and "sum >> 9" is simply their high half
            const u16x2 one = {1, 1};
#pragma unroll
            for (int c = 0; c < 4; c++)
                d[c * kJColDw] = as_u32(as_u16x2(perm_b32(bot, top, 0x040c000cu + 0x01000100u * c)) >> one);
        }
    }
}

}  // namespace svo

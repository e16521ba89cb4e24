// lk.hip -- pyramidal iterative Lucas-Kanade tracking for gfx950: the body of
// cv::calcOpticalFlowPyrLK(prev, next, pts, out, status, err, Size(21,21), 3,
//                          TermCriteria(COUNT+EPS, 30, 0.01), 0, 0.001)
// as called four times per frame by Tracking::LK_Robust_Find_MuliImage_MatchedFeatures
// (reference src/tracking.cpp:583-622), plus the deleteBadmatchFeatures predicate (:623-660).
//
// Mapping: ONE WAVEFRONT PER POINT, four points per 256-thread workgroup, no workgroup barrier
// (each wave owns a private LDS region, so waves with different iteration counts never wait on
// each other).  With ncalls == 4 the same wave walks the whole circular chain
// L1 -> R1 -> R2 -> L2 -> L1' for its point, keeping the running point in registers.
//
// Per pyramid level (coarse to fine):
//   * the 24x24 source tile of I (window + bilinear + Scharr reach) is gathered from the padded
//     level with 4-byte aligned coalesced loads into LDS;
//   * lane l owns window row l/3, columns (l%3)*7..+6 (63 lanes x 7 px = 441 px).  It reads its
//     4 rows x 10 bytes of the tile, forms the Scharr derivatives of the 2x8 positions it
//     interpolates from ON THE FLY with packed 16-bit math (the reference path materialises a
//     full int16x2 derivative image per level per call; here it never exists, not even in LDS),
//     and interpolates I, Ix, Iy with v_dot2_i32_i16; the patch stays in VGPRs for all iterations;
//   * each iteration reads the lane's two 8-byte J row segments from an LDS-staged 40x32 J tile
//     (re-gathered only when the window drifts out of it), forms the four-tap fixed-point
//     bilinear samples with two v_dot4_u32_u8 per pixel (14-bit weights split into bytes), the
//     two mismatch sums with 24-bit MADs, and reduces them across the wave with DPP adds.
//
// Exactness: all pixel arithmetic is upstream's fixed point (14-bit weights, 5 fractional bits);
// the five sums A11,A12,A22,b1,b2 are accumulated as exact integers (per-lane int32 partials,
// 64-bit recombination) and converted to float once -- the canonical recipe of oracle/lk.c, so
// status bytes and point coordinates are bit-identical to the oracle.  FP contraction is off.
//
// Algorithmic HBM bytes (SURVEY.md 8d gather convention): per point per call
//   sum over 4 levels (24*24 + 22*22) + 8 in + 8 out + 1 status = 4257 B.
#include "svo_device.h"
#include <type_traits>
#include "svo_kernels.h"

namespace svo {

constexpr int kTileIRows = 24, kTileIDw = 8;              // 24 rows x 32 bytes (28 used)
constexpr int kTileJRows = 32, kTileJDw = 10;             // 32 rows x 40 bytes
constexpr int kLdsDwPerWave = kTileIRows * kTileIDw + kTileJRows * kTileJDw;   // 512 dwords
constexpr int W_BITS = 14;

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int cv_floor(float v) { return __float2int_rd(v); }
__device__ __forceinline__ uint32_t perm_b32(uint32_t s0, uint32_t s1, uint32_t sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ uint32_t alignbit16(uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); }
__device__ __forceinline__ int dot2(uint32_t a, uint32_t b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, a), __builtin_bit_cast(s16x2, b), c, false);
}
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }

// exact (float)(hi * 65536 + lo) with round-to-nearest-even, via one exact double
__device__ __forceinline__ float wide_to_f32(int hi, int lo)
{
    return (float)((double)hi * 65536.0 + (double)lo);
}
struct Weights { int w00, w01, w10, w11; };
// iw00 = cvRound((1-a)(1-b) 2^14) ...: the 2^14 scale is folded into the b factors first; scaling
// by a power of two is exact, so every product rounds exactly as upstream's expression does.
__device__ __forceinline__ Weights bilinear_weights(float a, float b)
{
    Weights w;
    const float a1 = 1.f - a, b1 = (1.f - b) * (float)(1 << W_BITS), b0 = b * (float)(1 << W_BITS);
    w.w00 = cv_round(a1 * b1);
    w.w01 = cv_round(a * b1);
    w.w10 = cv_round(a1 * b0);
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

// 12 aligned bytes starting at byte offset `off` of an LDS row of dwords: bytes 0-3, 4-7, 8-11
__device__ __forceinline__ void load12(const uint32_t *row, int off, uint32_t &lo, uint32_t &mid, uint32_t &hi)
{
    const uint32_t *p = row + (off >> 2);
    uint32_t d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3];
    int sh = off & 3;
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    mid = __builtin_amdgcn_alignbyte(d2, d1, sh);
    hi = __builtin_amdgcn_alignbyte(d3, d2, sh);
}
// 8 aligned bytes
__device__ __forceinline__ void load8(const uint32_t *row, int off, uint32_t &lo, uint32_t &hi)
{
    const uint32_t *p = row + (off >> 2);
    uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    int sh = off & 3;
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
}

// "ix < -win || ix >= w || iy < -win || iy >= h" with two unsigned compares
__device__ __forceinline__ bool window_oob(int ix, int iy, int w, int h)
{
    return (unsigned)(ix + kWin) >= (unsigned)(w + kWin) || (unsigned)(iy + kWin) >= (unsigned)(h + kWin);
}

// zero-extended u16 pairs (b0,b1),(b2,b3) of a dword
__device__ __forceinline__ uint32_t pair01(uint32_t x) { return perm_b32(0, x, 0x0c010c00u); }
__device__ __forceinline__ uint32_t pair23(uint32_t x) { return perm_b32(0, x, 0x0c030c02u); }

// One cv::calcOpticalFlowPyrLK call for one point, executed by one wave.
__device__ __forceinline__ void lk_call(const PyrGeom &g, const uint8_t *slotI, const uint8_t *slotJ, float2 prevPt,
                                     float2 &outPt, int &status, uint32_t *lds, int lane)
{
    uint32_t *tileI = lds;
    uint32_t *tileJ = lds + kTileIRows * kTileIDw;

    const int row = min(lane / 3, kWin - 1), seg = lane - (lane / 3) * 3;
    const bool lane_on = lane < 63;
    const float half = 10.f;                     // (winSize - 1) * 0.5
    const float FLT_SCALE = 1.f / (1 << 20);

    status = 1;
    float nx = 0.f, ny = 0.f;                    // nextPts[i]
    for (int level = g.nlevels - 1; level >= 0; --level) {
        const int w = g.w[level], h = g.h[level], pitch = g.pitch[level];
        const uint8_t *I = slotI + g.origin[level];
        const uint8_t *J = slotJ + g.origin[level];
        const float lscale = 1.f / (float)(1 << level);
        float px = prevPt.x * lscale, py = prevPt.y * lscale;
        if (level == g.nlevels - 1) { nx = px; ny = py; }
        else { nx = nx * 2.f; ny = ny * 2.f; }
        px -= half; py -= half;
        const int ipx = cv_floor(px), ipy = cv_floor(py);
        if (window_oob(ipx, ipy, w, h)) {
            if (level == 0) status = 0;
            continue;
        }
        const Weights wt = bilinear_weights(px - (float)ipx, py - (float)ipy);
        // (lane 63 carries no window pixel: zero weights make its I, Ix, Iy and sums vanish)
        const uint32_t W01 = lane_on ? ((uint32_t)wt.w00 | ((uint32_t)wt.w01 << 16)) : 0u;
        const uint32_t W23 = lane_on ? ((uint32_t)wt.w10 | ((uint32_t)wt.w11 << 16)) : 0u;

        // ---- gather the 24x24 I tile (rows ipy-1.., columns from the aligned x0 <= ipx-1)
        const int x0 = (ipx - 1) & ~3;
        const int offI = (ipx - 1) - x0;
        for (int i = lane; i < kTileIRows * kTileIDw; i += kWave) {
            int r = i >> 3, c = i & 7;
            uint32_t v = 0;
            if (c < 7) v = *(const uint32_t *)(I + (int64_t)(ipy - 1 + r) * pitch + x0 + 4 * c);
            tileI[i] = v;
        }
        wave_lds_fence();

        // ---- this lane's 7 patch pixels.  Tile rows row..row+3 = image rows ipy+row-1..ipy+row+2,
        //      bytes j = 0..9 = image columns ipx-1+seg*7+j.  Everything below is packed u16/i16
        //      pairs m = (column 2m, column 2m+1).
        int Iv[7], Ix[7], Iy[7];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        {
            uint32_t P[4][5];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                uint32_t lo, mid, hi;
                load12(tileI + (row + r) * kTileIDw, offI + seg * 7, lo, mid, hi);
                P[r][0] = pair01(lo); P[r][1] = pair23(lo); P[r][2] = pair01(mid); P[r][3] = pair23(mid);
                P[r][4] = pair01(hi);
            }
            // vertical Scharr passes for derivative rows A (image row ipy+row) and B (ipy+row+1)
            uint32_t T0A[5], T1A[5], T0B[5], T1B[5];
#pragma unroll
            for (int m = 0; m < 5; m++) {
                u16x2 p0 = as_u16x2(P[0][m]), p1 = as_u16x2(P[1][m]), p2 = as_u16x2(P[2][m]), p3 = as_u16x2(P[3][m]);
                const u16x2 k3 = {3, 3}, k10 = {10, 10};
                T0A[m] = as_u32((p0 + p2) * k3 + p1 * k10);
                T1A[m] = as_u32(p2 - p0);
                T0B[m] = as_u32((p1 + p3) * k3 + p2 * k10);
                T1B[m] = as_u32(p3 - p1);
            }
            // horizontal passes: derivative positions c = 0..7 (image column ipx+seg*7+c) as pairs
            uint32_t DXA[4], DYA[4], DXB[4], DYB[4];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const u16x2 k3 = {3, 3}, k10 = {10, 10};
                DXA[m] = as_u32(as_u16x2(T0A[m + 1]) - as_u16x2(T0A[m]));
                DXB[m] = as_u32(as_u16x2(T0B[m + 1]) - as_u16x2(T0B[m]));
                u16x2 qa = as_u16x2(alignbit16(T1A[m + 1], T1A[m]));
                u16x2 qb = as_u16x2(alignbit16(T1B[m + 1], T1B[m]));
                DYA[m] = as_u32((as_u16x2(T1A[m]) + as_u16x2(T1A[m + 1])) * k3 + qa * k10);
                DYB[m] = as_u32((as_u16x2(T1B[m]) + as_u16x2(T1B[m + 1])) * k3 + qb * k10);
            }
            // the derivative image's border is BORDER_CONSTANT 0: mask positions outside the image
            // (only possible when the window hangs over the edge)
            if (ipx < 0 || ipx + kWin >= w || ipy < 0 || ipy + kWin >= h) {
                const int gyA = ipy + row, gyB = gyA + 1;
                const uint32_t rowA = (gyA >= 0 && gyA < h) ? 0xFFFFFFFFu : 0u;
                const uint32_t rowB = (gyB >= 0 && gyB < h) ? 0xFFFFFFFFu : 0u;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    const int gx = ipx + seg * 7 + 2 * m;
                    uint32_t cm = ((gx >= 0 && gx < w) ? 0x0000FFFFu : 0u) | ((gx + 1 >= 0 && gx + 1 < w) ? 0xFFFF0000u : 0u);
                    DXA[m] &= cm & rowA; DYA[m] &= cm & rowA;
                    DXB[m] &= cm & rowB; DYB[m] &= cm & rowB;
                }
            }
#pragma unroll
            for (int k = 0; k < 7; k++) {
                const int m = k >> 1;
                uint32_t dxa, dya, dxb, dyb, i1, i2;
                if ((k & 1) == 0) {
                    dxa = DXA[m]; dya = DYA[m]; dxb = DXB[m]; dyb = DYB[m];
                    // intensity bytes j = k+1, k+2: an odd-aligned pair
                    i1 = alignbit16(P[1][m + 1], P[1][m]);
                    i2 = alignbit16(P[2][m + 1], P[2][m]);
                } else {
                    dxa = alignbit16(DXA[m + 1], DXA[m]); dya = alignbit16(DYA[m + 1], DYA[m]);
                    dxb = alignbit16(DXB[m + 1], DXB[m]); dyb = alignbit16(DYB[m + 1], DYB[m]);
                    i1 = P[1][m + 1]; i2 = P[2][m + 1];
                }
                int ival = dot2(i2, W23, dot2(i1, W01, 1 << (W_BITS - 5 - 1))) >> (W_BITS - 5);
                int ixv = dot2(dxb, W23, dot2(dxa, W01, 1 << (W_BITS - 1))) >> W_BITS;
                int iyv = dot2(dyb, W23, dot2(dya, W01, 1 << (W_BITS - 1))) >> W_BITS;
                Iv[k] = ival; Ix[k] = ixv; Iy[k] = iyv;
                pA11 += __mul24(ixv, ixv); pA12 += __mul24(ixv, iyv); pA22 += __mul24(iyv, iyv);
            }
        }
        int s11l, s11h, s12l, s12h, s22l, s22h;
        wave_sum4_i32(lane, pA11 & 0xFFFF, pA11 >> 16, pA12 & 0xFFFF, pA12 >> 16, s11l, s11h, s12l, s12h);
        wave_sum2_i32(lane, pA22 & 0xFFFF, pA22 >> 16, s22l, s22h);
        const float A11 = wide_to_f32(s11h, s11l) * FLT_SCALE;
        const float A12 = wide_to_f32(s12h, s12l) * FLT_SCALE;
        const float A22 = wide_to_f32(s22h, s22l) * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                             (float)(2 * kWin * kWin);
        if (minEig < 0.001f || D < 1.1920929e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;

        // ---- iterations
        float qx = nx - half, qy = ny - half;       // nextPt - halfWin
        float pdx = 0.f, pdy = 0.f;
        int tx0 = -(1 << 20), ty0 = 0;                // no J tile staged yet
        for (int j = 0; j < kLkMaxIter; j++) {
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (window_oob(inx, iny, w, h)) {
                if (level == 0) status = 0;
                break;
            }
            Weights wj = bilinear_weights(qx - (float)inx, qy - (float)iny);
            // w11 = 2^14 - (three rounded products) can come out as -1 (never lower); the byte-plane
            // dot products need non-negative weights, so that case is corrected with J11 afterwards
            const bool w11neg = wj.w11 < 0;
            if (w11neg) wj.w11 = 0;
            // weights as byte planes for v_dot4_u32_u8: taps (J00, J01, J10, J11)
            const uint32_t p01 = (uint32_t)wj.w00 | ((uint32_t)wj.w01 << 16);
            const uint32_t p23 = (uint32_t)wj.w10 | ((uint32_t)wj.w11 << 16);
            const uint32_t WL = perm_b32(p23, p01, 0x06040200u);
            const uint32_t WH = perm_b32(p23, p01, 0x07050301u);
            int cx = inx - tx0, cy = iny - ty0;
            if ((unsigned)cx > 17u || (unsigned)cy > 10u) {
                tx0 = (inx - 8) & ~3; ty0 = iny - 5;
                for (int i = lane; i < kTileJRows * kTileJDw; i += kWave) {
                    int r = i / kTileJDw, c = i - r * kTileJDw;
                    tileJ[i] = *(const uint32_t *)(J + (int64_t)(ty0 + r) * pitch + tx0 + 4 * c);
                }
                wave_lds_fence();
                cx = inx - tx0; cy = iny - ty0;
            }
            uint32_t a0, b0, a1, b1;                 // row 0 / row 1: bytes 0-3 (a), 4-7 (b)
            load8(tileJ + (cy + row) * kTileJDw, cx + seg * 7, a0, b0);
            load8(tileJ + (cy + row + 1) * kTileJDw, cx + seg * 7, a1, b1);
            const uint32_t m0 = __builtin_amdgcn_alignbyte(b0, a0, 2);      // bytes 2-5
            const uint32_t m1 = __builtin_amdgcn_alignbyte(b1, a1, 2);
            int pb1 = 0, pb2 = 0;
            auto pixels = [&](auto neg) {
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    // T = (J[r0][k], J[r0][k+1], J[r1][k], J[r1][k+1])
                    uint32_t T;
                    if (k < 3) T = perm_b32(a1, a0, 0x05040100u + 0x01010101u * k);
                    else if (k == 3) T = perm_b32(m1, m0, 0x06050201u);
                    else T = perm_b32(b1, b0, 0x05040100u + 0x01010101u * (k - 4));
                    uint32_t vlo = __builtin_amdgcn_udot4(T, WL, 1u << (W_BITS - 5 - 1), false);
                    uint32_t vhi = __builtin_amdgcn_udot4(T, WH, 0u, false);
                    uint32_t val = (vhi << 8) + vlo;
                    if (decltype(neg)::value) val -= T >> 24;        // w11 == -1
                    int diff = (int)(val >> (W_BITS - 5)) - Iv[k];
                    pb1 += __mul24(diff, Ix[k]);
                    pb2 += __mul24(diff, Iy[k]);
                }
            };
            if (__builtin_expect(w11neg, 0)) pixels(std::true_type{});
            else pixels(std::false_type{});
            int s1l, s1h, s2l, s2h;
            wave_sum4_i32(lane, pb1 & 0xFFFF, pb1 >> 16, pb2 & 0xFFFF, pb2 >> 16, s1l, s1h, s2l, s2h);
            const float b1f = wide_to_f32(s1h, s1l) * FLT_SCALE;
            const float b2f = wide_to_f32(s2h, s2l) * FLT_SCALE;
            const float dlx = (A12 * b2f - A22 * b1f) * D;
            const float dly = (A12 * b1f - A11 * b2f) * D;
            qx += dlx; qy += dly;
            nx = qx + half; ny = qy + half;
            if ((double)dlx * (double)dlx + (double)dly * (double)dly <= 0.01 * 0.01) break;
            // "std::abs(delta.x + prevDelta.x) < 0.01" compares a float with the double 0.01; the
            // largest float below 0.01 is 0.01f itself, so "<= 0.01f" in float is the same predicate
            if (j > 0 && fabsf(dlx + pdx) <= 0.01f && fabsf(dly + pdy) <= 0.01f) {
                nx -= dlx * 0.5f; ny -= dly * 0.5f;
                break;
            }
            pdx = dlx; pdy = dly;
        }
        if (status && level == 0) {
            // err is requested by the reference: the final window must still be inside (A.4 step 7)
            int fx = cv_floor(nx - half), fy = cv_floor(ny - half);
            if (window_oob(fx, fy, w, h)) status = 0;
        }
    }
    outPt = make_float2(nx, ny);
}

__global__ __launch_bounds__(256) void lk_kernel(LkArgs a)
{
    __shared__ uint32_t lds[4 * kLdsDwPerWave];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + wave;
    int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    n = min(n, a.cap);
    if (idx >= n) return;
    uint32_t *my = lds + wave * kLdsDwPerWave;
    const int64_t po = (int64_t)b * a.pts_stride + idx;
    const float2 p0 = a.pts_in[po];
    float2 cur = p0, nxt;
    bool outside = p0.x < 0 || p0.y < 0, bad = false, noepi = false;
    float prev_y = p0.y;
#pragma nounroll
    for (int c = 0; c < a.ncalls; c++) {
        const uint8_t *sI = a.prev[c] + (int64_t)b * a.slot_stride;
        const uint8_t *sJ = a.next[c] + (int64_t)b * a.slot_stride;
        int st;
        lk_call(a.g, sI, sJ, cur, nxt, st, my, lane);
        if (lane == 0) {
            a.pts_out[c][po] = nxt;
            a.status[c][po] = (uint8_t)st;
        }
        // Tracking::deleteBadmatchFeatures terms (p0 = t1_left, p1 = t1_right, p2 = t2_right,
        // p3 = t2_left, p0_return = LK#4 output; call-site mapping src/tracking.cpp:619-620)
        outside = outside || nxt.x < 0 || nxt.y < 0;
        bad = bad || st == 0;
        if (c == 0 || c == 2) noepi = noepi || (double)fabsf(prev_y - nxt.y) > a.match_err;   // |y0-y1|, |y2-y3|
        prev_y = nxt.y;
        cur = nxt;
        // a rejected point can never be kept: the remaining calls of the circular chain only feed
        // the keep predicate (their pts_out/status entries are scratch in the fused mode)
        if (a.ncalls == 4 && (outside || bad || noepi)) break;
    }
    if (a.ncalls == 4 && lane == 0) a.keep[po] = !(outside || bad || noepi);
}

// Stable compaction (deleteBadmatchFeatures erases in place, preserving order): one workgroup of
// 1024 threads per batch item, ballot/popcount ranks inside waves, LDS scan across waves.
__global__ __launch_bounds__(1024) void compact_kernel(CompactArgs a)
{
    __shared__ int wave_tot[16];
    __shared__ int base_s;
    const int b = blockIdx.x;
    int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    n = min(n, a.cap);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t o = (int64_t)b * a.pts_stride;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int start = 0; start < n; start += 1024) {
        int i = start + threadIdx.x;
        bool k = i < n && a.keep[o + i] != 0;
        unsigned long long m = __ballot(k);
        int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wv] = __popcll(m);
        __syncthreads();
        int pre = 0, tot = 0;
        for (int q = 0; q < 16; q++) { int t = wave_tot[q]; if (q < wv) pre += t; tot += t; }
        int dst = base_s + pre + rank;
        if (k) {
#pragma unroll
            for (int c = 0; c < 4; c++) a.out[c][o + dst] = a.in[c][o + i];
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.m_out[b] = base_s;
}

void launch_lk(const LkArgs &a, int batch, int max_pts, hipStream_t st)
{
    if (max_pts <= 0 || batch <= 0) return;
    dim3 grid((max_pts + 3) / 4, batch, 1), blk(256, 1, 1);
    hipLaunchKernelGGL(lk_kernel, grid, blk, 0, st, a);
}

void launch_compact(const CompactArgs &a, int batch, hipStream_t st)
{
    if (batch <= 0) return;
    hipLaunchKernelGGL(compact_kernel, dim3(batch), dim3(1024), 0, st, a);
}

}  // namespace svo

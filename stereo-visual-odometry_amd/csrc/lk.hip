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
//   * the 22x22 Scharr derivative tile is computed ON THE FLY from it (the reference path builds
//     a full int16x2 derivative image per level per call; here it never exists in HBM);
//   * lane l owns window row l/3, columns (l%3)*7..+6 (63 lanes x 7 px = 441 px): its I, Ix, Iy
//     patch values stay in VGPRs for all iterations;
//   * each iteration reads the lane's two 8-byte J row segments from an LDS-staged 40x32 J tile
//     (re-gathered only when the window drifts out of it), forms the fixed-point bilinear
//     differences and the two mismatch sums, and reduces them across the wave with DPP adds.
//
// Exactness: all pixel arithmetic is upstream's fixed point (14-bit weights, 5 fractional bits);
// the five sums A11,A12,A22,b1,b2 are accumulated as exact integers (per-lane int32 partials,
// 64-bit recombination) and converted to float once -- the canonical recipe of oracle/lk.c, so
// status bytes and point coordinates are bit-identical to the oracle.  FP contraction is off.
//
// Algorithmic HBM bytes (SURVEY.md 8d gather convention): per point per call
//   sum over 4 levels (24*24 + 22*22) + 8 in + 8 out + 1 status = 4257 B.
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

constexpr int kTileIRows = 24, kTileIDw = 7;              // 24 rows x 28 bytes
constexpr int kDerivW = 22;                               // 22 x 22 (dx | dy << 16)
constexpr int kTileJRows = 32, kTileJDw = 10;             // 32 rows x 40 bytes
constexpr int kLdsDwPerWave = kTileIRows * kTileIDw + kDerivW * kDerivW + kTileJRows * kTileJDw;  // 972
constexpr int W_BITS = 14;

__device__ __forceinline__ int cv_round(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int cv_floor(float v) { return __float2int_rd(v); }
__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// 8 consecutive bytes starting at byte offset `off` of an LDS row of dwords
__device__ __forceinline__ void load8(const uint32_t *row, int off, uint32_t &lo, uint32_t &hi)
{
    const uint32_t *p = row + (off >> 2);
    uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
    int sh = off & 3;
    lo = __builtin_amdgcn_alignbyte(d1, d0, sh);
    hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
}
__device__ __forceinline__ int byte_of(uint32_t lo, uint32_t hi, int k)
{
    return k < 4 ? (int)((lo >> (8 * k)) & 0xFFu) : (int)((hi >> (8 * (k - 4))) & 0xFFu);
}

struct Weights { int w00, w01, w10, w11; };
__device__ __forceinline__ Weights bilinear_weights(float a, float b)
{
    Weights w;
    w.w00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << W_BITS));
    w.w01 = cv_round(a * (1.f - b) * (float)(1 << W_BITS));
    w.w10 = cv_round((1.f - a) * b * (float)(1 << W_BITS));
    w.w11 = (1 << W_BITS) - w.w00 - w.w01 - w.w10;
    return w;
}

// One cv::calcOpticalFlowPyrLK call for one point, executed by one wave.
__device__ void lk_call(const PyrGeom &g, const uint8_t *slotI, const uint8_t *slotJ, float2 prevPt,
                        float2 &outPt, int &status, uint32_t *lds, int lane)
{
    uint32_t *tileI = lds;
    uint32_t *deriv = lds + kTileIRows * kTileIDw;
    uint32_t *tileJ = deriv + kDerivW * kDerivW;
    const uint8_t *tileIb = (const uint8_t *)tileI;

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
        if (ipx < -kWin || ipx >= w || ipy < -kWin || ipy >= h) {
            if (level == 0) status = 0;
            continue;
        }
        Weights wt = bilinear_weights(px - (float)ipx, py - (float)ipy);

        // ---- gather the 24x24 I tile (rows ipy-1.., columns from the aligned x0 <= ipx-1)
        const int x0 = (ipx - 1) & ~3;
        const int offI = (ipx - 1) - x0;
        for (int i = lane; i < kTileIRows * kTileIDw; i += kWave) {
            int r = i / kTileIDw, c = i - r * kTileIDw;
            tileI[i] = *(const uint32_t *)(I + (int64_t)(ipy - 1 + r) * pitch + x0 + 4 * c);
        }
        wave_lds_fence();

        // ---- Scharr derivatives of the 22x22 positions (ipx + c, ipy + r); zero outside the image
        for (int i = lane; i < kDerivW * kDerivW; i += kWave) {
            int r = i / kDerivW, c = i - r * kDerivW;
            int gx = ipx + c, gy = ipy + r;
            uint32_t v = 0;
            if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
                const uint8_t *p = tileIb + r * (kTileIDw * 4) + offI + c;      // top-left of the 3x3
                int r00 = p[0], r01 = p[1], r02 = p[2];
                const uint8_t *q = p + kTileIDw * 4;
                int r10 = q[0], r12 = q[2];
                const uint8_t *s = q + kTileIDw * 4;
                int r20 = s[0], r21 = s[1], r22 = s[2];
                int t0m = (r00 + r20) * 3 + r10 * 10, t0p = (r02 + r22) * 3 + r12 * 10;
                int t1m = r20 - r00, t1c = r21 - r01, t1p = r22 - r02;
                int dx = t0p - t0m, dy = (t1p + t1m) * 3 + t1c * 10;
                v = ((uint32_t)dx & 0xFFFFu) | ((uint32_t)dy << 16);
            }
            deriv[i] = v;
        }
        wave_lds_fence();

        // ---- this lane's 7 patch pixels: I (5 fractional bits), Ix, Iy; exact A sums
        int Iv[7], Ix[7], Iy[7];
        int pA11 = 0, pA12 = 0, pA22 = 0;
        {
            uint32_t lo0, hi0, lo1, hi1;
            const int bo = offI + 1 + seg * 7;                 // window column 0 sits at tile byte offI+1
            load8(tileI + (row + 1) * kTileIDw, bo, lo0, hi0);
            load8(tileI + (row + 2) * kTileIDw, bo, lo1, hi1);
            const uint32_t *d0 = deriv + row * kDerivW + seg * 7;
            const uint32_t *d1 = d0 + kDerivW;
            uint32_t da = d0[0], db = d1[0];
#pragma unroll
            for (int k = 0; k < 7; k++) {
                uint32_t da1 = d0[k + 1], db1 = d1[k + 1];
                int ival = descale(byte_of(lo0, hi0, k) * wt.w00 + byte_of(lo0, hi0, k + 1) * wt.w01 +
                                   byte_of(lo1, hi1, k) * wt.w10 + byte_of(lo1, hi1, k + 1) * wt.w11,
                                   W_BITS - 5);
                int dx00 = (short)(da & 0xFFFF), dy00 = (int)da >> 16;
                int dx01 = (short)(da1 & 0xFFFF), dy01 = (int)da1 >> 16;
                int dx10 = (short)(db & 0xFFFF), dy10 = (int)db >> 16;
                int dx11 = (short)(db1 & 0xFFFF), dy11 = (int)db1 >> 16;
                int ixv = descale(dx00 * wt.w00 + dx01 * wt.w01 + dx10 * wt.w10 + dx11 * wt.w11, W_BITS);
                int iyv = descale(dy00 * wt.w00 + dy01 * wt.w01 + dy10 * wt.w10 + dy11 * wt.w11, W_BITS);
                if (!lane_on) { ixv = 0; iyv = 0; }
                Iv[k] = ival; Ix[k] = ixv; Iy[k] = iyv;
                pA11 += ixv * ixv; pA12 += ixv * iyv; pA22 += iyv * iyv;
                da = da1; db = db1;
            }
        }
        const float A11 = (float)wave_sum_i32_wide(pA11) * FLT_SCALE;
        const float A12 = (float)wave_sum_i32_wide(pA12) * FLT_SCALE;
        const float A22 = (float)wave_sum_i32_wide(pA22) * FLT_SCALE;
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
        int tx0 = 0, ty0 = 0;
        bool tile_ok = false;
        for (int j = 0; j < kLkMaxIter; j++) {
            const int inx = cv_floor(qx), iny = cv_floor(qy);
            if (inx < -kWin || inx >= w || iny < -kWin || iny >= h) {
                if (level == 0) status = 0;
                break;
            }
            Weights wj = bilinear_weights(qx - (float)inx, qy - (float)iny);
            int cx = inx - tx0, cy = iny - ty0;
            if (!tile_ok || cx < 0 || cx > 17 || cy < 0 || cy > 10) {
                tx0 = (inx - 8) & ~3; ty0 = iny - 5;
                for (int i = lane; i < kTileJRows * kTileJDw; i += kWave) {
                    int r = i / kTileJDw, c = i - r * kTileJDw;
                    tileJ[i] = *(const uint32_t *)(J + (int64_t)(ty0 + r) * pitch + tx0 + 4 * c);
                }
                wave_lds_fence();
                tile_ok = true;
                cx = inx - tx0; cy = iny - ty0;
            }
            uint32_t lo0, hi0, lo1, hi1;
            load8(tileJ + (cy + row) * kTileJDw, cx + seg * 7, lo0, hi0);
            load8(tileJ + (cy + row + 1) * kTileJDw, cx + seg * 7, lo1, hi1);
            int pb1 = 0, pb2 = 0;
#pragma unroll
            for (int k = 0; k < 7; k++) {
                int diff = descale(byte_of(lo0, hi0, k) * wj.w00 + byte_of(lo0, hi0, k + 1) * wj.w01 +
                                   byte_of(lo1, hi1, k) * wj.w10 + byte_of(lo1, hi1, k + 1) * wj.w11,
                                   W_BITS - 5) - Iv[k];
                pb1 += diff * Ix[k];
                pb2 += diff * Iy[k];
            }
            const float b1 = (float)wave_sum_i32_wide(pb1) * FLT_SCALE;
            const float b2 = (float)wave_sum_i32_wide(pb2) * FLT_SCALE;
            const float dlx = (A12 * b2 - A22 * b1) * D;
            const float dly = (A12 * b1 - A11 * b2) * D;
            qx += dlx; qy += dly;
            nx = qx + half; ny = qy + half;
            if ((double)dlx * (double)dlx + (double)dly * (double)dly <= 0.01 * 0.01) break;
            if (j > 0 && fabs((double)(dlx + pdx)) < 0.01 && fabs((double)(dly + pdy)) < 0.01) {
                nx -= dlx * 0.5f; ny -= dly * 0.5f;
                break;
            }
            pdx = dlx; pdy = dly;
        }
        if (status && level == 0) {
            // err is requested by the reference: the final window must still be inside (A.4 step 7)
            int fx = cv_floor(nx - half), fy = cv_floor(ny - half);
            if (fx < -kWin || fx >= w || fy < -kWin || fy >= h) status = 0;
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
    float2 p[kMaxChain + 1];
    int st[kMaxChain];
    p[0] = a.pts_in[po];
#pragma unroll
    for (int c = 0; c < kMaxChain; c++) {
        if (c < a.ncalls) {
            const uint8_t *sI = a.prev[c] + (int64_t)b * a.slot_stride;
            const uint8_t *sJ = a.next[c] + (int64_t)b * a.slot_stride;
            lk_call(a.g, sI, sJ, p[c], p[c + 1], st[c], my, lane);
            if (lane == 0) {
                a.pts_out[c][po] = p[c + 1];
                a.status[c][po] = (uint8_t)st[c];
            }
        }
    }
    if (a.ncalls == 4 && lane == 0) {
        // Tracking::deleteBadmatchFeatures: p0 = t1_left, p1 = t1_right, p2 = t2_right,
        // p3 = t2_left, p0_return = LK#4 output (call-site mapping src/tracking.cpp:619-620)
        bool outside = p[3].x < 0 || p[3].y < 0 || p[2].x < 0 || p[2].y < 0 || p[1].x < 0 || p[1].y < 0 ||
                       p[0].x < 0 || p[0].y < 0 || p[4].x < 0 || p[4].y < 0;
        bool bad = st[0] == 0 || st[1] == 0 || st[2] == 0 || st[3] == 0;
        bool noepi = (double)fabsf(p[0].y - p[1].y) > a.match_err ||
                     (double)fabsf(p[2].y - p[3].y) > a.match_err;
        a.keep[po] = !(outside || bad || noepi);
    }
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

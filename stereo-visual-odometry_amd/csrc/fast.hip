// fast.hip -- FAST-9/16 corner detection with 3x3 non-max suppression and row-major ordered
// output, for gfx950.  Replaces cv::FAST(img, kps, thr, true) at reference src/tracking.cpp:101
// (Tracking::Detect_OpenCVFASTFeatures).  All-integer: bit-exact against oracle/fast.c.
//
// Two launches per batch of images:
//   fast_score_kernel : one 64x16 pixel tile per 256-thread workgroup.  The raw tile (+4 halo)
//       is staged in LDS once; the segment test + corner score are evaluated for the tile plus
//       a 1-pixel ring (the NMS neighbours) in three passes over compacted LDS position lists
//       (quick reject -> arc test -> score), NMS is done out of LDS, the suppressed score map
//       (u8, 0 = no keypoint) goes to HBM with 64-byte coalesced row stores, and per-row
//       keypoint counts are accumulated with one integer atomic per (row, tile).
//   fast_emit_kernel  : one wave per image row; its output offset is the sum of the row counts
//       above it (<= 17 coalesced loads), so keypoints come out in cv::FAST's row-major order
//       without a sort or an append-atomic.
// Algorithmic HBM bytes per image: W*H read + W*H score write/read + 12*N written.
#include "svo_device.h"
#include "svo_kernels.h"

namespace svo {

constexpr int kTileW = 64, kTileH = 16, kHalo = 4;
constexpr int kRawW = kTileW + 2 * kHalo;   // 72
constexpr int kRawH = kTileH + 2 * kHalo;   // 24
constexpr int kScW = kTileW + 2;            // 66
constexpr int kScH = kTileH + 2;            // 18

// Bresenham circle, radius 3 (cv::FAST pattern 16), as byte offsets in the 72-wide LDS tile.
__device__ __forceinline__ void load_circle(const uint8_t *c, int d[16], int v)
{
    constexpr int P = kRawW;
    d[0] = v - c[3 * P];       d[1] = v - c[3 * P + 1];   d[2] = v - c[2 * P + 2];   d[3] = v - c[P + 3];
    d[4] = v - c[3];           d[5] = v - c[-P + 3];      d[6] = v - c[-2 * P + 2];  d[7] = v - c[-3 * P + 1];
    d[8] = v - c[-3 * P];      d[9] = v - c[-3 * P - 1];  d[10] = v - c[-2 * P - 2]; d[11] = v - c[-P - 3];
    d[12] = v - c[-3];         d[13] = v - c[P - 3];      d[14] = v - c[2 * P - 2];  d[15] = v - c[3 * P - 1];
}

// >= 9 contiguous set bits in a circular 16-bit mask
__device__ __forceinline__ bool has_arc9(unsigned m16)
{
    unsigned m = m16 | (m16 << 16);
    unsigned a = m & (m >> 1);
    unsigned b = a & (a >> 2);
    unsigned c = b & (b >> 4);
    unsigned e = c & (m >> 8);
    return (e & 0xFFFFu) != 0;
}

// The segment test runs as three passes of decreasing population, each over a DENSE list so no
// wave pays for work only a few of its lanes need (the kernel is VALU bound, not HBM bound):
//   quick : every position -- any 9-arc of the 16-circle contains at least one pixel of each
//           opposite pair (k, k+8), so a pixel whose pairs (0,8) and (4,12) cannot both be
//           "brighter" or both be "darker" is rejected after 4 loads (cv::FAST's high-speed test);
//           four pixels per thread in packed 16-bit arithmetic, see the kernel;
//   arc   : survivors of quick -- the full 16-pixel contiguous-arc test;
//   score : corners only -- cornerScore<16>.
__device__ __forceinline__ bool fast_arc(const uint8_t *c, int thr)
{
    int d[16];
    load_circle(c, d, c[0]);
    unsigned dark = 0, bright = 0;          // d > thr: pixel darker than centre; d < -thr: brighter
#pragma unroll
    for (int k = 0; k < 16; k++) {
        dark |= (unsigned)(d[k] > thr) << k;
        bright |= (unsigned)(d[k] < -thr) << k;
    }
    return has_arc9(dark) || has_arc9(bright);
}

// cornerScore<16> of a corner (>= thr): min / max over every 9-arc d[s..s+8] with three-input ops
__device__ __forceinline__ int fast_corner_score(const uint8_t *c, int thr)
{
    int d[16];
    load_circle(c, d, c[0]);
    int m3[16], x3[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        m3[i] = min(min(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
        x3[i] = max(max(d[i], d[(i + 1) & 15]), d[(i + 2) & 15]);
    }
    int a0 = thr, bmin = 255;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        a0 = max(a0, min(min(m3[i], m3[(i + 3) & 15]), m3[(i + 6) & 15]));
        bmin = min(bmin, max(max(x3[i], x3[(i + 3) & 15]), x3[(i + 6) & 15]));
    }
    const int b0 = min(-a0, bmin);
    return -b0 - 1;
}

// append the positions of the lanes with `flag` to an LDS list (order inside the list is free:
// results are scattered back by position)
__device__ __forceinline__ void list_push(bool flag, int pos, uint16_t *list, int *count)
{
    const unsigned long long m = __ballot(flag);
    if (m == 0) return;
    const int lane = threadIdx.x;                       // blockDim.x == 64: x is the lane
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(m));
    base = __builtin_amdgcn_readfirstlane(base);
    if (flag) list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint16_t)pos;
}

__global__ __launch_bounds__(256) void fast_score_kernel(FastArgs a)
{
    __shared__ __attribute__((aligned(16))) uint8_t raw[kRawH * kRawW];
    __shared__ __attribute__((aligned(4))) uint8_t sc[(kScH * kScW + 3) & ~3];
    __shared__ uint16_t cand[kScH * kScW], corners[kScH * kScW];
    __shared__ int n_cand, n_corner;
    const int b = blockIdx.z;
    const uint8_t *img = a.img + (int64_t)b * a.img_stride;
    uint8_t *score = a.score + (int64_t)b * a.score_stride;
    int *rowcount = a.rowcount + (int64_t)b * a.rowcount_stride;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    if (tid == 0) { n_cand = 0; n_corner = 0; }

    // stage raw tile + halo (72 x 24 bytes); out-of-image bytes are never used by a valid centre.
    // x0 - 4 is a multiple of 4, so with a 4-byte aligned image the tile is 18 dwords per row.
    if ((((uintptr_t)img | (uintptr_t)a.pitch) & 3) == 0) {
        uint32_t *raw32 = (uint32_t *)raw;
        for (int i = tid; i < kRawH * (kRawW / 4); i += 256) {
            int ry = i / (kRawW / 4), rx4 = i - ry * (kRawW / 4);
            int gx = x0 - kHalo + rx4 * 4, gy = y0 - kHalo + ry;
            uint32_t v = 0;
            if (gy >= 0 && gy < a.h && gx >= 0) {
                const uint8_t *src = img + (int64_t)gy * a.pitch + gx;
                if (gx + 3 < a.w) v = *(const uint32_t *)src;
                else for (int q = 0; q < 4; q++) if (gx + q < a.w) v |= (uint32_t)src[q] << (8 * q);
            }
            raw32[i] = v;
        }
    } else {
        for (int i = tid; i < kRawH * kRawW; i += 256) {
            int ry = i / kRawW, rx = i - ry * kRawW;
            int gx = x0 - kHalo + rx, gy = y0 - kHalo + ry;
            uint8_t v = 0;
            if (gx >= 0 && gx < a.w && gy >= 0 && gy < a.h) v = img[(int64_t)gy * a.pitch + gx];
            raw[i] = v;
        }
    }
    for (int i = tid; i < (int)sizeof(sc) / 4; i += 256) ((uint32_t *)sc)[i] = 0;
    __syncthreads();

    // quick test over the tile and its 1-pixel ring (the NMS neighbours), FOUR pixels per thread: one
    // aligned LDS dword of raw row ry and its neighbours three rows up / down and three columns left /
    // right (five dword reads instead of twenty byte reads), packed 16-bit arithmetic on the even and the
    // odd bytes:   alive <=> max( min(v - min(p0,p8), v - min(p4,p12)),  min(max(p0,p8) - v, max(p4,p12) - v) ) > thr
    {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        constexpr int RD = kRawW / 4;                          // 18 dwords per raw row
        const uint32_t *rawd = (const uint32_t *)raw;
        const s16x2 T1 = {(short)(a.thr + 1), (short)(a.thr + 1)};
        for (int i0 = 0; i0 < kScH * RD; i0 += 256) {
            const int i = i0 + tid;
            uint32_t m4 = 0;
            int sy = 0, sxb = 0;
            if (i < kScH * RD) {
                sy = i / RD;
                const int gq = i - sy * RD;
                sxb = 4 * gq - (kHalo - 1);                    // score-map x of the dword's first byte
                const uint32_t *r = rawd + (sy + kHalo - 1) * RD + gq;
                const uint32_t C = r[0], U = r[-3 * RD], D = r[3 * RD], Lf = gq > 0 ? r[-1] : 0u, Rt = gq < RD - 1 ? r[1] : 0u;
                const uint32_t Lv = __builtin_amdgcn_alignbyte(C, Lf, 1);    // x - 3 neighbours of the four pixels
                const uint32_t Rv = __builtin_amdgcn_alignbyte(Rt, C, 3);    // x + 3 neighbours
                uint32_t sgn[2];
#pragma unroll
                for (int hb = 0; hb < 2; hb++) {                          // even bytes (pixels 0, 2), odd bytes (1, 3)
                    auto half = [&](uint32_t w) { return __builtin_bit_cast(s16x2, (hb ? w >> 8 : w) & 0x00FF00FFu); };
                    const s16x2 v = half(C), u = half(U), d = half(D), l = half(Lv), rr = half(Rv);
                    const s16x2 mnA = __builtin_elementwise_min(u, d), mxA = __builtin_elementwise_max(u, d);
                    const s16x2 mnB = __builtin_elementwise_min(l, rr), mxB = __builtin_elementwise_max(l, rr);
                    const s16x2 dark = __builtin_elementwise_min(v - mnA, v - mnB);
                    const s16x2 bright = __builtin_elementwise_min(mxA - v, mxB - v);
                    sgn[hb] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(dark, bright) - T1);   // sign set <=> rejected
                }
                const uint32_t dead = ((sgn[0] >> 15) & 1u) | ((sgn[1] >> 14) & 2u) | ((sgn[0] >> 29) & 4u) | ((sgn[1] >> 28) & 8u);
                // positions inside the score map [0, 66) and inside the image's testable area [3, w-3) x [3, h-3)
                const int gy = y0 - 1 + sy, gxb = x0 - 1 + sxb;
                const int lo = max(max(0, -sxb), 3 - gxb), hi = min(min(4, kScW - sxb), a.w - 3 - gxb);
                const uint32_t inside = (hi > lo && gy >= 3 && gy < a.h - 3) ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
                m4 = ~dead & inside;
            }
            // survivors -> list (any order): one LDS atomic per wave and step
            const int lane = threadIdx.x;
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long b0 = __ballot(m4 & 1u), b1 = __ballot(m4 & 2u), b2 = __ballot(m4 & 4u), b3 = __ballot(m4 & 8u);
            const int c0 = __popcll(b0), c1 = __popcll(b1), c2 = __popcll(b2), c3 = __popcll(b3);
            if (c0 + c1 + c2 + c3) {
                int base = 0;
                if (lane == 0) base = atomicAdd(&n_cand, c0 + c1 + c2 + c3);
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = sy * kScW + sxb;
                if (m4 & 1u) cand[base + __popcll(b0 & lt)] = (uint16_t)pos;
                if (m4 & 2u) cand[base + c0 + __popcll(b1 & lt)] = (uint16_t)(pos + 1);
                if (m4 & 4u) cand[base + c0 + c1 + __popcll(b2 & lt)] = (uint16_t)(pos + 2);
                if (m4 & 8u) cand[base + c0 + c1 + c2 + __popcll(b3 & lt)] = (uint16_t)(pos + 3);
            }
        }
    }
    __syncthreads();
    const int nc = n_cand;
    for (int i0 = 0; i0 < nc; i0 += 256) {
        const int i = i0 + tid;
        const int pos = i < nc ? cand[i] : 0;
        const int sy = pos / kScW, sx = pos - sy * kScW;
        const bool corner = i < nc && fast_arc(&raw[(sy + kHalo - 1) * kRawW + sx + kHalo - 1], a.thr);
        if (a.nms) list_push(corner, pos, corners, &n_corner);
        else if (corner) sc[pos] = 1;
    }
    __syncthreads();
    if (a.nms) {
        const int nk = n_corner;
        for (int i = tid; i < nk; i += 256) {
            const int pos = corners[i];
            const int sy = pos / kScW, sx = pos - sy * kScW;
            sc[pos] = (uint8_t)fast_corner_score(&raw[(sy + kHalo - 1) * kRawW + sx + kHalo - 1], a.thr);
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < kTileH / 4; j++) {
        int ly = threadIdx.y + 4 * j, lx = threadIdx.x;
        int gx = x0 + lx, gy = y0 + ly;
        const uint8_t *p = &sc[(ly + 1) * kScW + lx + 1];
        int s = p[0];
        if (a.nms && s) {
            bool keep = s > p[-1] && s > p[1] && s > p[-kScW - 1] && s > p[-kScW] && s > p[-kScW + 1] &&
                        s > p[kScW - 1] && s > p[kScW] && s > p[kScW + 1];
            if (!keep) s = 0;
        }
        bool inimg = gx < a.w && gy < a.h;
        if (!inimg) s = 0;
        unsigned long long m = __ballot(s != 0);
        if (inimg) score[(int64_t)gy * a.spitch + gx] = (uint8_t)s;
        if (threadIdx.x == 0 && m && gy < a.h) atomicAdd(&rowcount[gy], __popcll(m));
    }
}

// One wave per row: ordered emission of (x, y, response).
__global__ __launch_bounds__(256) void fast_emit_kernel(FastArgs a)
{
    const int b = blockIdx.y;
    const int y = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= a.h) return;
    const uint8_t *score = a.score + (int64_t)b * a.score_stride + (int64_t)y * a.spitch;
    const int *rowcount = a.rowcount + (int64_t)b * a.rowcount_stride;
    int mine = rowcount[y];
    bool last = (y == a.h - 1);
    if (mine == 0 && !last) return;
    // exclusive prefix over the rows above
    int part = 0;
    for (int r = lane; r < y; r += 64) part += rowcount[r];
    int off = wave_sum_i32(part);
    if (last && lane == 0) a.n_out[b] = off + mine;
    if (mine == 0) return;
    float2 *xy = a.kp_xy + (int64_t)b * a.kp_stride;
    float *resp = a.kp_resp + (int64_t)b * a.kp_stride;
    for (int x = lane; x < ((a.w + 63) & ~63); x += 64) {
        int s = x < a.w ? score[x] : 0;
        unsigned long long m = __ballot(s != 0);
        if (s) {
            int idx = off + __popcll(m & ((1ull << lane) - 1ull));
            if (idx < a.cap) {
                xy[idx] = make_float2((float)x, (float)y);
                resp[idx] = a.nms ? (float)s : 0.f;
            }
        }
        off += __popcll(m);
    }
}

// svo_config.fast_keep_strongest (BASELINE config #4: "exactly N features per frame"): keeps the `keep` highest-response
// corners of every image -- ties by raster order, i.e. np.argsort(-response, kind="stable")[:keep] -- IN PLACE and in
// raster order.  Responses are the integer FAST scores (1..255): a 256-bin histogram gives the cut-off score t, every
// corner above t stays and so do the first (keep - #above) corners AT t.  One workgroup per image walks the list in
// chunks of 1024 (a chunk is read before anything is written, and the write positions never overtake the reads).
__global__ __launch_bounds__(1024) void fast_keep_strongest_kernel(FastArgs a, int keep)
{
    __shared__ int hist[256];
    __shared__ int wave_eq[16], wave_kp[16];
    __shared__ int s_t, s_need_eq, s_eq_base, s_out_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n_out[b];
    if (n <= keep || n > a.cap) return;          // nothing to drop / over capacity: the pair fails as SVO_FAIL_CAPACITY anyway
    float2 *xy = a.kp_xy + (int64_t)b * a.kp_stride;
    float *resp = a.kp_resp + (int64_t)b * a.kp_stride;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[min(255, max(0, (int)resp[i]))], 1);
    __syncthreads();
    if (tid == 0) {
        int above = 0, t = 255;
        for (; t > 0; t--) {                     // largest t with #(response >= t) >= keep
            if (above + hist[t] >= keep) break;
            above += hist[t];
        }
        s_t = t; s_need_eq = keep - above; s_eq_base = 0; s_out_base = 0;
    }
    __syncthreads();
    const int t = s_t, need_eq = s_need_eq;
    for (int start = 0; start < n; start += 1024) {
        const int i = start + tid;
        const bool in = i < n;
        const float2 p = in ? xy[i] : make_float2(0.f, 0.f);
        const float r = in ? resp[i] : 0.f;
        const int ri = (int)r;
        const bool eq = in && ri == t;
        const unsigned long long me = __ballot(eq);
        if (lane == 0) wave_eq[wv] = __popcll(me);
        __syncthreads();
        int eq_pre = s_eq_base, eq_tot = 0;
        for (int q = 0; q < 16; q++) { const int c = wave_eq[q]; if (q < wv) eq_pre += c; eq_tot += c; }
        const int eq_rank = eq_pre + __popcll(me & ((1ull << lane) - 1ull));
        const bool kp = in && (ri > t || (eq && eq_rank < need_eq));
        const unsigned long long mk = __ballot(kp);
        if (lane == 0) wave_kp[wv] = __popcll(mk);
        __syncthreads();
        int kp_pre = s_out_base, kp_tot = 0;
        for (int q = 0; q < 16; q++) { const int c = wave_kp[q]; if (q < wv) kp_pre += c; kp_tot += c; }
        if (kp) {
            const int dst = kp_pre + __popcll(mk & ((1ull << lane) - 1ull));
            xy[dst] = p; resp[dst] = r;
        }
        __syncthreads();
        if (tid == 0) { s_eq_base += eq_tot; s_out_base += kp_tot; }
        __syncthreads();
    }
    if (tid == 0) a.n_out[b] = keep;
}

void launch_fast_keep_strongest(const FastArgs &a, int batch, int keep, hipStream_t st)
{
    if (keep <= 0 || batch <= 0 || !a.nms) return;
    hipLaunchKernelGGL(fast_keep_strongest_kernel, dim3(batch), dim3(1024), 0, st, a, keep);
}

void launch_fast(const FastArgs &a, int batch, hipStream_t st)
{
    (void)hipMemsetAsync(a.rowcount, 0, sizeof(int) * (size_t)a.rowcount_stride * batch, st);
    dim3 g1((a.w + kTileW - 1) / kTileW, (a.h + kTileH - 1) / kTileH, batch), b1(64, 4, 1);
    hipLaunchKernelGGL(fast_score_kernel, g1, b1, 0, st, a);
    dim3 g2((a.h + 3) / 4, batch, 1), b2(256, 1, 1);
    hipLaunchKernelGGL(fast_emit_kernel, g2, b2, 0, st, a);
}

}  // namespace svo

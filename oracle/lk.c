/*
 * oracle/lk.c -- CPU restatement of cv::buildOpticalFlowPyramid + cv::calcOpticalFlowPyrLK
 * (OpenCV 3.4 modules/video/src/lkpyramid.cpp, modules/imgproc/src/pyramids.cpp) exactly as the
 * reference calls it four times per frame at src/tracking.cpp:593-618:
 *   winSize 21x21, maxLevel 3, TermCriteria(COUNT+EPS, 30, 0.01), flags 0, minEigThreshold 1e-3.
 * Follows SURVEY.md Appendix A.2-A.4.  TEST INFRASTRUCTURE ONLY.
 *
 * CANONICAL: upstream accumulates A11,A12,A22,b1,b2 in float in a SIMD-lane
 * dependent order.  Here the integer products are summed EXACTLY in int64 and converted to
 * float once before the 2^-20 scale (upstream's own acctype=int64 variant) -- order independent,
 * so a 64-lane GPU reduction is bit-identical.  Everything else is upstream's single-precision
 * recipe with FP contraction off.
 *
 * ACCUMULATION SWITCH (orc_lk_set_accum, tests only).  Mode 0 is the parity target of lk_kernel (svo_config.lk_accum =
 * exact, the default); mode 2 is the parity target of lk_sse2_kernel (lk_accum = sse2, round 4); mode 1 exists for the
 * sensitivity study only.  Modes 1 and 2 are the same function with upstream's x86 accumulation instead -- acctype = itemtype = float (every OpenCV 3 build that is not
 * the Tegra one), either in plain raster order (the scalar loop of lkpyramid.cpp) or in the lane order
 * of its CV_SSE2 block as recalled from OpenCV 3.4 (A: four lanes over x = 0..19 plus a scalar tail
 * for x = 20; b: two 4-lane accumulators fed by _mm_madd_epi16 pairs (d_k, d_k+4) over x = 0..15 plus
 * a scalar tail for x = 16..20).  tests/test_lk_accum_sensitivity.py measures how far tracks and
 * poses move between the three orders; tests/test_gpu_parity_lk_sse2.py holds the HIP kernel to mode 2 bit for bit.
 */
#include "svo_oracle.h"
#include <math.h>
#include <stddef.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* BORDER_REFLECT_101: ...2 1 | 0 1 2 ... n-1 | n-2 n-3 ... */
static inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * n - 2 - i;
    }
    return i;
}

/* pyrDown_<FixPtCast<uchar,8>>: separable [1 4 6 4 1], unnormalised int rows, (v+128)>>8. */
void orc_pyr_down(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2, x, y, k;
    int *rows = (int *)malloc(sizeof(int) * (size_t)dw * 5);
    for (y = 0; y < dh; y++) {
        for (k = 0; k < 5; k++) {
            const uint8_t *s = src + (size_t)refl101(2 * y + k - 2, h) * spitch;
            int *r = rows + (size_t)k * dw;
            for (x = 0; x < dw; x++) {
                int c = 2 * x;
                r[x] = s[refl101(c, w)] * 6 + (s[refl101(c - 1, w)] + s[refl101(c + 1, w)]) * 4 +
                       s[refl101(c - 2, w)] + s[refl101(c + 2, w)];
            }
        }
        uint8_t *d = dst + (size_t)y * dpitch;
        for (x = 0; x < dw; x++) {
            int v = rows[2 * dw + x] * 6 + (rows[dw + x] + rows[3 * dw + x]) * 4 + rows[x] +
                    rows[4 * dw + x];
            d[x] = (uint8_t)((v + 128) >> 8);
        }
    }
    free(rows);
}

/* copyMakeBorder(level, padded, win, win, win, win, BORDER_REFLECT_101 | BORDER_ISOLATED) */
static void fill_border(uint8_t *buf, int w, int h, int pad, int pitch)
{
    int x, y;
    for (y = -pad; y < h + pad; y++) {
        uint8_t *d = buf + (size_t)(y + pad) * pitch + pad;
        const uint8_t *s = buf + (size_t)(refl101(y, h) + pad) * pitch + pad;
        if (y < 0 || y >= h)
            for (x = 0; x < w; x++) d[x] = s[x];
        for (x = 1; x <= pad; x++) {
            d[-x] = s[refl101(-x, w)];
            d[w - 1 + x] = s[refl101(w - 1 + x, w)];
        }
    }
}

int orc_pyramid_build(const uint8_t *img, int w, int h, int pitch, int win, int max_level,
                      orc_pyramid *pyr)
{
    int l, y;
    memset(pyr, 0, sizeof(*pyr));
    if (max_level >= ORC_LK_MAX_LEVELS) max_level = ORC_LK_MAX_LEVELS - 1;
    pyr->pad = win;
    pyr->w[0] = w; pyr->h[0] = h; pyr->pitch[0] = w + 2 * win;
    pyr->data[0] = (uint8_t *)malloc((size_t)pyr->pitch[0] * (h + 2 * win));
    for (y = 0; y < h; y++)
        memcpy(pyr->data[0] + (size_t)(y + win) * pyr->pitch[0] + win, img + (size_t)y * pitch, w);
    fill_border(pyr->data[0], w, h, win, pyr->pitch[0]);
    pyr->nlevels = 1;
    for (l = 1; l <= max_level; l++) {
        int pw = pyr->w[l - 1], ph = pyr->h[l - 1];
        int nw = (pw + 1) / 2, nh = (ph + 1) / 2;
        /* buildOpticalFlowPyramid stops when the next level would not exceed the window:
         * "if (sz.width <= winSize.width || sz.height <= winSize.height) return level - 1;" */
        if (nw <= win || nh <= win) break;
        pyr->w[l] = nw; pyr->h[l] = nh; pyr->pitch[l] = nw + 2 * win;
        pyr->data[l] = (uint8_t *)malloc((size_t)pyr->pitch[l] * (nh + 2 * win));
        /* pyrDown reads only the level proper (borderInterpolate on its own size) */
        orc_pyr_down(pyr->data[l - 1] + (size_t)win * pyr->pitch[l - 1] + win, pw, ph,
                     pyr->pitch[l - 1], pyr->data[l] + (size_t)win * pyr->pitch[l] + win,
                     pyr->pitch[l]);
        fill_border(pyr->data[l], nw, nh, win, pyr->pitch[l]);
        pyr->nlevels = l + 1;
    }
    return pyr->nlevels;
}

void orc_pyramid_free(orc_pyramid *pyr)
{
    int l;
    for (l = 0; l < ORC_LK_MAX_LEVELS; l++) { free(pyr->data[l]); pyr->data[l] = NULL; }
    pyr->nlevels = 0;
}

/* calcSharrDeriv(level) into a (w+2pad)x(h+2pad) int16 x2 image whose border is
 * BORDER_CONSTANT 0 (copyMakeBorder(derivI, _derivI, ..., BORDER_CONSTANT|BORDER_ISOLATED)).
 * The level's own reflect-101 border supplies the +-1 neighbours at the image edge, which is
 * what calcSharrDeriv's explicit row/column mirroring computes. */
static int16_t *scharr_deriv(const uint8_t *lvl, int w, int h, int pad, int pitch)
{
    int dp = (w + 2 * pad) * 2, x, y;
    int16_t *d = (int16_t *)calloc((size_t)dp * (h + 2 * pad), sizeof(int16_t));
    for (y = 0; y < h; y++) {
        const uint8_t *r0 = lvl + (size_t)(y - 1 + pad) * pitch + pad;
        const uint8_t *r1 = r0 + pitch, *r2 = r1 + pitch;
        int16_t *o = d + (size_t)(y + pad) * dp + pad * 2;
        for (x = 0; x < w; x++) {
            int t0m = (r0[x - 1] + r2[x - 1]) * 3 + r1[x - 1] * 10;
            int t0p = (r0[x + 1] + r2[x + 1]) * 3 + r1[x + 1] * 10;
            int t1m = r2[x - 1] - r0[x - 1], t1c = r2[x] - r0[x], t1p = r2[x + 1] - r0[x + 1];
            o[2 * x] = (int16_t)(t0p - t0m);
            o[2 * x + 1] = (int16_t)((t1p + t1m) * 3 + t1c * 10);
        }
    }
    return d;
}

/* Iteration log (tools only: the wave-grouping cost models of DESIGN.md section 6): when set, the number of
 * iterations every (point, level) ran is written to log[point * ORC_LK_MAX_LEVELS + level] (-1: level skipped). */
static int32_t *g_iter_log = NULL;
static _Thread_local int g_iter_pt = 0;
void orc_lk_set_iter_log(int32_t *log) { g_iter_log = log; }
/* Exactness log (tools only: tools/model/lk_guard_model.py prices lk_sse2_kernel's integer fast path with it).  For every
 * (point, level) eight 32-bit masks over the iterations j (bit j), at log[(point * ORC_LK_MAX_LEVELS + level) * 8 + k]:
 *   k = 0  every one of the ten b chains of the madd-pair orders (modes 2, 4: eight lane chains of 42 pair sums, two tails of
 *          105 products) has max(sum of its positive terms, sum of |negative terms|) < 2^24 -- then every partial sum in ANY
 *          order is an integer below 2^24 and every float add of the chain is exact
 *   k = 1  the same for the legacy order's chains (mode 3: 84 single products a lane chain)
 *   k = 2  the float b of the current accumulation mode equals the exact integer b bit for bit
 *   k = 3  the iteration ran
 *   k = 4  as k = 0 with the lane chains merged in the pairs the final adds join first: (k0, k2), (k1, k3), tail -- six sums
 *   k = 5  ... all four lane chains of a coordinate merged: lanes, tail -- four sums
 *   k = 6  ... the whole window merged: two sums (then b is lk_kernel's exact b)
 *   k = 7  as k = 0 with the positive sums taken from the terms' high halves only (t >> 16, one unit of slack a term: what a
 *          packed 16-bit reduction on the GPU can afford) */
static uint32_t *g_guard_log = NULL;
/* ... and the claim itself, checked where it is cheap: iterations in which the guard of the current float mode held for all ten
 * chains but a chain's FLOAT total (before the final six adds) differed from its integer total.  Must stay 0. */
static long g_guard_violations = 0, g_guard_checked = 0;
void orc_lk_set_guard_log(uint32_t *log) { g_guard_log = log; if (log) { g_guard_violations = 0; g_guard_checked = 0; } }
long orc_lk_guard_violations(long *checked) { if (checked) *checked = g_guard_checked; return g_guard_violations; }

/* 0 exact int64 (CANONICAL); float accumulation: 1 raster order (the scalar loop), 2 the round-4 restatement (A: four
 * lanes over x = 0..19, b: madd pairs over x = 0..15 -- the parity target of lk_sse2_kernel), and the two upstream SIMD
 * blocks as recalled WHOLE (C11): 3 the legacy CV_SSE2 block (2.4 .. early 3.4: A as in mode 2;
 * b from _mm_mullo/_mm_mulhi_epi16 products, every pixel its own float add -- qb0 lanes [bx0 by0 bx1 by1] then
 * [bx4 by4 bx5 by5], qb1 [bx2 by2 bx3 by3] then [bx6 by6 bx7 by7]), 4 the universal-intrinsic CV_SIMD128 block (later
 * 3.4 / 4.x: A over x = 0..15 in groups of eight with v_muladd -- exact products, so fused or not is the same float --,
 * tail x = 16..20; b = v_dotprod pairs (k, k + 4) as in mode 2).  Mode 2 mixes the A loop of one with the b loop of the
 * other; it stays because the HIP kernel and its committed evidence are held to it. */
static int g_lk_accum = 0;
void orc_lk_set_accum(int mode) { g_lk_accum = (mode >= 1 && mode <= 4) ? mode : 0; }
int orc_lk_get_accum(void) { return g_lk_accum; }

#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))
#define W_BITS 14

static inline int cv_round_f(float v) { return (int)lrintf(v); }   /* round-half-even */
static inline int cv_floor_f(float v) { return (int)floorf(v); }

/* One pyramid level for one point (LKTrackerInvoker::operator() body).  I, J point at pixel
 * (0,0) of the padded levels; dI at deriv (0,0). */
static void lk_point_level(const uint8_t *I, int pitchI, const int16_t *dI, int dpitch,
                           const uint8_t *J, int pitchJ, int w, int h, int level, int max_level,
                           orc_pt2f prev_in, orc_pt2f *next_io, uint8_t *status, int win,
                           int max_iter, double eps2, float min_eig, int16_t *Ibuf,
                           int16_t *dIbuf)
{
    const float half = (float)(win - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    float lscale = (float)(1. / (1 << level));
    float prevx = prev_in.x * lscale, prevy = prev_in.y * lscale;
    float nextx, nexty;
    if (level == max_level) { nextx = prevx; nexty = prevy; }
    else { nextx = next_io->x * 2.f; nexty = next_io->y * 2.f; }
    next_io->x = nextx; next_io->y = nexty;

    prevx -= half; prevy -= half;
    int ipx = cv_floor_f(prevx), ipy = cv_floor_f(prevy);
    if (ipx < -win || ipx >= w || ipy < -win || ipy >= h) {
        if (level == 0) *status = 0;
        return;
    }
    float a = prevx - ipx, b = prevy - ipy;
    int iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
    int iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
    int iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
    int iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;

    const int accum = g_lk_accum;
    int64_t iA11 = 0, iA12 = 0, iA22 = 0;
    float fA11 = 0.f, fA12 = 0.f, fA22 = 0.f;              /* upstream's float accumulators (sensitivity modes) */
    float qA11[4] = {0.f, 0.f, 0.f, 0.f}, qA12[4] = {0.f, 0.f, 0.f, 0.f}, qA22[4] = {0.f, 0.f, 0.f, 0.f};
    const int simdA = (accum == 2 || accum == 3) ? (win / 4) * 4 : (accum == 4 ? (win / 8) * 8 : 0);   /* x < simdA rides the four SSE lanes */
    int x, y;
    for (y = 0; y < win; y++) {
        const uint8_t *src = I + (ptrdiff_t)(y + ipy) * pitchI + ipx;
        const int16_t *ds = dI + (ptrdiff_t)(y + ipy) * dpitch + ipx * 2;
        for (x = 0; x < win; x++, ds += 2) {
            int ival = DESCALE(src[x] * iw00 + src[x + 1] * iw01 + src[x + pitchI] * iw10 +
                               src[x + pitchI + 1] * iw11, W_BITS - 5);
            int ixval = DESCALE(ds[0] * iw00 + ds[2] * iw01 + ds[dpitch] * iw10 +
                                ds[dpitch + 2] * iw11, W_BITS);
            int iyval = DESCALE(ds[1] * iw00 + ds[3] * iw01 + ds[dpitch + 1] * iw10 +
                                ds[dpitch + 3] * iw11, W_BITS);
            Ibuf[y * win + x] = (int16_t)ival;
            dIbuf[(y * win + x) * 2] = (int16_t)ixval;
            dIbuf[(y * win + x) * 2 + 1] = (int16_t)iyval;
            iA11 += (int64_t)ixval * ixval;
            iA12 += (int64_t)ixval * iyval;
            iA22 += (int64_t)iyval * iyval;
            if (accum) {
                if (x < simdA) {                           /* _mm_cvtepi32_ps, _mm_mul_ps, _mm_add_ps per lane */
                    float fx = (float)ixval, fy = (float)iyval;
                    qA22[x & 3] += fy * fy;
                    qA12[x & 3] += fx * fy;
                    qA11[x & 3] += fx * fx;
                } else {                                   /* iA11 += (itemtype)(ixval*ixval) */
                    fA11 += (float)(ixval * ixval);
                    fA12 += (float)(ixval * iyval);
                    fA22 += (float)(iyval * iyval);
                }
            }
        }
    }
    float A11, A12, A22;
    if (accum) {
        if (accum >= 2) {                                  /* iA11 += A11buf[0] + A11buf[1] + A11buf[2] + A11buf[3] */
            fA11 += qA11[0] + qA11[1] + qA11[2] + qA11[3];
            fA12 += qA12[0] + qA12[1] + qA12[2] + qA12[3];
            fA22 += qA22[0] + qA22[1] + qA22[2] + qA22[3];
        }
        A11 = fA11 * FLT_SCALE; A12 = fA12 * FLT_SCALE; A22 = fA22 * FLT_SCALE;
    } else {
        A11 = (float)iA11 * FLT_SCALE; A12 = (float)iA12 * FLT_SCALE; A22 = (float)iA22 * FLT_SCALE;
    }
    float D = A11 * A22 - A12 * A12;
    float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) /
                   (float)(2 * win * win);
    if (minEig < min_eig || D < FLT_EPSILON) {
        if (level == 0) *status = 0;
        return;
    }
    D = 1.f / D;

    nextx -= half; nexty -= half;
    float pdx = 0.f, pdy = 0.f;
    int j;
    for (j = 0; j < max_iter; j++) {
        int inx = cv_floor_f(nextx), iny = cv_floor_f(nexty);
        if (inx < -win || inx >= w || iny < -win || iny >= h) {
            if (level == 0) *status = 0;
            break;
        }
        a = nextx - inx; b = nexty - iny;
        iw00 = cv_round_f((1.f - a) * (1.f - b) * (1 << W_BITS));
        iw01 = cv_round_f(a * (1.f - b) * (1 << W_BITS));
        iw10 = cv_round_f((1.f - a) * b * (1 << W_BITS));
        iw11 = (1 << W_BITS) - iw00 - iw01 - iw10;
        int64_t ib1 = 0, ib2 = 0;
        float fb1 = 0.f, fb2 = 0.f;
        float qb0[4] = {0.f, 0.f, 0.f, 0.f}, qb1[4] = {0.f, 0.f, 0.f, 0.f};
        const int simdB = accum >= 2 ? (win / 8) * 8 : 0;
        /* exactness log: [chain][0 positive | 1 negative] sums; chains 0..7 = 2 k + xy (pixels k, k + 4 of a group), 8 / 9 the tails */
        int64_t gpair[10][2], gprod[10][2], gq[10][2], gtl[21][2][2];
        memset(gpair, 0, sizeof gpair); memset(gprod, 0, sizeof gprod); memset(gq, 0, sizeof gq); memset(gtl, 0, sizeof gtl);
        for (y = 0; y < win; y++) {
            const uint8_t *Jp = J + (ptrdiff_t)(y + iny) * pitchJ + inx;
            const int16_t *Ip = Ibuf + y * win, *dIp = dIbuf + y * win * 2;
            int dgrp[8];
            for (x = 0; x < win; x++) {
                int diff = DESCALE(Jp[x] * iw00 + Jp[x + 1] * iw01 + Jp[x + pitchJ] * iw10 +
                                   Jp[x + pitchJ + 1] * iw11, W_BITS - 5) - Ip[x];
                ib1 += (int64_t)(diff * dIp[2 * x]);
                ib2 += (int64_t)(diff * dIp[2 * x + 1]);
                if (g_guard_log) {
                    const int gb = (win / 8) * 8, xy_n = 2;
                    int xy;
                    for (xy = 0; xy < xy_n; xy++) {
                        const int64_t p = (int64_t)diff * dIp[2 * x + xy];
                        const int c = x >= gb ? 8 + xy : 2 * (x & 3) + xy;
                        gprod[c][p < 0] += p < 0 ? -p : p;
                        if (x >= gb) { gpair[c][p < 0] += p < 0 ? -p : p; gtl[y][xy][p < 0] += p < 0 ? -p : p; }
                        else if ((x & 7) >= 4) {         /* the pair (k, k + 4) is complete */
                            const int64_t q = (int64_t)(DESCALE(Jp[x - 4] * iw00 + Jp[x - 3] * iw01 + Jp[x - 4 + pitchJ] * iw10 +
                                                                Jp[x - 3 + pitchJ] * iw11, W_BITS - 5) - Ip[x - 4]) * dIp[2 * (x - 4) + xy] + p;
                            gpair[c][q < 0] += q < 0 ? -q : q;
                            gq[c][q < 0] += ((q < 0 ? -q : q) >> 16) + 1;
                        }
                    }
                }
                if (accum && x >= simdB) {                 /* ib1 += (itemtype)(diff*dIptr[0]) */
                    fb1 += (float)(diff * dIp[2 * x]);
                    fb2 += (float)(diff * dIp[2 * x + 1]);
                } else if (accum == 3) {
                    /* legacy CV_SSE2: mullo / mulhi products of (It_k It_k) x (Ix_k Iy_k), converted and added pixel by
                     * pixel: pixels 0, 1 (then 4, 5) of a group of eight to qb0, 2, 3 (then 6, 7) to qb1 */
                    const int k = x & 7;
                    float *q = (k & 2) ? qb1 : qb0;
                    q[(k & 1) * 2] += (float)(diff * dIp[2 * x]);
                    q[(k & 1) * 2 + 1] += (float)(diff * dIp[2 * x + 1]);
                } else if (accum) {
                    dgrp[x & 7] = diff;
                    if ((x & 7) == 7) {
                        /* _mm_madd_epi16 pairs pixel k with k+4 of the group of eight, exactly in int32;
                         * qb0 = [bx(0,4) by(0,4) bx(1,5) by(1,5)], qb1 = [bx(2,6) by(2,6) bx(3,7) by(3,7)] */
                        const int16_t *g = dIp + 2 * (x - 7);
                        int k;
                        for (k = 0; k < 4; k++) {
                            int sx = dgrp[k] * g[2 * k] + dgrp[k + 4] * g[2 * (k + 4)];
                            int sy = dgrp[k] * g[2 * k + 1] + dgrp[k + 4] * g[2 * (k + 4) + 1];
                            float *q = (k < 2) ? qb0 : qb1;
                            q[(k & 1) * 2] += (float)sx;
                            q[(k & 1) * 2 + 1] += (float)sy;
                        }
                    }
                }
            }
        }
        if (g_guard_log && accum >= 2) {
            /* chain c = 2 k + xy: k = 0, 1 ride qb0[2 k + xy], k = 2, 3 qb1[2 (k - 2) + xy]; the tails are fb1 / fb2 before the final adds */
            int c, held = 1, bad = 0;
            int64_t (*g)[2] = accum == 3 ? gprod : gpair;
            for (c = 0; c < 10; c++) if (g[c][0] >= (1 << 24) || g[c][1] >= (1 << 24)) held = 0;
            if (held) {
                for (c = 0; c < 8; c++) {
                    const int k = c >> 1, xy = c & 1;
                    const float f = k < 2 ? qb0[2 * k + xy] : qb1[2 * (k - 2) + xy];
                    if (f != (float)(g[c][0] - g[c][1])) bad = 1;
                }
                if (fb1 != (float)(g[8][0] - g[8][1]) || fb2 != (float)(g[9][0] - g[9][1])) bad = 1;
#ifdef _OPENMP
#pragma omp atomic
#endif
                g_guard_checked++;
                if (bad) {
#ifdef _OPENMP
#pragma omp atomic
#endif
                    g_guard_violations++;
                }
            }
        }
        float b1, b2;
        if (accum) {
            if (accum >= 2) {                              /* bbuf = qb0 + qb1; ib1 += bbuf[0] + bbuf[2]; ib2 += bbuf[1] + bbuf[3] */
                float bb0 = qb0[0] + qb1[0], bb1 = qb0[1] + qb1[1], bb2 = qb0[2] + qb1[2], bb3 = qb0[3] + qb1[3];
                fb1 += bb0 + bb2;
                fb2 += bb1 + bb3;
            }
            b1 = fb1 * FLT_SCALE; b2 = fb2 * FLT_SCALE;
        } else {
            b1 = (float)ib1 * FLT_SCALE; b2 = (float)ib2 * FLT_SCALE;
        }
        if (g_guard_log) {
            uint32_t *gl = g_guard_log + ((size_t)g_iter_pt * ORC_LK_MAX_LEVELS + level) * 8;
            int c, okp = 1, okl = 1, ok6 = 1, ok4 = 1, ok2 = 1, okq = 1, xy, sg;
            const int64_t lim = 1 << 24;
            for (xy = 0; xy < 2; xy++)
                for (sg = 0; sg < 2; sg++) {
                    const int64_t k0 = gpair[0 + xy][sg], k1 = gpair[2 + xy][sg], k2 = gpair[4 + xy][sg], k3 = gpair[6 + xy][sg], tl = gpair[8 + xy][sg];
                    if (k0 + k2 >= lim || k1 + k3 >= lim || tl >= lim) ok6 = 0;
                    if (k0 + k1 + k2 + k3 >= lim || tl >= lim) ok4 = 0;
                    if (k0 + k1 + k2 + k3 + tl >= lim) ok2 = 0;
                    /* packed version: a lane chain's terms by their high halves; a tail lane (one row, five products) sums its
                     * products exactly first */
                    for (c = 0; c < 4; c++) if (gq[2 * c + xy][sg] >= 256) okq = 0;
                    { int64_t q = 0; int r; for (r = 0; r < win; r++) q += (gtl[r][xy][sg] >> 16) + 1; if (q >= 256) okq = 0; }
                }
            for (c = 0; c < 10; c++) {
                if (gpair[c][0] >= (1 << 24) || gpair[c][1] >= (1 << 24)) okp = 0;
                if (gprod[c][0] >= (1 << 24) || gprod[c][1] >= (1 << 24)) okl = 0;
            }
            if (j < 32) {
                if (okp) gl[0] |= 1u << j;
                if (okl) gl[1] |= 1u << j;
                if (b1 == (float)ib1 * FLT_SCALE && b2 == (float)ib2 * FLT_SCALE) gl[2] |= 1u << j;
                gl[3] |= 1u << j;
                if (ok6) gl[4] |= 1u << j;
                if (ok4) gl[5] |= 1u << j;
                if (ok2) gl[6] |= 1u << j;
                if (okq) gl[7] |= 1u << j;
            }
        }
        float dlx = (A12 * b2 - A22 * b1) * D;
        float dly = (A12 * b1 - A11 * b2) * D;
        nextx += dlx; nexty += dly;
        next_io->x = nextx + half; next_io->y = nexty + half;
        if ((double)dlx * dlx + (double)dly * dly <= eps2) break;
        if (j > 0 && fabs((double)(dlx + pdx)) < 0.01 && fabs((double)(dly + pdy)) < 0.01) {
            next_io->x -= dlx * 0.5f; next_io->y -= dly * 0.5f;
            break;
        }
        pdx = dlx; pdy = dly;
    }
    if (g_iter_log) g_iter_log[(size_t)g_iter_pt * ORC_LK_MAX_LEVELS + level] = j < max_iter ? j + 1 : max_iter;

    /* err is requested by the reference (error1..4) and flags has no GET_MIN_EIGENVALS: the
     * level-0 post-pass can still clear status when the final window is out of bounds
     * (Appendix A.4 step 7).  The err value itself is never read by the reference. */
    if (*status && level == 0) {
        float fx = next_io->x - half, fy = next_io->y - half;
        int inx = cv_floor_f(fx), iny = cv_floor_f(fy);
        if (inx < -win || inx >= w || iny < -win || iny >= h) *status = 0;
    }
}

int orc_lk_track(const orc_pyramid *prev, const orc_pyramid *next, const orc_pt2f *prev_pts,
                 int n, orc_pt2f *next_pts, uint8_t *status, int win, int max_iter, double eps,
                 float min_eig, int threads)
{
    int nl = prev->nlevels < next->nlevels ? prev->nlevels : next->nlevels;
    int max_level = nl - 1, l;
    int pad = prev->pad;
    if (win != prev->pad || win != next->pad) return -1;
    /* criteria clamps: maxCount in [0,100], epsilon in [0,10], then squared */
    if (max_iter < 0) max_iter = 0;
    if (max_iter > 100) max_iter = 100;
    if (eps < 0.) eps = 0.;
    if (eps > 10.) eps = 10.;
    double eps2 = eps * eps;
    int16_t *deriv[ORC_LK_MAX_LEVELS];
    for (l = 0; l <= max_level; l++)
        deriv[l] = scharr_deriv(prev->data[l], prev->w[l], prev->h[l], pad, prev->pitch[l]);
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
#endif
    {
        int16_t *Ibuf = (int16_t *)malloc(sizeof(int16_t) * (size_t)win * win * 3);
        int16_t *dIbuf = Ibuf + win * win;
        int i, lv;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
        for (i = 0; i < n; i++) {
            status[i] = 1;
            next_pts[i].x = 0.f; next_pts[i].y = 0.f;
            g_iter_pt = i;
            if (g_iter_log) for (lv = 0; lv < ORC_LK_MAX_LEVELS; lv++) g_iter_log[(size_t)i * ORC_LK_MAX_LEVELS + lv] = -1;
            if (g_guard_log) memset(g_guard_log + (size_t)i * ORC_LK_MAX_LEVELS * 8, 0, sizeof(uint32_t) * ORC_LK_MAX_LEVELS * 8);
            for (lv = max_level; lv >= 0; lv--) {
                int dp = (prev->w[lv] + 2 * pad) * 2;
                lk_point_level(prev->data[lv] + (size_t)pad * prev->pitch[lv] + pad,
                               prev->pitch[lv], deriv[lv] + (size_t)pad * dp + pad * 2, dp,
                               next->data[lv] + (size_t)pad * next->pitch[lv] + pad,
                               next->pitch[lv], prev->w[lv], prev->h[lv], lv, max_level,
                               prev_pts[i], &next_pts[i], &status[i], win, max_iter, eps2,
                               min_eig, Ibuf, dIbuf);
            }
        }
        free(Ibuf);
    }
    for (l = 0; l <= max_level; l++) free(deriv[l]);
    return 0;
}

/* Tracking::deleteBadmatchFeatures (reference src/tracking.cpp:623-660): a stable filter.
 * Call-site mapping (:619-620): p0=t1_left, p1=t1_right, p2=t2_right, p3=t2_left, p0r=LK#4 out;
 * s0..s3 = status1..4. */
int orc_circular_keep(const orc_pt2f *p0, const orc_pt2f *p1, const orc_pt2f *p2,
                      const orc_pt2f *p3, const orc_pt2f *p0r, const uint8_t *s0,
                      const uint8_t *s1, const uint8_t *s2, const uint8_t *s3, int n,
                      double match_err, uint8_t *keep)
{
    int i, m = 0;
    for (i = 0; i < n; i++) {
        int outside = (p3[i].x < 0) || (p3[i].y < 0) || (p2[i].x < 0) || (p2[i].y < 0) ||
                      (p1[i].x < 0) || (p1[i].y < 0) || (p0[i].x < 0) || (p0[i].y < 0) ||
                      (p0r[i].x < 0) || (p0r[i].y < 0);
        int bad = (s2[i] == 0) || (s1[i] == 0) || (s0[i] == 0) || (s3[i] == 0);
        /* std::abs(float) compared with the double feature_match_error_ */
        int noepi = ((double)fabsf(p0[i].y - p1[i].y) > match_err) ||
                    ((double)fabsf(p2[i].y - p3[i].y) > match_err);
        keep[i] = !(outside || bad || noepi);
        m += keep[i];
    }
    return m;
}

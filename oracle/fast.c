/*
 * oracle/fast.c -- CPU restatement of cv::FAST(img, kps, threshold, nonmaxSuppression=true),
 * FastFeatureDetector::TYPE_9_16, as called at reference src/tracking.cpp:101 (thr 20) and
 * src/ORBextractor.cpp:763,768 (per-cell, thr 20 / 7).  Follows SURVEY.md Appendix A.1
 * (OpenCV 3.4 modules/features2d/src/fast.cpp + fast_score.cpp).  TEST INFRASTRUCTURE ONLY.
 *
 * All integer: bit-exact target for the HIP kernel.
 */
#include "svo_oracle.h"
#include <stdlib.h>
#include <string.h>

/* Bresenham circle of radius 3, k = 0..15 (dx, dy). */
static const int CIRC[16][2] = {
    {0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
    {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Segment test: >= 9 contiguous circle pixels all brighter than v+thr or all darker than
 * v-thr (fast.cpp: "if( ++count > K )" with K = 8 over k = 0..24). */
static int is_corner(const int d[25] /* v - p_k */, int thr)
{
    int count = 0, k;
    for (k = 0; k < 25; k++) {           /* p_k < v - thr  <=> d > thr  (darker arc) */
        if (d[k] > thr) { if (++count > 8) return 1; } else count = 0;
    }
    count = 0;
    for (k = 0; k < 25; k++) {           /* p_k > v + thr  <=> d < -thr (brighter arc) */
        if (d[k] < -thr) { if (++count > 8) return 1; } else count = 0;
    }
    return 0;
}

/* cornerScore<16>: the largest threshold for which the pixel is still a corner. */
static int corner_score(const int d[25], int thr)
{
    int k, a0 = thr;
    for (k = 0; k < 16; k += 2) {
        int a = imin(d[k + 1], d[k + 2]);
        a = imin(a, d[k + 3]);
        if (a <= a0) continue;
        a = imin(a, d[k + 4]); a = imin(a, d[k + 5]); a = imin(a, d[k + 6]);
        a = imin(a, d[k + 7]); a = imin(a, d[k + 8]);
        a0 = imax(a0, imin(a, d[k]));
        a0 = imax(a0, imin(a, d[k + 9]));
    }
    int b0 = -a0;
    for (k = 0; k < 16; k += 2) {
        int b = imax(d[k + 1], d[k + 2]);
        b = imax(b, d[k + 3]); b = imax(b, d[k + 4]); b = imax(b, d[k + 5]);
        if (b >= b0) continue;
        b = imax(b, d[k + 6]); b = imax(b, d[k + 7]); b = imax(b, d[k + 8]);
        b0 = imin(b0, imax(b, d[k]));
        b0 = imin(b0, imax(b, d[k + 9]));
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t *img, int w, int h, int pitch, int thr, int nms,
                 orc_keypoint *out, int cap)
{
    if (w < 7 || h < 7) return 0;
    thr = imin(imax(thr, 0), 255);
    /* score map: 0 = not a corner (scores of real corners are >= thr >= ... > 0 unless thr == 0;
     * upstream stores scores in uchar buffers cleared to 0 the same way). */
    uint8_t *score = (uint8_t *)calloc((size_t)w * h, 1);
    uint8_t *corner = (uint8_t *)calloc((size_t)w * h, 1);
    int x, y, k, n = 0;
    for (y = 3; y < h - 3; y++) {
        const uint8_t *row = img + (size_t)y * pitch;
        for (x = 3; x < w - 3; x++) {
            int v = row[x], d[25];
            for (k = 0; k < 16; k++)
                d[k] = v - img[(size_t)(y + CIRC[k][1]) * pitch + x + CIRC[k][0]];
            for (k = 16; k < 25; k++) d[k] = d[k - 16];
            if (!is_corner(d, thr)) continue;
            corner[(size_t)y * w + x] = 1;
            if (nms) score[(size_t)y * w + x] = (uint8_t)corner_score(d, thr);
        }
    }
    /* output in row-major order; with NMS keep only strict 3x3 maxima of the score map */
    for (y = 3; y < h - 3; y++) {
        for (x = 3; x < w - 3; x++) {
            size_t i = (size_t)y * w + x;
            if (!corner[i]) continue;
            int s = score[i];
            if (nms) {
                if (!(s > score[i - 1] && s > score[i + 1] &&
                      s > score[i - w - 1] && s > score[i - w] && s > score[i - w + 1] &&
                      s > score[i + w - 1] && s > score[i + w] && s > score[i + w + 1]))
                    continue;
            }
            if (n < cap && out) {
                orc_keypoint *kp = &out[n];
                kp->x = (float)x; kp->y = (float)y; kp->size = 7.f; kp->angle = -1.f;
                kp->response = nms ? (float)s : 0.f; kp->octave = 0; kp->class_id = -1;
            }
            n++;
        }
    }
    free(score); free(corner);
    return n;
}

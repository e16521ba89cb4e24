/*
 * oracle/orb.c -- CPU restatement of the reference's ORB path (BASELINE config #3):
 *   ORBextractor (reference src/ORBextractor.cpp, ORB-SLAM2 lineage): scale tables and quotas
 *   (:359-418), ComputePyramid (:1061-1085), ComputeKeyPointsOctTree (:717-807), DistributeOctTree /
 *   DivideNode (:430-485, 487-715), IC_Angle (:21-48), 7x7 sigma-2 blur + rBRIEF (:51-97, 981-1055),
 *   and Tracking::ORB_Robust_Find_MuliImage_MatchedFeatures (src/tracking.cpp:534-581) with its
 *   BruteForce-Hamming matcher.  OpenCV callees (resize INTER_LINEAR, copyMakeBorder, FAST,
 *   GaussianBlur, fastAtan2) follow SURVEY.md Appendix A.7.  TEST INFRASTRUCTURE ONLY.
 *
 * CANONICAL choices: (O1) quadtree ties in "sort(vector<pair<int, ExtractorNode*>>)" are broken
 * by node CREATION ORDER instead of heap address (the reference itself is nondeterministic
 * there); (O2) GaussianBlur uses the legacy 8-bit integer kernel
 * round(g*256) = [18 34 49 55 49 34 18] with (sum + 2^15) >> 16 and saturation (OpenCV <= 3.4.0);
 * (O3) resize uses scale = 1 / (dst/src) evaluated in double as cv::resize does.
 */
#include "svo_oracle.h"
#include "orb_pattern.h"
#include <math.h>
#include <stddef.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define ORB_EDGE 19
#define ORB_HALF_PATCH 15
#define ORB_PATCH 31

static inline int refl101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}
static inline int cv_round_f(float v) { return (int)lrintf(v); }
static inline int cv_round_d(double v) { return (int)lrint(v); }

/* ---- scale tables, per-level quotas, umax (ORBextractor::ORBextractor) ------------------------ */
void orc_orb_setup(int nfeatures, float scale_factor_f, int nlevels, float *scale, float *inv_scale,
                   int *quota, int *umax)
{
    const double scaleFactor = (double)scale_factor_f;       /* member `double scaleFactor` */
    int i, level, v, v0;
    scale[0] = 1.0f;
    for (i = 1; i < nlevels; i++) scale[i] = (float)(scale[i - 1] * scaleFactor);
    for (i = 0; i < nlevels; i++) inv_scale[i] = 1.0f / scale[i];
    float factor = (float)(1.0f / scaleFactor);
    float nDesired = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (level = 0; level < nlevels - 1; level++) {
        quota[level] = cv_round_f(nDesired);
        sum += quota[level];
        nDesired *= factor;
    }
    quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    int vmax = (int)floor(ORB_HALF_PATCH * sqrt(2.f) / 2 + 1);
    int vmin = (int)ceil(ORB_HALF_PATCH * sqrt(2.f) / 2);
    const double hp2 = ORB_HALF_PATCH * ORB_HALF_PATCH;
    for (v = 0; v <= ORB_HALF_PATCH; v++) umax[v] = 0;
    for (v = 0; v <= vmax; ++v) umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = ORB_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0;
        ++v0;
    }
}

/* ---- cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8-bit single channel ----------------- */
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int spitch, uint8_t *dst, int dw, int dh,
                          int dpitch)
{
    const double inv_sx = (double)dw / sw, inv_sy = (double)dh / sh;
    const double scale_x = 1. / inv_sx, scale_y = 1. / inv_sy;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    short *ialpha = (short *)malloc(sizeof(short) * 2 * dw);
    int *rows = (int *)malloc(sizeof(int) * 2 * dw);
    int dx, dy;
    for (dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ialpha[2 * dx] = (short)cv_round_f((1.f - fx) * 2048);
        ialpha[2 * dx + 1] = (short)cv_round_f(fx * 2048);
    }
    for (dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= sy;
        short b0 = (short)cv_round_f((1.f - fy) * 2048), b1 = (short)cv_round_f(fy * 2048);
        int y0 = sy < 0 ? 0 : (sy >= sh ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 >= sh ? sh - 1 : sy + 1);
        const uint8_t *S0 = src + (size_t)y0 * spitch, *S1 = src + (size_t)y1 * spitch;
        for (dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            rows[dx] = S0[sx] * ialpha[2 * dx] + S0[sx1] * ialpha[2 * dx + 1];
            rows[dw + dx] = S1[sx] * ialpha[2 * dx] + S1[sx1] * ialpha[2 * dx + 1];
        }
        uint8_t *D = dst + (size_t)dy * dpitch;
        for (dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (rows[dx] >> 4)) >> 16) + ((b1 * (rows[dw + dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(ialpha); free(rows);
}

/* ---- pyramid ---------------------------------------------------------------------------------- */
typedef struct {
    int nlevels;
    int w[8], h[8], pitch[8];
    uint8_t *buf[8];            /* bordered buffer; pixel (x,y) at buf[(y+19)*pitch + x+19] */
    float scale[8], inv_scale[8];
    int quota[8], umax[16];
} orb_pyr;

static uint8_t *lvl_px(const orb_pyr *p, int l) { return p->buf[l] + (size_t)ORB_EDGE * p->pitch[l] + ORB_EDGE; }

static void orb_fill_border(uint8_t *img0, int w, int h, int pitch)
{
    int x, y;
    for (y = -ORB_EDGE; y < h + ORB_EDGE; y++) {
        uint8_t *d = img0 + (ptrdiff_t)y * pitch;
        const uint8_t *s = img0 + (ptrdiff_t)refl101(y, h) * pitch;
        if (y < 0 || y >= h) for (x = 0; x < w; x++) d[x] = s[x];
        for (x = 1; x <= ORB_EDGE; x++) { d[-x] = s[refl101(-x, w)]; d[w - 1 + x] = s[refl101(w - 1 + x, w)]; }
    }
}

static void orb_pyramid_build(orb_pyr *p, const uint8_t *img, int w, int h, int pitch, int nfeatures,
                              float scaleFactor, int nlevels)
{
    int l, y;
    memset(p, 0, sizeof(*p));
    p->nlevels = nlevels;
    orc_orb_setup(nfeatures, scaleFactor, nlevels, p->scale, p->inv_scale, p->quota, p->umax);
    for (l = 0; l < nlevels; l++) {
        float sc = p->inv_scale[l];
        p->w[l] = cv_round_f((float)w * sc); p->h[l] = cv_round_f((float)h * sc);
        p->pitch[l] = p->w[l] + 2 * ORB_EDGE;
        p->buf[l] = (uint8_t *)calloc((size_t)p->pitch[l] * (p->h[l] + 2 * ORB_EDGE), 1);
        if (l == 0)
            for (y = 0; y < h; y++) memcpy(lvl_px(p, 0) + (size_t)y * p->pitch[0], img + (size_t)y * pitch, w);
        else
            orc_resize_linear_u8(lvl_px(p, l - 1), p->w[l - 1], p->h[l - 1], p->pitch[l - 1], lvl_px(p, l), p->w[l],
                                 p->h[l], p->pitch[l]);
        orb_fill_border(lvl_px(p, l), p->w[l], p->h[l], p->pitch[l]);
    }
}

static void orb_pyramid_free(orb_pyr *p) { int l; for (l = 0; l < 8; l++) free(p->buf[l]); }

/* ---- FAST cornerness: V = (largest t for which the pixel is a FAST-9/16 corner), 0 if < 1 ----- */
static const int CIRC[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};
static int fast_cornerness(const uint8_t *p, int pitch)
{
    int d[25], k, s, best = -255, bestn = -255, v = p[0];
    for (k = 0; k < 16; k++) d[k] = v - p[CIRC[k][1] * pitch + CIRC[k][0]];
    for (k = 16; k < 25; k++) d[k] = d[k - 16];
    for (s = 0; s < 16; s++) {
        int mn = d[s], mx = d[s];
        for (k = 1; k < 9; k++) { if (d[s + k] < mn) mn = d[s + k]; if (d[s + k] > mx) mx = d[s + k]; }
        if (mn > best) best = mn;            /* max over arcs of min(d)  */
        if (-mx > bestn) bestn = -mx;        /* max over arcs of min(-d) */
    }
    v = (best > bestn ? best : bestn) - 1;   /* corner at threshold t  <=>  V >= t; score == V */
    return v > 0 ? v : 0;
}

typedef struct { float x, y, response; } orb_cand;

/* ComputeKeyPointsOctTree's per-cell FAST with threshold fallback; candidates relative to
 * (minBorderX, minBorderY), cells row-major, row-major inside a cell. */
static int orb_cell_fast(const orb_pyr *p, int level, int iniTh, int minTh, orb_cand **out)
{
    const int W = p->w[level], H = p->h[level], pitch = p->pitch[level];
    const uint8_t *img = lvl_px(p, level);
    const int minBX = ORB_EDGE - 3, minBY = minBX, maxBX = W - ORB_EDGE + 3, maxBY = H - ORB_EDGE + 3;
    const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
    const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
    int cap = 4096, n = 0, i, j, x, y;
    orb_cand *c = (orb_cand *)malloc(sizeof(orb_cand) * cap);
    *out = c;
    if (nCols <= 0 || nRows <= 0) return 0;
    const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
    uint8_t *V = (uint8_t *)malloc((size_t)(wCell + 6) * (hCell + 6));
    for (i = 0; i < nRows; i++) {
        const float iniY = (float)(minBY + i * hCell);
        float maxY = iniY + hCell + 6;
        if (iniY >= maxBY - 3) continue;
        if (maxY > maxBY) maxY = (float)maxBY;
        for (j = 0; j < nCols; j++) {
            const float iniX = (float)(minBX + j * wCell);
            float maxX = iniX + wCell + 6;
            if (iniX >= maxBX - 6) continue;
            if (maxX > maxBX) maxX = (float)maxBX;
            const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0, cp = wCell + 6;
            if (cw < 7 || ch < 7) continue;
            memset(V, 0, (size_t)cp * (hCell + 6));
            int any20 = 0, thr, pass;
            for (y = 3; y < ch - 3; y++)
                for (x = 3; x < cw - 3; x++) {
                    int v = fast_cornerness(img + (size_t)(y0 + y) * pitch + x0 + x, pitch);
                    V[y * cp + x] = (uint8_t)v;
                    if (v >= iniTh) any20 = 1;
                }
            thr = iniTh;
            for (pass = 0; pass < 2; pass++) {
                int found = 0;
                if (pass == 0 && !any20) { thr = minTh; continue; }
                for (y = 3; y < ch - 3; y++)
                    for (x = 3; x < cw - 3; x++) {
                        int s = V[y * cp + x];
                        if (s < thr) continue;
#define SC(dx, dy) (V[(y + (dy)) * cp + x + (dx)] >= thr ? V[(y + (dy)) * cp + x + (dx)] : 0)
                        if (!(s > SC(-1, 0) && s > SC(1, 0) && s > SC(-1, -1) && s > SC(0, -1) && s > SC(1, -1) &&
                              s > SC(-1, 1) && s > SC(0, 1) && s > SC(1, 1)))
                            continue;
#undef SC
                        if (n == cap) { cap *= 2; c = (orb_cand *)realloc(c, sizeof(orb_cand) * cap); *out = c; }
                        c[n].x = (float)x + (float)(j * wCell); c[n].y = (float)y + (float)(i * hCell);
                        c[n].response = (float)s;
                        n++; found = 1;
                    }
                /* "if (vKeysCell.empty()) FAST(..., minThFAST)": NMS can empty a cell whose corners all tie */
                if (found || thr == minTh) break;
                thr = minTh;
            }
        }
    }
    free(V);
    return n;
}

/* ---- DistributeOctTree ------------------------------------------------------------------------ */
typedef struct {
    int ULx, ULy, URx, URy, BLx, BLy, BRx, BRy;
    int begin, count;      /* keys: idx[begin .. begin+count) */
    int prev, next;        /* std::list links */
    int noMore;
} qnode;

typedef struct {
    qnode *nodes; int n_nodes, cap_nodes;
    int *idx; int n_idx, cap_idx;
    int head, tail, size;
    const orb_cand *keys;
} qtree;

static int qt_new_node(qtree *t)
{
    if (t->n_nodes == t->cap_nodes) { t->cap_nodes *= 2; t->nodes = (qnode *)realloc(t->nodes, sizeof(qnode) * t->cap_nodes); }
    memset(&t->nodes[t->n_nodes], 0, sizeof(qnode));
    return t->n_nodes++;
}
static int qt_alloc_idx(qtree *t, int n)
{
    while (t->n_idx + n > t->cap_idx) { t->cap_idx *= 2; t->idx = (int *)realloc(t->idx, sizeof(int) * t->cap_idx); }
    int b = t->n_idx; t->n_idx += n; return b;
}
static void qt_push_back(qtree *t, int id)
{
    t->nodes[id].prev = t->tail; t->nodes[id].next = -1;
    if (t->tail >= 0) t->nodes[t->tail].next = id; else t->head = id;
    t->tail = id; t->size++;
}
static void qt_push_front(qtree *t, int id)
{
    t->nodes[id].next = t->head; t->nodes[id].prev = -1;
    if (t->head >= 0) t->nodes[t->head].prev = id; else t->tail = id;
    t->head = id; t->size++;
}
static void qt_erase(qtree *t, int id)
{
    int p = t->nodes[id].prev, n = t->nodes[id].next;
    if (p >= 0) t->nodes[p].next = n; else t->head = n;
    if (n >= 0) t->nodes[n].prev = p; else t->tail = p;
    t->size--;
}
/* ExtractorNode::DivideNode: children ids (or -1 when empty) in n1..n4 order */
static void qt_divide(qtree *t, int id, int ch[4])
{
    qnode P = t->nodes[id];
    const int halfX = (int)ceil((float)(P.URx - P.ULx) / 2), halfY = (int)ceil((float)(P.BRy - P.ULy) / 2);
    qnode c[4];
    int k, q, cnt[4] = {0, 0, 0, 0};
    memset(c, 0, sizeof(c));
    c[0].ULx = P.ULx; c[0].ULy = P.ULy; c[0].URx = P.ULx + halfX; c[0].URy = P.ULy;
    c[0].BLx = P.ULx; c[0].BLy = P.ULy + halfY; c[0].BRx = P.ULx + halfX; c[0].BRy = P.ULy + halfY;
    c[1].ULx = c[0].URx; c[1].ULy = c[0].URy; c[1].URx = P.URx; c[1].URy = P.URy;
    c[1].BLx = c[0].BRx; c[1].BLy = c[0].BRy; c[1].BRx = P.URx; c[1].BRy = P.ULy + halfY;
    c[2].ULx = c[0].BLx; c[2].ULy = c[0].BLy; c[2].URx = c[0].BRx; c[2].URy = c[0].BRy;
    c[2].BLx = P.BLx; c[2].BLy = P.BLy; c[2].BRx = c[0].BRx; c[2].BRy = P.BLy;
    c[3].ULx = c[2].URx; c[3].ULy = c[2].URy; c[3].URx = c[1].BRx; c[3].URy = c[1].BRy;
    c[3].BLx = c[2].BRx; c[3].BLy = c[2].BRy; c[3].BRx = P.BRx; c[3].BRy = P.BRy;
    int *which = (int *)malloc(sizeof(int) * (P.count > 0 ? P.count : 1));
    for (k = 0; k < P.count; k++) {
        const orb_cand *kp = &t->keys[t->idx[P.begin + k]];
        if (kp->x < (float)c[0].URx) q = kp->y < (float)c[0].BRy ? 0 : 2;
        else q = kp->y < (float)c[0].BRy ? 1 : 3;
        which[k] = q; cnt[q]++;
    }
    for (q = 0; q < 4; q++) {
        ch[q] = -1;
        if (cnt[q] == 0) continue;
        int b = qt_alloc_idx(t, cnt[q]), m = 0;
        for (k = 0; k < P.count; k++) if (which[k] == q) t->idx[b + m++] = t->idx[P.begin + k];
        int nid = qt_new_node(t);
        c[q].begin = b; c[q].count = cnt[q]; c[q].noMore = cnt[q] == 1;
        t->nodes[nid] = c[q];
        ch[q] = nid;
    }
    free(which);
}

typedef struct { int size, id; } size_id;
static int cmp_size_id(const void *a, const void *b)
{
    const size_id *x = (const size_id *)a, *y = (const size_id *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);          /* CANONICAL (O1): creation order */
}

/* returns the selected candidate indices in list order */
static int orb_distribute(const orb_cand *keys, int nkeys, int minX, int maxX, int minY, int maxY, int N, int *sel)
{
    qtree t;
    int i, q;
    t.cap_nodes = 64; t.nodes = (qnode *)malloc(sizeof(qnode) * t.cap_nodes); t.n_nodes = 0;
    t.cap_idx = nkeys * 2 + 64; t.idx = (int *)malloc(sizeof(int) * t.cap_idx); t.n_idx = 0;
    t.head = t.tail = -1; t.size = 0; t.keys = keys;
    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));
    const float hX = (float)(maxX - minX) / nIni;
    int *root = (int *)malloc(sizeof(int) * (nIni > 0 ? nIni : 1));
    int *cnt = (int *)calloc(nIni > 0 ? nIni : 1, sizeof(int));
    if (nIni <= 0 || nkeys == 0) { free(root); free(cnt); free(t.nodes); free(t.idx); return 0; }
    for (i = 0; i < nkeys; i++) cnt[(int)(keys[i].x / hX)]++;
    for (i = 0; i < nIni; i++) {
        int id = qt_new_node(&t);
        qnode *n = &t.nodes[id];
        n->ULx = (int)(hX * (float)i); n->ULy = 0; n->URx = (int)(hX * (float)(i + 1)); n->URy = 0;
        n->BLx = n->ULx; n->BLy = maxY - minY; n->BRx = n->URx; n->BRy = maxY - minY;
        n->begin = qt_alloc_idx(&t, cnt[i]); n->count = 0;
        qt_push_back(&t, id);
        root[i] = id;
    }
    for (i = 0; i < nkeys; i++) { qnode *n = &t.nodes[root[(int)(keys[i].x / hX)]]; t.idx[n->begin + n->count++] = i; }
    for (i = t.head; i >= 0;) {
        int nx = t.nodes[i].next;
        if (t.nodes[i].count == 1) t.nodes[i].noMore = 1;
        else if (t.nodes[i].count == 0) qt_erase(&t, i);
        i = nx;
    }
    int finish = 0, nexp_cap = 1024, n_exp = 0;
    size_id *exp_ = (size_id *)malloc(sizeof(size_id) * nexp_cap), *prev_ = (size_id *)malloc(sizeof(size_id) * nexp_cap);
#define PUSH_EXP(sz_, id_) do { if (n_exp == nexp_cap) { nexp_cap *= 2; exp_ = (size_id *)realloc(exp_, sizeof(size_id) * nexp_cap); \
        prev_ = (size_id *)realloc(prev_, sizeof(size_id) * nexp_cap); } exp_[n_exp].size = (sz_); exp_[n_exp].id = (id_); n_exp++; } while (0)
    while (!finish) {
        int prevSize = t.size, nToExpand = 0, lit = t.head, ch[4];
        n_exp = 0;
        while (lit >= 0) {
            if (t.nodes[lit].noMore) { lit = t.nodes[lit].next; continue; }
            qt_divide(&t, lit, ch);
            for (q = 0; q < 4; q++)
                if (ch[q] >= 0) {
                    qt_push_front(&t, ch[q]);
                    if (t.nodes[ch[q]].count > 1) { nToExpand++; PUSH_EXP(t.nodes[ch[q]].count, ch[q]); }
                }
            int nx = t.nodes[lit].next;
            qt_erase(&t, lit);
            lit = nx;
        }
        if (t.size >= N || t.size == prevSize) finish = 1;
        else if (t.size + nToExpand * 3 > N) {
            while (!finish) {
                prevSize = t.size;
                int n_prev = n_exp, j;
                memcpy(prev_, exp_, sizeof(size_id) * n_prev);
                n_exp = 0;
                qsort(prev_, n_prev, sizeof(size_id), cmp_size_id);
                for (j = n_prev - 1; j >= 0; j--) {
                    qt_divide(&t, prev_[j].id, ch);
                    for (q = 0; q < 4; q++)
                        if (ch[q] >= 0) {
                            qt_push_front(&t, ch[q]);
                            if (t.nodes[ch[q]].count > 1) PUSH_EXP(t.nodes[ch[q]].count, ch[q]);
                        }
                    qt_erase(&t, prev_[j].id);
                    if (t.size >= N) break;
                }
                if (t.size >= N || t.size == prevSize) finish = 1;
            }
        }
    }
#undef PUSH_EXP
    /* retain the best response of every node (first maximum), in list order */
    int m = 0;
    for (i = t.head; i >= 0; i = t.nodes[i].next) {
        const qnode *n = &t.nodes[i];
        int best = t.idx[n->begin], k;
        float maxR = keys[best].response;
        for (k = 1; k < n->count; k++) {
            int c = t.idx[n->begin + k];
            if (keys[c].response > maxR) { best = c; maxR = keys[c].response; }
        }
        sel[m++] = best;
    }
    free(root); free(cnt); free(exp_); free(prev_); free(t.nodes); free(t.idx);
    return m;
}

/* ---- cv::fastAtan2 (degrees) ------------------------------------------------------------------ */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale, p3 = -0.3258083974640975f * scale,
                p5 = 0.1555786518463281f * scale, p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

static float ic_angle(const uint8_t *center, int step, const int *umax)
{
    int m_01 = 0, m_10 = 0, u, v;
    for (u = -ORB_HALF_PATCH; u <= ORB_HALF_PATCH; ++u) m_10 += u * center[u];
    for (v = 1; v <= ORB_HALF_PATCH; ++v) {
        int v_sum = 0, d = umax[v];
        for (u = -d; u <= d; ++u) {
            int val_plus = center[u + v * step], val_minus = center[u - v * step];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* ---- GaussianBlur(7x7, sigma 2, REFLECT_101), legacy 8-bit fixed point (O2) ------------------- */
void orc_gauss7_kernel(int k[7])
{
    /* getGaussianKernel(7, 2, CV_32F): exp in double, normalised, stored as float; then * 256 */
    double t[7], sum = 0;
    int i;
    for (i = 0; i < 7; i++) { double x = i - 3; t[i] = exp(-0.5 / (2.0 * 2.0) * x * x); float f = (float)t[i]; sum += f; t[i] = f; }
    sum = 1. / sum;
    for (i = 0; i < 7; i++) { float f = (float)(t[i] * sum); k[i] = cv_round_f(f * 256.f); }
}

void orc_gauss_blur7(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch)
{
    int k[7], x, y, i;
    orc_gauss7_kernel(k);
    int *tmp = (int *)malloc(sizeof(int) * (size_t)w * h);
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            int s = 0;
            for (i = 0; i < 7; i++) s += k[i] * src[(size_t)y * spitch + refl101(x + i - 3, w)];
            tmp[(size_t)y * w + x] = s;
        }
    for (y = 0; y < h; y++)
        for (x = 0; x < w; x++) {
            int s = 0;
            for (i = 0; i < 7; i++) s += k[i] * tmp[(size_t)refl101(y + i - 3, h) * w + x];
            s = (s + (1 << 15)) >> 16;
            dst[(size_t)y * dpitch + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(tmp);
}

/* ---- computeOrbDescriptor --------------------------------------------------------------------- */
static void orb_descriptor(float angle_deg, const uint8_t *center, int step, uint8_t *desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    float angle = angle_deg * factorPI;
    float a = (float)cos(angle), b = (float)sin(angle);
    const signed char *pat = orc_bit_pattern_31;
    int i, k;
#define GET_VALUE(idx) center[cv_round_f(pat[2 * (idx)] * b + pat[2 * (idx) + 1] * a) * step + \
                              cv_round_f(pat[2 * (idx)] * a - pat[2 * (idx) + 1] * b)]
    for (i = 0; i < 32; ++i, pat += 32) {
        int val = 0;
        for (k = 0; k < 8; k++) {
            int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
}

/* ---- ORBextractor::operator() ------------------------------------------------------------------ */
int orc_orb_extract(const uint8_t *img, int w, int h, int pitch, int nfeatures, float scaleFactor, int nlevels,
                    int iniTh, int minTh, orc_keypoint *kps, uint8_t *desc, int cap, int *per_level /* 8 or NULL */)
{
    orb_pyr p;
    int level, n = 0, i;
    if (nlevels > 8) nlevels = 8;
    orb_pyramid_build(&p, img, w, h, pitch, nfeatures, scaleFactor, nlevels);
    for (level = 0; level < nlevels; level++) {
        const int W = p.w[level], H = p.h[level], lp = p.pitch[level];
        const int minBX = ORB_EDGE - 3, minBY = minBX, maxBX = W - ORB_EDGE + 3, maxBY = H - ORB_EDGE + 3;
        orb_cand *cand = NULL;
        int nc = (maxBX > minBX && maxBY > minBY) ? orb_cell_fast(&p, level, iniTh, minTh, &cand) : 0;
        int *sel = (int *)malloc(sizeof(int) * (nc > 0 ? nc : 1));
        int m = nc > 0 ? orb_distribute(cand, nc, minBX, maxBX, minBY, maxBY, p.quota[level], sel) : 0;
        if (per_level) per_level[level] = m;
        if (m > 0) {
            uint8_t *blur = (uint8_t *)malloc((size_t)W * H);
            orc_gauss_blur7(lvl_px(&p, level), W, H, lp, blur, W);
            const int scaledPatchSize = (int)(ORB_PATCH * p.scale[level]);
            for (i = 0; i < m && n < cap; i++, n++) {
                float x = cand[sel[i]].x + (float)minBX, y = cand[sel[i]].y + (float)minBY;
                orc_keypoint *kp = &kps[n];
                kp->angle = ic_angle(lvl_px(&p, level) + (size_t)cv_round_f(y) * lp + cv_round_f(x), lp, p.umax);
                orb_descriptor(kp->angle, blur + (size_t)cv_round_f(y) * W + cv_round_f(x), W, desc + (size_t)n * 32);
                kp->response = cand[sel[i]].response; kp->octave = level; kp->class_id = -1;
                kp->size = (float)scaledPatchSize;
                if (level != 0) { x *= p.scale[level]; y *= p.scale[level]; }
                kp->x = x; kp->y = y;
            }
            free(blur);
        }
        free(sel); free(cand);
    }
    orb_pyramid_free(&p);
    return n;
}

/* pieces for unit tests */
int orc_orb_pyramid_level(const uint8_t *img, int w, int h, int pitch, float scaleFactor, int nlevels, int level,
                          uint8_t *out, int *ow, int *oh)
{
    orb_pyr p;
    int y;
    orb_pyramid_build(&p, img, w, h, pitch, 2000, scaleFactor, nlevels);
    *ow = p.w[level]; *oh = p.h[level];
    if (out) for (y = 0; y < p.h[level]; y++) memcpy(out + (size_t)y * p.w[level], lvl_px(&p, level) + (size_t)y * p.pitch[level], p.w[level]);
    orb_pyramid_free(&p);
    return 0;
}

/* DistributeOctTree alone on caller-supplied candidates (x, y, response triples): the selected candidate indices in
 * list order (tests of the host mirror's ORBextractor::DistributeOctTree) */
int orc_orb_distribute(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, int *sel)
{
    orb_cand *c = (orb_cand *)malloc(sizeof(orb_cand) * (size_t)(n > 0 ? n : 1));
    int i, m;
    for (i = 0; i < n; i++) { c[i].x = xyr[3 * i]; c[i].y = xyr[3 * i + 1]; c[i].response = xyr[3 * i + 2]; }
    m = orb_distribute(c, n, minX, maxX, minY, maxY, N, sel);
    free(c);
    return m;
}

/* FAST candidates of one level before the quadtree (x, y relative to the 16-px border, response) */
int orc_orb_candidates(const uint8_t *img, int w, int h, int pitch, float scaleFactor, int nlevels, int level,
                       int iniTh, int minTh, float *out3, int cap)
{
    orb_pyr p;
    orb_cand *cand = NULL;
    int i, n;
    orb_pyramid_build(&p, img, w, h, pitch, 2000, scaleFactor, nlevels);
    n = orb_cell_fast(&p, level, iniTh, minTh, &cand);
    for (i = 0; i < n && i < cap; i++) { out3[3 * i] = cand[i].x; out3[3 * i + 1] = cand[i].y; out3[3 * i + 2] = cand[i].response; }
    free(cand);
    orb_pyramid_free(&p);
    return n;
}

/* ---- BruteForce-Hamming DescriptorMatcher::match ----------------------------------------------- */
void orc_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, float *dist)
{
    int i, j, k;
    for (i = 0; i < nq; i++) {
        int best = -1, bd = 1 << 30;
        for (j = 0; j < nt; j++) {
            int d = 0;
            for (k = 0; k < 32; k++) d += __builtin_popcount(q[i * 32 + k] ^ t[j * 32 + k]);
            if (d < bd) { bd = d; best = j; }
        }
        idx[i] = best; dist[i] = (float)bd;
    }
}

/* Tracking::ORB_Robust_Find_MuliImage_MatchedFeatures (src/tracking.cpp:534-581).  Outputs the
 * (t2_left, t1_left, t1_right) triplets; returns their number. */
int orc_orb_robust_match(const orc_keypoint *lastL, const uint8_t *dLastL, int nLastL, const orc_keypoint *lastR,
                         const uint8_t *dLastR, int nLastR, const orc_keypoint *curL, const uint8_t *dCurL, int nCurL,
                         double match_err, orc_pt2f *t2l, orc_pt2f *t1l, orc_pt2f *t1r)
{
    int des_index = nLastL < nLastR ? nLastL : nLastR, i, m = 0;
    if (nCurL < des_index) des_index = nCurL;
    if (des_index <= 0) return 0;
    int *i1 = (int *)malloc(sizeof(int) * nLastL * 2), *i2 = i1 + nLastL;
    float *d1 = (float *)malloc(sizeof(float) * nLastL * 2), *d2 = d1 + nLastL;
    orc_match_hamming(dLastL, nLastL, dLastR, nLastR, i1, d1);
    orc_match_hamming(dLastL, nLastL, dCurL, nCurL, i2, d2);
    double min_dist = 10000, max_dist = 0;
    for (i = 0; i < des_index; i++) {
        double dist = d1[i] > d2[i] ? d1[i] : d2[i];
        if (dist < min_dist) min_dist = dist;
        if (dist > max_dist) max_dist = dist;
    }
    const double thr = 2 * min_dist > 30.0 ? 2 * min_dist : 30.0;
    for (i = 0; i < des_index; i++) {
        if ((d1[i] <= thr) && (d2[i] <= thr) &&
            ((double)fabsf(lastL[i].y - lastR[i1[i]].y) < match_err)) {
            t1l[m].x = lastL[i].x; t1l[m].y = lastL[i].y;
            t1r[m].x = lastR[i1[i]].x; t1r[m].y = lastR[i1[i]].y;
            t2l[m].x = curL[i2[i]].x; t2l[m].y = curL[i2[i]].y;
            m++;
        }
    }
    free(i1); free(d1);
    return m;
}

/* Tracking::ORB_StereoF2F_PnP_Track (src/tracking.cpp:168-249) given both frames' features */
int orc_orb_track_step(const orc_track_params *prm, const orc_keypoint *lastL, const uint8_t *dLastL, int nLastL,
                       const orc_keypoint *lastR, const uint8_t *dLastR, int nLastR, const orc_keypoint *curL,
                       const uint8_t *dCurL, int nCurL, double pose[16], orc_step_result *res)
{
    memset(res, 0, sizeof(*res));
    res->n_prev_kps = nLastL; res->n_cur_kps = nCurL;
    int cap = nLastL > 0 ? nLastL : 1;
    orc_pt2f *t2l = (orc_pt2f *)malloc(sizeof(orc_pt2f) * cap * 3), *t1l = t2l + cap, *t1r = t1l + cap;
    int m = orc_orb_robust_match(lastL, dLastL, nLastL, lastR, dLastR, nLastR, curL, dCurL, nCurL,
                                 prm->feature_match_error, t2l, t1l, t1r);
    res->n_tracked = m;
    int ok = 0;
    if (m < prm->num_features_tracking) { res->fail_stage = 2; goto done; }
    {
        orc_pt3f *X = (orc_pt3f *)malloc(sizeof(orc_pt3f) * m);
        orc_triangulate(prm->P1, prm->P2, t1l, t1r, m, X, NULL);
        double K[9] = {prm->P1[0], prm->P1[1], prm->P1[2], prm->P1[4], prm->P1[5], prm->P1[6],
                       prm->P1[8], prm->P1[9], prm->P1[10]};
        orc_pnp_result pr;
        orc_pnp_ransac(X, t2l, m, K, prm->iterations, prm->reproj_err, (double)prm->confidence, &pr, NULL);
        free(X);
        res->n_inliers = pr.n_inliers;
        memcpy(res->rvec, pr.rvec, sizeof(pr.rvec)); memcpy(res->tvec, pr.tvec, sizeof(pr.tvec));
        memcpy(res->R, pr.R, sizeof(pr.R));
        if ((double)pr.n_inliers / (double)m < prm->inlier_rate) { res->fail_stage = 3; goto done; }
        /* ORB mode gates the translation with the configured minmove / maxmove (:215) */
        int g = orc_gate_and_accumulate(pr.R, pr.tvec, prm->min_t2, prm->max_t2, pose, res->T_rel_inv);
        if (g < 0) { res->fail_stage = -g; goto done; }
        ok = 1;
    }
done:
    res->ok = ok;
    free(t2l);
    return ok;
}

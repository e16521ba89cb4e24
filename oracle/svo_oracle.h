/*
 * svo_oracle.h -- CPU restatement ("oracle") of the stereo-VO hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (stereo-visual-odometry_amd/) never links,
 * imports or executes it and fails loudly when its HIP library is missing.
 *
 * PARITY UNPINNED: the reference (liuzhenboo/Stereo-Visual-Odometry) ships no
 * tests, golden vectors or fixtures, and all of its hot-path arithmetic lives
 * in OpenCV 3.x, which is neither vendored in /root/reference nor installed in
 * this image.  This file restates the published OpenCV 3.4
 * algorithms the reference calls, and is pinned only by
 * analytic known-answer tests (tests/test_oracle_*.py).  Deliberate canonical
 * choices that differ from upstream at the last-ulp level are marked
 * "CANONICAL:" next to the code and listed in DESIGN.md.
 *
 * All functions are plain C, single-threaded unless noted, deterministic, and
 * compiled with -ffp-contract=off so float results do not depend on FMA.
 */
#ifndef SVO_ORACLE_H
#define SVO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::KeyPoint field order (pt.x, pt.y, size, angle, response, octave, class_id). */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} orc_keypoint;

typedef struct { float x, y; } orc_pt2f;
typedef struct { float x, y, z; } orc_pt3f;

/* ---- a2: cv::FAST(img, kps, thr, nms) TYPE_9_16; reference src/tracking.cpp:94-113 ---- */
/* Returns the number of keypoints found (may exceed cap; only cap are written). */
int orc_fast9_16(const uint8_t *img, int w, int h, int pitch, int thr, int nms,
                 orc_keypoint *out, int cap);

/* ---- a3: pyramid + cv::calcOpticalFlowPyrLK; reference src/tracking.cpp:583-618 ---- */
#define ORC_LK_MAX_LEVELS 8
typedef struct {
    int nlevels;                       /* maxLevel + 1 actually built */
    int pad;                           /* REFLECT_101 border kept around every level (= win) */
    int w[ORC_LK_MAX_LEVELS], h[ORC_LK_MAX_LEVELS];
    int pitch[ORC_LK_MAX_LEVELS];      /* = w[l] + 2*pad */
    uint8_t *data[ORC_LK_MAX_LEVELS];  /* padded buffer; pixel (x,y) at data[(y+pad)*pitch + x+pad] */
} orc_pyramid;

/* buildOpticalFlowPyramid(img, pyr, winSize, maxLevel): level l+1 = pyrDown(level l). */
int  orc_pyramid_build(const uint8_t *img, int w, int h, int pitch, int win, int max_level,
                       orc_pyramid *pyr);
void orc_pyramid_free(orc_pyramid *pyr);
/* one pyrDown step (5x5 [1 4 6 4 1], BORDER_REFLECT_101, (s+128)>>8); dst is ((w+1)/2)x((h+1)/2) */
void orc_pyr_down(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch);

/* calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(win,win), max_level,
 *                      TermCriteria(COUNT+EPS, max_iter, eps), flags = 0, min_eig)
 * threads > 1 splits the point range over OpenMP threads (results do not depend on it). */
/* sensitivity switch (tests only): 0 = exact int64 sums (canonical, the parity target), 1 = upstream's
 * float accumulation in raster order, 2 = in the lane order of upstream's SSE2 block.  Process-wide. */
void orc_lk_set_accum(int mode);
int orc_lk_get_accum(void);
/* Version forks of the restated OpenCV callees (tests only; geom.c).  Value 0 = CANONICAL. */
enum { ORC_COMPAT_TRIANGULATE = 0,   /* 0: 4 x 4 system (3.4); 1: 6 x 4 system with the x P[1] - y P[0] rows (<= 3.3) */
       ORC_COMPAT_PNP_REFIT = 1,     /* 0: LM refit from the best hypothesis; 1: no refit (<= 3.2); 2: from the LAST evaluated
                                        hypothesis (3.4's shared rvec / tvec); 3: from the caller's zeros (DLT start-up skipped) */
       ORC_COMPAT_PNP_MINIMAL = 2,   /* npoints == 5: 0: one RANSAC round + LM; 1: 3.4's direct return of the kernel's EPnP pose */
       ORC_COMPAT_LK_LANES = 3,      /* = orc_lk_set_accum: 0 exact, 1 raster, 2 round-4 hybrid, 3 legacy CV_SSE2 block, 4 CV_SIMD128 block */
       ORC_COMPAT_KNOBS = 4 };
void orc_set_opencv_compat(int knob, int value);
int orc_get_opencv_compat(int knob);
void orc_lk_set_guard_log(uint32_t *log);   /* tools only: exactness masks per (point, level), see oracle/lk.c */
long orc_lk_guard_violations(long *checked); /* iterations whose guard held but a float chain total != its integer total (must be 0) */
void orc_lk_set_iter_log(int32_t *log);      /* tools only: iterations per (point, level) of the next orc_lk_track calls */
int orc_lk_track(const orc_pyramid *prev, const orc_pyramid *next,
                 const orc_pt2f *prev_pts, int n, orc_pt2f *next_pts, uint8_t *status,
                 int win, int max_iter, double eps, float min_eig, int threads);

/* ---- a4: Tracking::deleteBadmatchFeatures; reference src/tracking.cpp:623-660 ---- */
/* keep[i] = 1 iff point i survives; returns M.  Points are NOT moved. */
int orc_circular_keep(const orc_pt2f *p0, const orc_pt2f *p1, const orc_pt2f *p2,
                      const orc_pt2f *p3, const orc_pt2f *p0r,
                      const uint8_t *s0, const uint8_t *s1, const uint8_t *s2, const uint8_t *s3,
                      int n, double match_err, uint8_t *keep);

/* ---- a5: cv::triangulatePoints + convertPointsFromHomogeneous; src/tracking.cpp:288-294 ---- */
void orc_triangulate(const double P1[12], const double P2[12], const orc_pt2f *x1,
                     const orc_pt2f *x2, int n, orc_pt3f *out, float *out4 /* 4*n or NULL */);

/* ---- a6: cv::solvePnPRansac(..., useExtrinsicGuess=true, iters, reproj, conf, inliers,
 *          SOLVEPNP_ITERATIVE) + cv::Rodrigues; reference src/tracking.cpp:464-501 ---- */
typedef struct {
    double rvec[3], tvec[3];
    double R[9];
    int n_inliers;
    int ransac_iters;      /* hypotheses evaluated before the adaptive stop */
    int best_iter;         /* index of the winning hypothesis (-1: none) */
    int lm_iters;
    int ok;                /* solvePnPRansac's boolean result */
} orc_pnp_result;

/* cv::solvePnP(SOLVEPNP_P3P) on exactly four points (oracle/p3p.c): pws 4 x (X Y Z), us 4 x (u v) in pixels; 1 = solved */
int orc_p3p4(const double pws[12], const double us[8], double fx, double fy, double cx, double cy, double R[9], double t[3]);
int orc_solve_deg4(double a, double b, double c, double d, double e, double x[4]);    /* polynom_solver.cpp: real roots */
int orc_pnp_ransac(const orc_pt3f *obj, const orc_pt2f *img, int n, const double K[9],
                   int iterations, float reproj_err, double confidence,
                   orc_pnp_result *res, uint8_t *inlier_mask /* n or NULL */);

/* pieces exposed for unit tests */
/* One-sided Jacobi SVD as used by cv::SVD for doubles.  At is n x m row-major (the TRANSPOSE
 * of the m x n input, m >= n); on return its rows are the left singular vectors (if Vt given
 * or want_u), W the singular values (descending), Vt the n x n right singular vectors. */
void orc_jacobi_svd(double *At, int m, int n, double *W, double *Vt);
int  orc_epnp(const double *pws, const double *us, int n, double fu, double fv, double uc,
              double vc, double R[9], double t[3]);
void orc_rodrigues_vec2mat(const double r[3], double R[9], double dRdr[27] /* or NULL */);
void orc_rodrigues_mat2vec(const double R[9], double r[3]);
uint32_t orc_rng_next(uint64_t *state);

/* ---- a7: rotationMatrixToEulerAngles + gating + pose accumulation; src/tracking.cpp:305-329,
 *          440-463 ---- */
/* Returns 1 and right-multiplies pose (4x4 row-major) by inv([R t;0 1]) iff the motion passes
 * the gates; min_t2/max_t2 are the squared-translation bounds (LK mode: 0.0005^2, 100). */
int orc_gate_and_accumulate(const double R[9], const double t[3], double min_t2, double max_t2,
                            double pose[16], double T_rel_inv[16] /* or NULL */);

/* ---- a1: one LK-mode frame step (Tracking::LK_StereoF2F_PnP_Track, src/tracking.cpp:258-344),
 *          given the t-1 keypoints already detected ---- */
typedef struct {
    int n_prev_kps;      /* FAST keypoints of frame t-1 fed to LK */
    int n_cur_kps;       /* FAST keypoints of frame t (stored for the next step) */
    int n_tracked;       /* survivors of the circular match */
    int n_inliers;
    int ok;              /* Track() return value */
    int fail_stage;      /* 0 none, 1 <30 kps, 2 too few tracks, 3 inlier ratio, 4 rot gate, 5 trans gate */
    double rvec[3], tvec[3], R[9];
    double T_rel_inv[16];
} orc_step_result;

typedef struct {
    double P1[12], P2[12];
    double feature_match_error;
    int    num_features_tracking;
    double inlier_rate;
    int    iterations;
    float  reproj_err;
    float  confidence;
    int    fast_thr;
    double min_t2, max_t2;   /* squared translation gate; LK mode hard-codes 0.0005^2, 100 (:311) */
} orc_track_params;

/* threads: OpenMP threads for the LK point loop (1 = scalar port). */
int orc_lk_track_step(const orc_track_params *prm,
                      const uint8_t *prevL, const uint8_t *prevR,
                      const uint8_t *curL, const uint8_t *curR, int w, int h, int pitch,
                      const orc_keypoint *prev_kps, int n_prev,
                      orc_keypoint *cur_kps, int cur_cap,
                      double pose[16], orc_step_result *res,
                      orc_pt2f *tracks /* 4*n_prev (t1l,t1r,t2r,t2l compacted) or NULL */,
                      int threads);

/* ---- a8-a14: ORB path (reference src/ORBextractor.cpp, src/tracking.cpp:168-249, 502-581) ------ */
void orc_orb_setup(int nfeatures, float scale_factor, int nlevels, float *scale, float *inv_scale,
                   int *quota /* nlevels */, int *umax /* 16 */);
void orc_resize_linear_u8(const uint8_t *src, int sw, int sh, int spitch, uint8_t *dst, int dw, int dh,
                          int dpitch);
int  orc_orb_pyramid_level(const uint8_t *img, int w, int h, int pitch, float scaleFactor, int nlevels, int level,
                           uint8_t *out /* tight, may be NULL */, int *ow, int *oh);
int  orc_orb_distribute(const float *xyr, int n, int minX, int maxX, int minY, int maxY, int N, int *sel);
int  orc_orb_candidates(const uint8_t *img, int w, int h, int pitch, float scaleFactor, int nlevels, int level,
                        int iniTh, int minTh, float *out3, int cap);
float orc_fast_atan2(float y, float x);
void orc_gauss7_kernel(int k[7]);
void orc_gauss_blur7(const uint8_t *src, int w, int h, int spitch, uint8_t *dst, int dpitch);
/* ORBextractor::operator()(image, mask, keypoints, descriptors): returns the keypoint count;
 * desc is n x 32 bytes; per_level (8 ints, may be NULL) receives the keypoints kept per level. */
int  orc_orb_extract(const uint8_t *img, int w, int h, int pitch, int nfeatures, float scaleFactor, int nlevels,
                     int iniTh, int minTh, orc_keypoint *kps, uint8_t *desc, int cap, int *per_level);
/* DescriptorMatcher("BruteForce-Hamming")->match(query, train): first minimum per query row */
void orc_match_hamming(const uint8_t *q, int nq, const uint8_t *t, int nt, int *idx, float *dist);
int  orc_orb_robust_match(const orc_keypoint *lastL, const uint8_t *dLastL, int nLastL, const orc_keypoint *lastR,
                          const uint8_t *dLastR, int nLastR, const orc_keypoint *curL, const uint8_t *dCurL,
                          int nCurL, double match_err, orc_pt2f *t2l, orc_pt2f *t1l, orc_pt2f *t1r);
/* one ORB-mode frame step given the features of the last frame (L, R) and the current left image;
 * prm->min_t2 / max_t2 are minmove^2 / maxmove^2 (src/tracking.cpp:215) */
int  orc_orb_track_step(const orc_track_params *prm, const orc_keypoint *lastL, const uint8_t *dLastL, int nLastL,
                        const orc_keypoint *lastR, const uint8_t *dLastR, int nLastR, const orc_keypoint *curL,
                        const uint8_t *dCurL, int nCurL, double pose[16], orc_step_result *res);

#ifdef __cplusplus
}
#endif
#endif

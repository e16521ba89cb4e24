/*
 * oracle/track.c -- one LK-mode frame step: Tracking::LK_StereoF2F_PnP_Track
 * (reference src/tracking.cpp:258-344) with its callees restated in fast.c / lk.c / geom.c / pnp.c.
 * TEST INFRASTRUCTURE ONLY.
 */
#include "svo_oracle.h"
#include <stdlib.h>
#include <string.h>

int orc_lk_track_step(const orc_track_params *prm, const uint8_t *prevL, const uint8_t *prevR,
                      const uint8_t *curL, const uint8_t *curR, int w, int h, int pitch,
                      const orc_keypoint *prev_kps, int n_prev, orc_keypoint *cur_kps,
                      int cur_cap, double pose[16], orc_step_result *res, orc_pt2f *tracks,
                      int threads)
{
    const int WIN = 21, MAXLVL = 3, MAXIT = 30;
    const double EPS = 0.01;
    const float MINEIG = 0.001f;
    int i;
    memset(res, 0, sizeof(*res));
    res->n_prev_kps = n_prev;

    /* (1) Detect_OpenCVFASTFeatures on the current left image (:260); stored for the next step */
    int ncur = orc_fast9_16(curL, w, h, pitch, prm->fast_thr, 1, cur_kps, cur_cap);
    res->n_cur_kps = ncur;
    if (ncur < 30) { res->fail_stage = 1; return 0; }

    /* (2) matched_t1_left = positions of the last frame's features (:268-271) */
    int n = n_prev;
    orc_pt2f *t1l = (orc_pt2f *)malloc(sizeof(orc_pt2f) * (size_t)(n > 0 ? n : 1) * 5);
    orc_pt2f *t1r = t1l + n, *t2r = t1r + n, *t2l = t2r + n, *ret = t2l + n;
    uint8_t *st = (uint8_t *)malloc((size_t)(n > 0 ? n : 1) * 5);
    uint8_t *s1 = st, *s2 = s1 + n, *s3 = s2 + n, *s4 = s3 + n, *keep = s4 + n;
    for (i = 0; i < n; i++) { t1l[i].x = prev_kps[i].x; t1l[i].y = prev_kps[i].y; }

    /* (3) LK_Robust_Find_MuliImage_MatchedFeatures (:583-622): each call's OUTPUT array,
     * failed points included, is the next call's input */
    orc_pyramid pL1, pR1, pL2, pR2;
    orc_pyramid_build(prevL, w, h, pitch, WIN, MAXLVL, &pL1);
    orc_pyramid_build(prevR, w, h, pitch, WIN, MAXLVL, &pR1);
    orc_pyramid_build(curL, w, h, pitch, WIN, MAXLVL, &pL2);
    orc_pyramid_build(curR, w, h, pitch, WIN, MAXLVL, &pR2);
    orc_lk_track(&pL1, &pR1, t1l, n, t1r, s1, WIN, MAXIT, EPS, MINEIG, threads);
    orc_lk_track(&pR1, &pR2, t1r, n, t2r, s2, WIN, MAXIT, EPS, MINEIG, threads);
    orc_lk_track(&pR2, &pL2, t2r, n, t2l, s3, WIN, MAXIT, EPS, MINEIG, threads);
    orc_lk_track(&pL2, &pL1, t2l, n, ret, s4, WIN, MAXIT, EPS, MINEIG, threads);
    orc_pyramid_free(&pL1); orc_pyramid_free(&pR1); orc_pyramid_free(&pL2); orc_pyramid_free(&pR2);

    int m = orc_circular_keep(t1l, t1r, t2r, t2l, ret, s1, s2, s3, s4, n,
                              prm->feature_match_error, keep);
    /* stable compaction of the four point lists */
    int k = 0;
    for (i = 0; i < n; i++)
        if (keep[i]) { t1l[k] = t1l[i]; t1r[k] = t1r[i]; t2r[k] = t2r[i]; t2l[k] = t2l[i]; k++; }
    res->n_tracked = m;
    if (tracks) {
        memcpy(tracks, t1l, sizeof(orc_pt2f) * m);
        memcpy(tracks + n, t1r, sizeof(orc_pt2f) * m);
        memcpy(tracks + 2 * (size_t)n, t2r, sizeof(orc_pt2f) * m);
        memcpy(tracks + 3 * (size_t)n, t2l, sizeof(orc_pt2f) * m);
    }
    int ok = 0;
    if (m < prm->num_features_tracking) { res->fail_stage = 2; goto done; }

    /* (4) triangulatePoints(P1, P2, t1_left, t1_right) (:292-294) */
    orc_pt3f *X = (orc_pt3f *)malloc(sizeof(orc_pt3f) * m);
    orc_triangulate(prm->P1, prm->P2, t1l, t1r, m, X, NULL);

    /* (5) OpenCV_EstimatePose_PnP(P1, t2_left, X) (:299, :464-501); K = P1[:, :3] */
    double K[9] = {prm->P1[0], prm->P1[1], prm->P1[2], prm->P1[4], prm->P1[5], prm->P1[6],
                   prm->P1[8], prm->P1[9], prm->P1[10]};
    orc_pnp_result pr;
    orc_pnp_ransac(X, t2l, m, K, prm->iterations, prm->reproj_err, (double)prm->confidence, &pr,
                   NULL);
    free(X);
    res->n_inliers = pr.n_inliers;
    memcpy(res->rvec, pr.rvec, sizeof(pr.rvec));
    memcpy(res->tvec, pr.tvec, sizeof(pr.tvec));
    memcpy(res->R, pr.R, sizeof(pr.R));
    if ((double)pr.n_inliers / (double)m < prm->inlier_rate) { res->fail_stage = 3; goto done; }

    /* (6,7) Euler + translation gates, frame_pose_ *= T^-1 (:305-329); LK mode hard-codes the
     * translation window 0.0005^2 < |t|^2 < 100 (:311); the caller passes it in prm */
    int g = orc_gate_and_accumulate(pr.R, pr.tvec, prm->min_t2, prm->max_t2, pose, res->T_rel_inv);
    if (g < 0) { res->fail_stage = -g; goto done; }
    ok = 1;
done:
    res->ok = ok;
    free(t1l); free(st);
    return ok;
}

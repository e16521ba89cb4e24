/*
 * oracle/pnp.c -- CPU restatement of
 *   cv::solvePnPRansac(obj, img, K, dist=0, rvec, tvec, useExtrinsicGuess=true, iterationsCount,
 *                      reprojectionError, confidence, inliers, SOLVEPNP_ITERATIVE) + cv::Rodrigues
 * as called by Tracking::OpenCV_EstimatePose_PnP (reference src/tracking.cpp:464-501).
 * TEST INFRASTRUCTURE ONLY.  Follows SURVEY.md Appendix A.6:
 *   OpenCV 3.4 calib3d: solvepnp.cpp (solvePnPRansac, PnPRansacCallback), ptsetreg.cpp
 *   (RANSACPointSetRegistrator, RANSACUpdateNumIters), epnp.cpp (Lepetit/Moreno-Noguer/Fua EPnP),
 *   calibration.cpp (cvProjectPoints2, cvFindExtrinsicCameraParams2), compat_ptsetreg.cpp
 *   (CvLevMarq), core rand.cpp (RNG = multiply-with-carry).
 *
 * CANONICAL choices (documented in DESIGN.md):
 *  C1 hypotheses are scored with the R EPnP produced; upstream converts R -> rvec -> R through
 *     Rodrigues (acos/sin/cos) first, which changes R by ~1e-16 and can only flip measure-zero
 *     borderline inliers.
 *  C2 the final Levenberg-Marquardt refit on the inlier set starts from the BEST hypothesis
 *     (3.1-3.3 restart from DLT, 3.4 from the last evaluated hypothesis; all converge to the same
 *     minimiser of the reprojection error over the inlier set).  orc_set_opencv_compat(ORC_COMPAT_PNP_REFIT, v)
 *     switches the start (C9: 1 = no refit at all, as before 3.3; 2 = the LAST evaluated hypothesis, 3.4's shared
 *     rvec / tvec; 3 = the caller's zero guess), ORC_COMPAT_PNP_MINIMAL the npoints == 5 case (C10: 1 = 3.4's
 *     early return of the kernel's EPnP pose with every point an inlier).  Tests only; value 0 is the parity target.
 *  C3 lambda = 10^k comes from an exact table instead of exp(k*log(10)).
 *  C4 J^T J and J^T e are summed sequentially in point order (upstream: blocked gemm order).
 */
#include "svo_oracle.h"
#include "orc_internal.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

/* cv::RNG::next(): state = (uint32)state * 4164903690 + (state >> 32) */
uint32_t orc_rng_next(uint64_t *state)
{
    *state = (uint64_t)(uint32_t)(*state) * 4164903690u + (uint32_t)(*state >> 32);
    return (uint32_t)(*state);
}

static double dot3(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double dist2_3(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

/* dst(n x n) = src^T src for src m x n (cvMulTransposed(src, dst, 1)): upper triangle summed over
 * rows in order, then mirrored. */
static void mul_transposed(const double *src, int m, int n, double *dst)
{
    int i, j, k;
    for (i = 0; i < n; i++)
        for (j = i; j < n; j++) {
            double s = 0;
            for (k = 0; k < m; k++) s += src[k * n + i] * src[k * n + j];
            dst[i * n + j] = s;
        }
    for (i = 0; i < n; i++) for (j = 0; j < i; j++) dst[i * n + j] = dst[j * n + i];
}

/* symmetric SVD with U^T output: cvSVD(&A, &D, &Ut, 0, CV_SVD_MODIFY_A | CV_SVD_U_T) */
static void svd_sym_ut(const double *A, int n, double *D, double *Ut)
{
    double Vt[ORC_SVD_MAXN * ORC_SVD_MAXN];
    int i, j;
    for (i = 0; i < n; i++) for (j = 0; j < n; j++) Ut[j * n + i] = A[i * n + j];
    orc_jacobi_svd(Ut, n, n, D, Vt);
}

/* epnp::qr_solve: Householder QR least squares for the 6x4 Gauss-Newton system.  Note the
 * column-max scan starts at A[k][k] and reads rows k..nr-2 (upstream quirk, kept). */
static void epnp_qr_solve(double *A, int nr, int nc, double *b, double *X)
{
    double A1[8], A2[8];
    int i, j, k;
    for (k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
        for (i = k + 1; i < nr; i++) {
            double elt = fabs(A[(i - 1) * nc + k]);
            if (eta < elt) eta = elt;
        }
        if (eta == 0) { A1[k] = A2[k] = 0.0; return; }
        double sum2 = 0.0, inv_eta = 1. / eta;
        for (i = k; i < nr; i++) {
            A[i * nc + k] *= inv_eta;
            sum2 += A[i * nc + k] * A[i * nc + k];
        }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (j = k + 1; j < nc; j++) {
            double sum = 0;
            for (i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            double tau = sum / A1[k];
            for (i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    /* b <- Q^T b */
    for (j = 0; j < nc; j++) {
        double tau = 0;
        for (i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        for (i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    /* X = R^-1 b */
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (i = nc - 2; i >= 0; i--) {
        double sum = 0;
        for (j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

typedef struct {
    int n;
    double fu, fv, uc, vc;
    const double *pws, *us;
    double *alphas, *pcs;
    double cws[4][3], ccs[4][3];
} epnp_t;

static void epnp_choose_control_points(epnp_t *e)
{
    int i, j, n = e->n;
    e->cws[0][0] = e->cws[0][1] = e->cws[0][2] = 0;
    for (i = 0; i < n; i++) for (j = 0; j < 3; j++) e->cws[0][j] += e->pws[3 * i + j];
    for (j = 0; j < 3; j++) e->cws[0][j] /= n;
    double *PW0 = (double *)malloc(sizeof(double) * 3 * n);
    for (i = 0; i < n; i++) for (j = 0; j < 3; j++) PW0[3 * i + j] = e->pws[3 * i + j] - e->cws[0][j];
    double pw0tpw0[9], dc[3], uct[9];
    mul_transposed(PW0, n, 3, pw0tpw0);
    svd_sym_ut(pw0tpw0, 3, dc, uct);
    free(PW0);
    for (i = 1; i < 4; i++) {
        double k = sqrt(dc[i - 1] / n);
        for (j = 0; j < 3; j++) e->cws[i][j] = e->cws[0][j] + k * uct[3 * (i - 1) + j];
    }
}

static void epnp_barycentric(epnp_t *e)
{
    double cc[9], ci[9];
    int i, j;
    for (i = 0; i < 3; i++) for (j = 1; j < 4; j++) cc[3 * i + j - 1] = e->cws[j][i] - e->cws[0][i];
    orc_svd_invert(cc, 3, ci);
    for (i = 0; i < e->n; i++) {
        const double *pi = e->pws + 3 * i;
        double *a = e->alphas + 4 * i;
        for (j = 0; j < 3; j++)
            a[1 + j] = ci[3 * j] * (pi[0] - e->cws[0][0]) + ci[3 * j + 1] * (pi[1] - e->cws[0][1]) +
                       ci[3 * j + 2] * (pi[2] - e->cws[0][2]);
        a[0] = 1.0 - a[1] - a[2] - a[3];
    }
}

static void epnp_L_6x10(const double *ut, double *L)
{
    const double *v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    int i, j;
    for (i = 0; i < 4; i++) {
        int a = 0, b = 1;
        for (j = 0; j < 6; j++) {
            dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
            dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
            dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
            b++;
            if (b > 3) { a++; b = a + 1; }
        }
    }
    for (i = 0; i < 6; i++) {
        double *row = L + 10 * i;
        row[0] = dot3(dv[0][i], dv[0][i]);
        row[1] = 2.0 * dot3(dv[0][i], dv[1][i]);
        row[2] = dot3(dv[1][i], dv[1][i]);
        row[3] = 2.0 * dot3(dv[0][i], dv[2][i]);
        row[4] = 2.0 * dot3(dv[1][i], dv[2][i]);
        row[5] = dot3(dv[2][i], dv[2][i]);
        row[6] = 2.0 * dot3(dv[0][i], dv[3][i]);
        row[7] = 2.0 * dot3(dv[1][i], dv[3][i]);
        row[8] = 2.0 * dot3(dv[2][i], dv[3][i]);
        row[9] = dot3(dv[3][i], dv[3][i]);
    }
}

/* betas10 = [B11 B12 B22 B13 B23 B33 B14 B24 B34 B44] */
static void epnp_betas_approx(const double *L, const double *rho, int which, double *betas)
{
    static const int cols1[4] = {0, 1, 3, 6}, cols2[3] = {0, 1, 2}, cols3[5] = {0, 1, 2, 3, 4};
    const int *cols = which == 1 ? cols1 : which == 2 ? cols2 : cols3;
    int nc = which == 1 ? 4 : which == 2 ? 3 : 5, i, j;
    double Lr[6 * 5], b[5];
    for (i = 0; i < 6; i++) for (j = 0; j < nc; j++) Lr[i * nc + j] = L[10 * i + cols[j]];
    orc_svd_solve(Lr, 6, nc, rho, b);
    if (which == 1) {
        if (b[0] < 0) {
            betas[0] = sqrt(-b[0]);
            betas[1] = -b[1] / betas[0]; betas[2] = -b[2] / betas[0]; betas[3] = -b[3] / betas[0];
        } else {
            betas[0] = sqrt(b[0]);
            betas[1] = b[1] / betas[0]; betas[2] = b[2] / betas[0]; betas[3] = b[3] / betas[0];
        }
        return;
    }
    if (b[0] < 0) {
        betas[0] = sqrt(-b[0]);
        betas[1] = (b[2] < 0) ? sqrt(-b[2]) : 0.0;
    } else {
        betas[0] = sqrt(b[0]);
        betas[1] = (b[2] > 0) ? sqrt(b[2]) : 0.0;
    }
    if (b[1] < 0) betas[0] = -betas[0];
    betas[2] = which == 3 ? b[3] / betas[0] : 0.0;
    betas[3] = 0.0;
}

static void epnp_gauss_newton(const double *L, const double *rho, double betas[4])
{
    int k, i;
    for (k = 0; k < 5; k++) {
        double A[24], b[6], x[4];
        for (i = 0; i < 6; i++) {
            const double *r = L + i * 10;
            double *ra = A + i * 4;
            ra[0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            ra[1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            ra[2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            ra[3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho[i] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] +
                             r[2] * betas[1] * betas[1] + r[3] * betas[0] * betas[2] +
                             r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                             r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] +
                             r[8] * betas[2] * betas[3] + r[9] * betas[3] * betas[3]);
        }
        x[0] = x[1] = x[2] = x[3] = 0;   /* qr_solve leaves X untouched on a singular column */
        epnp_qr_solve(A, 6, 4, b, x);
        for (i = 0; i < 4; i++) betas[i] += x[i];
    }
}

static double epnp_R_and_t(epnp_t *e, const double *ut, const double *betas, double R[9], double t[3])
{
    int i, j, k, n = e->n;
    /* compute_ccs */
    for (i = 0; i < 4; i++) e->ccs[i][0] = e->ccs[i][1] = e->ccs[i][2] = 0.0;
    for (i = 0; i < 4; i++) {
        const double *v = ut + 12 * (11 - i);
        for (j = 0; j < 4; j++) for (k = 0; k < 3; k++) e->ccs[j][k] += betas[i] * v[3 * j + k];
    }
    /* compute_pcs */
    for (i = 0; i < n; i++) {
        const double *a = e->alphas + 4 * i;
        double *pc = e->pcs + 3 * i;
        for (j = 0; j < 3; j++)
            pc[j] = a[0] * e->ccs[0][j] + a[1] * e->ccs[1][j] + a[2] * e->ccs[2][j] + a[3] * e->ccs[3][j];
    }
    /* solve_for_sign */
    if (e->pcs[2] < 0.0) {
        for (i = 0; i < 4; i++) for (j = 0; j < 3; j++) e->ccs[i][j] = -e->ccs[i][j];
        for (i = 0; i < 3 * n; i++) e->pcs[i] = -e->pcs[i];
    }
    /* estimate_R_and_t */
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (i = 0; i < n; i++) for (j = 0; j < 3; j++) { pc0[j] += e->pcs[3 * i + j]; pw0[j] += e->pws[3 * i + j]; }
    for (j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0}, abt_d[3], abt_ut[9], abt_vt[9];
    for (i = 0; i < n; i++) {
        const double *pc = e->pcs + 3 * i, *pw = e->pws + 3 * i;
        for (j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    /* cvSVD(&ABt, &D, &U, &V, CV_SVD_MODIFY_A): U and V (not transposed) */
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) abt_ut[j * 3 + i] = abt[i * 3 + j];
    orc_jacobi_svd(abt_ut, 3, 3, abt_d, abt_vt);
    /* R[i][j] = dot(U row i, V row j) = sum_k U[i][k] V[j][k];  U[i][k] = abt_ut[k][i], V[j][k] = abt_vt[k][j] */
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++)
            R[i * 3 + j] = abt_ut[0 * 3 + i] * abt_vt[0 * 3 + j] + abt_ut[1 * 3 + i] * abt_vt[1 * 3 + j] +
                           abt_ut[2 * 3 + i] * abt_vt[2 * 3 + j];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] -
                       R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - dot3(R, pw0);
    t[1] = pc0[1] - dot3(R + 3, pw0);
    t[2] = pc0[2] - dot3(R + 6, pw0);
    /* reprojection_error */
    double sum2 = 0.0;
    for (i = 0; i < n; i++) {
        const double *pw = e->pws + 3 * i;
        double Xc = dot3(R, pw) + t[0], Yc = dot3(R + 3, pw) + t[1];
        double inv_Zc = 1.0 / (dot3(R + 6, pw) + t[2]);
        double ue = e->uc + e->fu * Xc * inv_Zc, ve = e->vc + e->fv * Yc * inv_Zc;
        double u = e->us[2 * i], v = e->us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / n;
}

/* epnp::compute_pose.  pws: n x 3 world points, us: n x 2 pixel coordinates. */
int orc_epnp(const double *pws, const double *us, int n, double fu, double fv, double uc,
             double vc, double Rout[9], double tout[3])
{
    epnp_t e;
    int i, j;
    e.n = n; e.fu = fu; e.fv = fv; e.uc = uc; e.vc = vc; e.pws = pws; e.us = us;
    e.alphas = (double *)malloc(sizeof(double) * 4 * n);
    e.pcs = (double *)malloc(sizeof(double) * 3 * n);
    epnp_choose_control_points(&e);
    epnp_barycentric(&e);

    double *M = (double *)malloc(sizeof(double) * 2 * n * 12);
    for (i = 0; i < n; i++) {
        const double *as = e.alphas + 4 * i;
        double *M1 = M + (2 * i) * 12, *M2 = M1 + 12, u = us[2 * i], v = us[2 * i + 1];
        for (j = 0; j < 4; j++) {
            M1[3 * j] = as[j] * fu; M1[3 * j + 1] = 0.0; M1[3 * j + 2] = as[j] * (uc - u);
            M2[3 * j] = 0.0; M2[3 * j + 1] = as[j] * fv; M2[3 * j + 2] = as[j] * (vc - v);
        }
    }
    double mtm[144], d[12], ut[144];
    mul_transposed(M, 2 * n, 12, mtm);
    svd_sym_ut(mtm, 12, d, ut);
    free(M);

    double L[60], rho[6];
    epnp_L_6x10(ut, L);
    rho[0] = dist2_3(e.cws[0], e.cws[1]); rho[1] = dist2_3(e.cws[0], e.cws[2]);
    rho[2] = dist2_3(e.cws[0], e.cws[3]); rho[3] = dist2_3(e.cws[1], e.cws[2]);
    rho[4] = dist2_3(e.cws[1], e.cws[3]); rho[5] = dist2_3(e.cws[2], e.cws[3]);

    double Betas[4][4], rep[4], Rs[4][9], ts[4][3];
    int N;
    for (N = 1; N <= 3; N++) {
        epnp_betas_approx(L, rho, N, Betas[N]);
        epnp_gauss_newton(L, rho, Betas[N]);
        rep[N] = epnp_R_and_t(&e, ut, Betas[N], Rs[N], ts[N]);
    }
    N = 1;
    if (rep[2] < rep[1]) N = 2;
    if (rep[3] < rep[N]) N = 3;
    memcpy(Rout, Rs[N], sizeof(double) * 9);
    memcpy(tout, ts[N], sizeof(double) * 3);
    free(e.alphas); free(e.pcs);
    return 1;
}

/* RANSACUpdateNumIters(p, ep, modelPoints, maxIters) */
static int ransac_update_iters(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > DBL_MIN ? 1. - p : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)lrint(num / denom);
}

/* PnPRansacCallback::computeError for one point: cvProjectPoints2 in double, stored to float,
 * then the float squared distance. */
static float reproj_err2(const double R[9], const double t[3], double fx, double fy, double cx,
                         double cy, orc_pt3f P, orc_pt2f m)
{
    double X = P.x, Y = P.y, Z = P.z;
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
    float dx = m.x - px, dy = m.y - py;
    float s = 0.f;
    s += dx * dx;
    s += dy * dy;
    return s;
}

/* One LM evaluation: err = proj - m (2M), optionally J (2M x 6). */
static void lm_project(const double param[6], const double *Xw, const double *mm, int M, double fx,
                       double fy, double cx, double cy, double *err, double *J)
{
    double R[9], dRdr[27];
    int i, j;
    orc_rodrigues_vec2mat(param, R, J ? dRdr : NULL);
    const double *t = param + 3;
    for (i = 0; i < M; i++) {
        double X = Xw[3 * i], Y = Xw[3 * i + 1], Z = Xw[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        err[2 * i] = (x * fx + cx) - mm[2 * i];
        err[2 * i + 1] = (y * fy + cy) - mm[2 * i + 1];
        if (J) {
            double *Jx = J + (size_t)(2 * i) * 6, *Jy = Jx + 6;
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2],
                               X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                               X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5],
                               X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                               X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8],
                               X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                               X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                Jx[j] = fx * dxdr; Jy[j] = fy * dydr;
                Jx[3 + j] = fx * dxdt[j]; Jy[3 + j] = fy * dydt[j];
            }
        }
    }
}

static double vec_norm(const double *v, int n)
{
    double s = 0; int i;
    for (i = 0; i < n; i++) s += v[i] * v[i];
    return sqrt(s);
}

/* cvFindExtrinsicCameraParams2(useExtrinsicGuess = 1) + CvLevMarq(6, 2M, (EPS+ITER, 20,
 * FLT_EPSILON), completeSymmFlag = true).  Returns the number of accepted iterations. */
static int lm_refine(double param[6], const double *Xw, const double *mm, int M, double fx,
                     double fy, double cx, double cy)
{
    static const double POW10[33] = {
        1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4,
        1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12,
        1e13, 1e14, 1e15, 1e16};
    const int max_iter = 20;
    const double epsilon = FLT_EPSILON;
    double *J = (double *)malloc(sizeof(double) * (size_t)2 * M * 6);
    double *err = (double *)malloc(sizeof(double) * (size_t)2 * M);
    double JtJ[36], JtErr[6], prevParam[6];
    double prevErrNorm = DBL_MAX, errNorm;
    int lambdaLg10 = -3, iters = 0, i, j, k;

    for (;;) {
        /* state CALC_J: evaluate J and err at param, form the normal equations, take a step */
        lm_project(param, Xw, mm, M, fx, fy, cx, cy, err, J);
        for (i = 0; i < 6; i++)
            for (j = i; j < 6; j++) {
                double s = 0;
                for (k = 0; k < 2 * M; k++) s += J[(size_t)k * 6 + i] * J[(size_t)k * 6 + j];
                JtJ[i * 6 + j] = s;
            }
        for (i = 0; i < 6; i++) for (j = 0; j < i; j++) JtJ[i * 6 + j] = JtJ[j * 6 + i];
        for (i = 0; i < 6; i++) {
            double s = 0;
            for (k = 0; k < 2 * M; k++) s += J[(size_t)k * 6 + i] * err[k];
            JtErr[i] = s;
        }
        memcpy(prevParam, param, sizeof(prevParam));
        if (iters == 0) prevErrNorm = vec_norm(err, 2 * M);
        int done = 0;
        for (;;) {
            /* CvLevMarq::step(): (JtJ with diag *= 1 + lambda) x = JtErr by SVD; param = prev - x */
            double A[36], x[6], lambda = POW10[lambdaLg10 + 16];
            memcpy(A, JtJ, sizeof(A));
            for (i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
            orc_svd_solve(A, 6, 6, JtErr, x);
            for (i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
            /* state CHECK_ERR */
            lm_project(param, Xw, mm, M, fx, fy, cx, cy, err, NULL);
            errNorm = vec_norm(err, 2 * M);
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) continue;      /* retry with a larger lambda */
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            double dn = 0, pn = 0;
            for (i = 0; i < 6; i++) {
                dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
                pn += prevParam[i] * prevParam[i];
            }
            /* cvNorm(param, prevParam, CV_RELATIVE_L2) = |param - prev| / |prev| */
            if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < epsilon) done = 1;
            prevErrNorm = errNorm;
            break;
        }
        if (done) break;
    }
    free(J); free(err);
    return iters;
}

int orc_pnp_ransac(const orc_pt3f *obj, const orc_pt2f *img, int n, const double K[9],
                   int iterations, float reproj_err, double confidence, orc_pnp_result *res,
                   uint8_t *inlier_mask)
{
    /* "if (npoints == 4) { model_points = 4; ransac_kernel_method = SOLVEPNP_P3P; }" (solvepnp.cpp): with exactly four points
     * the minimal solver is P3P (oracle/p3p.c) on all of them, every point is an inlier, and the refit runs on the four */
    const int model_points = n == 4 ? 4 : 5;
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    int i, iter;
    memset(res, 0, sizeof(*res));
    res->best_iter = -1;
    res->R[0] = res->R[4] = res->R[8] = 1;      /* rvec = 0 -> Rodrigues gives I */
    if (n < model_points) return 0;             /* cv::solvePnPRansac asserts npoints >= 4 */
    const int refit_mode = orc_get_opencv_compat(ORC_COMPAT_PNP_REFIT);
    const int minimal_direct = orc_get_opencv_compat(ORC_COMPAT_PNP_MINIMAL) == 1 && n == model_points;

    uint8_t *mask = (uint8_t *)malloc(n), *best_mask = (uint8_t *)calloc(n, 1);
    double bestR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, bestt[3] = {0, 0, 0};
    double lastR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, lastt[3] = {0, 0, 0};
    int max_good = 0, niters = iterations > 1 ? iterations : 1;
    const float thr2 = (float)((double)reproj_err * (double)reproj_err);
    uint64_t rng = (uint64_t)-1;

    for (iter = 0; iter < niters; iter++) {
        int idx[5];
        double pws[15], us[10], R[9], t[3];
        if (n > model_points) {
            /* getSubset: draw 5 distinct indices, re-drawing the i-th until unique */
            for (i = 0; i < model_points;) {
                int j, idx_i;
                for (;;) {
                    idx_i = idx[i] = (int)(orc_rng_next(&rng) % (unsigned)n);
                    for (j = 0; j < i; j++) if (idx_i == idx[j]) break;
                    if (j == i) break;
                }
                i++;
            }
        } else {
            for (i = 0; i < model_points; i++) idx[i] = i;
        }
        for (i = 0; i < model_points; i++) {
            orc_pt3f P = obj[idx[i]];
            orc_pt2f m = img[idx[i]];
            pws[3 * i] = P.x; pws[3 * i + 1] = P.y; pws[3 * i + 2] = P.z;
            /* solvePnP(EPNP): undistortPoints (float out, zero distortion) then epnp::init_points
             * maps back with x*fu + uc */
            float xn = (float)(((double)m.x - cx) * (1. / fx));
            float yn = (float)(((double)m.y - cy) * (1. / fy));
            us[2 * i] = (double)xn * fx + cx;
            us[2 * i + 1] = (double)yn * fy + cy;
        }
        if (model_points == 4) {
            if (!orc_p3p4(pws, us, fx, fy, cx, cy, R, t)) { res->ransac_iters = iter + 1; break; }   /* runKernel: no model, and the same four points again would give none either */
        } else {
            orc_epnp(pws, us, model_points, fx, fy, cx, cy, R, t);
        }
        memcpy(lastR, R, sizeof(lastR)); memcpy(lastt, t, sizeof(lastt));

        int good = 0;
        if (n > model_points) {
            for (i = 0; i < n; i++) {
                int f = reproj_err2(R, t, fx, fy, cx, cy, obj[i], img[i]) <= thr2;
                mask[i] = (uint8_t)f;
                good += f;
            }
        } else {
            for (i = 0; i < n; i++) mask[i] = 1;
            good = n;
        }
        res->ransac_iters = iter + 1;
        if (n == model_points || good > (max_good > model_points - 1 ? max_good : model_points - 1)) {
            memcpy(best_mask, mask, n);
            memcpy(bestR, R, sizeof(bestR)); memcpy(bestt, t, sizeof(bestt));
            max_good = good;
            res->best_iter = iter;
            if (n == model_points) break;
            niters = ransac_update_iters(confidence, (double)(n - good) / n, model_points, niters);
        }
    }

    if (max_good <= 0) {
        /* solvePnPRansac returns false; rvec/tvec stay at the caller's zeros, no inliers */
        if (inlier_mask) memset(inlier_mask, 0, n);
        free(mask); free(best_mask);
        return 0;
    }

    /* refit on the inliers: solvePnP(ITERATIVE, useExtrinsicGuess = true) */
    double *Xw = (double *)malloc(sizeof(double) * 3 * max_good);
    double *mm = (double *)malloc(sizeof(double) * 2 * max_good);
    int M = 0;
    for (i = 0; i < n; i++)
        if (best_mask[i]) {
            Xw[3 * M] = obj[i].x; Xw[3 * M + 1] = obj[i].y; Xw[3 * M + 2] = obj[i].z;
            mm[2 * M] = img[i].x; mm[2 * M + 1] = img[i].y;
            M++;
        }
    double param[6];
    orc_rodrigues_mat2vec(refit_mode == 2 ? lastR : bestR, param);
    if (refit_mode == 2) { param[3] = lastt[0]; param[4] = lastt[1]; param[5] = lastt[2]; }
    else { param[3] = bestt[0]; param[4] = bestt[1]; param[5] = bestt[2]; }
    if (refit_mode == 3) memset(param, 0, sizeof(param));
    /* C9 = 1 / C10 = 1: the model the kernel produced IS the answer (rvec through Rodrigues, as the callback stores it) */
    res->lm_iters = (refit_mode == 1 || minimal_direct) ? 0 : lm_refine(param, Xw, mm, M, fx, fy, cx, cy);
    memcpy(res->rvec, param, sizeof(double) * 3);
    memcpy(res->tvec, param + 3, sizeof(double) * 3);
    orc_rodrigues_vec2mat(res->rvec, res->R, NULL);
    res->n_inliers = max_good;
    res->ok = 1;
    if (inlier_mask) memcpy(inlier_mask, best_mask, n);
    free(Xw); free(mm); free(mask); free(best_mask);
    return 1;
}

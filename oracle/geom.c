/*
 * oracle/geom.c -- small dense linear algebra + triangulation + Rodrigues + motion gating.
 * TEST INFRASTRUCTURE ONLY.
 *
 * Restates:
 *   - cv::SVD for doubles = one-sided Jacobi (OpenCV 3.4 modules/core/src/lapack.cpp
 *     JacobiSVDImpl_, SVBkSbImpl_), used by cvSVD / cvSolve(CV_SVD) / cvInvert(CV_SVD);
 *   - cv::triangulatePoints + cv::convertPointsFromHomogeneous (calib3d/triangulate.cpp,
 *     fundam.cpp), reference call sites src/tracking.cpp:292-294 (LK) and :190-192 (ORB);
 *   - cv::Rodrigues both directions (calib3d/calibration.cpp cvRodrigues2), reference call site
 *     src/tracking.cpp:488;
 *   - Tracking::rotationMatrixToEulerAngles + gates + pose product, src/tracking.cpp:305-329,
 *     440-463.
 *
 * CANONICAL: hypot(p, beta) in the Jacobi rotation is evaluated as sqrt(p*p + beta*beta) so the
 * result does not depend on a libm implementation (upstream calls hypot()).
 */
#include "svo_oracle.h"
#include "orc_internal.h"
#include <math.h>
#include <float.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * JacobiSVDImpl_<double>(At, astep, W, Vt, vstep, m, n, n1 = n, minval = DBL_MIN,
 *                        eps = DBL_EPSILON*10)
 * At: n rows of length m (row i = column i of the m x n matrix A).
 * ---------------------------------------------------------------------------------------- */
void orc_jacobi_svd(double *At, int m, int n, double *Wout, double *Vt)
{
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
    double W[ORC_SVD_MAXN];
    int i, j, k, iter, max_iter = m > 30 ? m : 30;

    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sd;
        if (Vt) {
            for (k = 0; k < n; k++) Vt[i * n + k] = 0;
            Vt[i * n + i] = 1;
        }
    }

    for (iter = 0; iter < max_iter; iter++) {
        int changed = 0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double *Ai = At + i * m, *Aj = At + j * m;
                double a = W[i], p = 0, b = W[j], c, s;
                for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);   /* CANONICAL hypot */
                if (beta < 0) {
                    double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
                for (k = 0; k < m; k++) {
                    double t0 = c * Ai[k] + s * Aj[k];
                    double t1 = -s * Ai[k] + c * Aj[k];
                    Ai[k] = t0; Aj[k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = 1;
                if (Vt) {
                    double *Vi = Vt + i * n, *Vj = Vt + j * n;
                    for (k = 0; k < n; k++) {
                        double t0 = c * Vi[k] + s * Vj[k];
                        double t1 = -s * Vi[k] + c * Vj[k];
                        Vi[k] = t0; Vj[k] = t1;
                    }
                }
            }
        if (!changed) break;
    }

    for (i = 0; i < n; i++) {
        double sd = 0;
        for (k = 0; k < m; k++) { double t = At[i * m + k]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    /* selection sort, descending; rows of At and Vt follow */
    for (i = 0; i < n - 1; i++) {
        j = i;
        for (k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (Vt) {
                for (k = 0; k < m; k++) { t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
                for (k = 0; k < n; k++) { t = Vt[i * n + k]; Vt[i * n + k] = Vt[j * n + k]; Vt[j * n + k] = t; }
            }
        }
    }
    for (i = 0; i < n; i++) Wout[i] = W[i];
    if (!Vt) return;

    /* left singular vectors: normalise the rows of At.  Upstream regenerates a vector from a
     * fixed-seed RNG when a singular value is <= DBL_MIN; CANONICAL: such rows are set to 0
     * (never reached by the hot path: SVBkSb skips singular values under its threshold). */
    for (i = 0; i < n; i++) {
        double sd = W[i];
        double s = sd > minval ? 1 / sd : 0.;
        for (k = 0; k < m; k++) At[i * m + k] *= s;
    }
}

/* SVBkSbImpl_<double> with nb == 1: x = V diag(1/w) U^T b, dropping w_i <= eps*sum(w). */
void orc_svd_backsubst_vec(int m, int n, const double *w, const double *Ut /* n x m */,
                           const double *Vt /* n x n */, const double *b, double *x)
{
    double threshold = 0;
    int i, j, nm = m < n ? m : n;
    for (i = 0; i < n; i++) x[i] = 0;
    for (i = 0; i < nm; i++) threshold += w[i];
    threshold *= DBL_EPSILON * 2;
    for (i = 0; i < nm; i++) {
        double wi = w[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (j = 0; j < m; j++) s += Ut[i * m + j] * b[j];
        s *= wi;
        for (j = 0; j < n; j++) x[j] = x[j] + s * Vt[i * n + j];
    }
}

/* cv::solve(A, b, x, DECOMP_SVD) for an m x n system (m >= n), single right-hand side. */
void orc_svd_solve(const double *A, int m, int n, const double *b, double *x)
{
    double At[ORC_SVD_MAXN * ORC_SVD_MAXM], W[ORC_SVD_MAXN], Vt[ORC_SVD_MAXN * ORC_SVD_MAXN];
    int i, j;
    for (i = 0; i < m; i++) for (j = 0; j < n; j++) At[j * m + i] = A[i * n + j];
    orc_jacobi_svd(At, m, n, W, Vt);
    orc_svd_backsubst_vec(m, n, W, At, Vt, b, x);
}

/* cv::invert(A, Ainv, DECOMP_SVD) for a square n x n matrix: SVD::compute then
 * SVD::backSubst(w, u, vt, Mat(), dst), i.e. SVBkSb with b == NULL, nb = m. */
void orc_svd_invert(const double *A, int n, double *Ainv)
{
    double At[ORC_SVD_MAXN * ORC_SVD_MAXN], W[ORC_SVD_MAXN], Vt[ORC_SVD_MAXN * ORC_SVD_MAXN];
    double buffer[ORC_SVD_MAXN], threshold = 0;
    int i, j, k;
    for (i = 0; i < n; i++) for (j = 0; j < n; j++) At[j * n + i] = A[i * n + j];
    orc_jacobi_svd(At, n, n, W, Vt);
    for (i = 0; i < n * n; i++) Ainv[i] = 0;
    for (i = 0; i < n; i++) threshold += W[i];
    threshold *= DBL_EPSILON * 2;
    for (i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        /* buffer[j] = u[j][i] * wi  (u column i == At row i) */
        for (j = 0; j < n; j++) buffer[j] = At[i * n + j] * wi;
        /* x[j][k] += v[j][i] * buffer[k]  (v column i == Vt row i) */
        for (j = 0; j < n; j++)
            for (k = 0; k < n; k++) Ainv[j * n + k] = Ainv[j * n + k] + Vt[i * n + j] * buffer[k];
    }
}

/* ------------------------------------------------------------------------------------------
 * Version forks of the restated OpenCV callees (tests only).  The reference asks for
 * "OpenCV 3" (CMakeLists.txt:17; the ROS build for 3.1) without pinning a patch release, and the callees changed
 * inside the 3.x line.  Knob 0 of every fork is the CANONICAL choice the HIP path is held to; the other values
 * restate what other releases compute, as recalled -- nothing here is checked against a binary (no OpenCV in this
 * image; tests/test_cv_crosscheck.py is the pin for the day one exists).
 * ---------------------------------------------------------------------------------------- */
static int g_compat[ORC_COMPAT_KNOBS];
void orc_set_opencv_compat(int knob, int value)
{
    if (knob == ORC_COMPAT_LK_LANES) { orc_lk_set_accum(value); return; }
    if (knob >= 0 && knob < ORC_COMPAT_KNOBS) g_compat[knob] = value;
}
int orc_get_opencv_compat(int knob)
{
    if (knob == ORC_COMPAT_LK_LANES) return orc_lk_get_accum();
    return (knob >= 0 && knob < ORC_COMPAT_KNOBS) ? g_compat[knob] : -1;
}

/* ------------------------------------------------------------------------------------------
 * cv::triangulatePoints(P1, P2, x1, x2) -> 4xN float; convertPointsFromHomogeneous -> Nx3 float
 * C8: ORC_COMPAT_TRIANGULATE = 0 (CANONICAL): the 4 x 4 system x P[2] - P[0], y P[2] - P[1] of both views (3.4's
 * cvTriangulatePoints as restated in round 1); = 1: the older 6 x 4 system with a third row x P[1] - y P[0] per view
 * (2.4 .. 3.3 as recalled: "matrA_dat[(j*3+2)*4+k] = x * P[1][k] - y * P[0][k]") -- the same null vector on exact data,
 * a slightly different least-squares point when the two rays do not meet.
 * ---------------------------------------------------------------------------------------- */
void orc_triangulate(const double P1[12], const double P2[12], const orc_pt2f *x1,
                     const orc_pt2f *x2, int n, orc_pt3f *out, float *out4)
{
    int i, k;
    const int rows_per_view = g_compat[ORC_COMPAT_TRIANGULATE] == 1 ? 3 : 2, m = 2 * rows_per_view;
    for (i = 0; i < n; i++) {
        double A[24], At[24], W[4], Vt[16];
        const double *P[2] = {P1, P2};
        double xs[2] = {(double)x1[i].x, (double)x2[i].x}, ys[2] = {(double)x1[i].y, (double)x2[i].y};
        int j;
        for (j = 0; j < 2; j++)
            for (k = 0; k < 4; k++) {
                A[(j * rows_per_view + 0) * 4 + k] = xs[j] * P[j][8 + k] - P[j][k];
                A[(j * rows_per_view + 1) * 4 + k] = ys[j] * P[j][8 + k] - P[j][4 + k];
                if (rows_per_view == 3) A[(j * 3 + 2) * 4 + k] = xs[j] * P[j][4 + k] - ys[j] * P[j][k];
            }
        for (j = 0; j < m; j++) for (k = 0; k < 4; k++) At[k * m + j] = A[j * 4 + k];
        orc_jacobi_svd(At, m, 4, W, Vt);
        float X4[4];
        for (k = 0; k < 4; k++) X4[k] = (float)Vt[12 + k];     /* last row of V^T, stored as f32 */
        if (out4) for (k = 0; k < 4; k++) out4[(size_t)k * n + i] = X4[k];
        float scale = X4[3] != 0.f ? 1.f / X4[3] : 1.f;
        out[i].x = X4[0] * scale; out[i].y = X4[1] * scale; out[i].z = X4[2] * scale;
    }
}

/* ------------------------------------------------------------------------------------------
 * cv::Rodrigues
 * ---------------------------------------------------------------------------------------- */
void orc_rodrigues_vec2mat(const double r[3], double R[9], double J[27])
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    int i, k;
    if (theta < DBL_EPSILON) {
        for (i = 0; i < 9; i++) R[i] = 0;
        R[0] = R[4] = R[8] = 1;
        if (J) {
            memset(J, 0, sizeof(double) * 27);
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    /* R = cos(theta)*I + (1 - cos(theta))*r*rT + sin(theta)*[r_x] */
    for (k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        static const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                          0, 0, 1, 0, 0, 0, -1, 0, 0,
                                          0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] +
                               a4 * d_r_x_[i * 9 + k];
        }
    }
}

void orc_rodrigues_mat2vec(const double Rin[9], double r[3])
{
    /* cvRodrigues2 first projects R onto SO(3): R = U * V^T of its SVD */
    double At[9], W[3], Vt[9], R[9];
    int i, j, k;
    for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) At[j * 3 + i] = Rin[i * 3 + j];
    orc_jacobi_svd(At, 3, 3, W, Vt);
    for (i = 0; i < 3; i++)
        for (j = 0; j < 3; j++) {
            double s = 0;
            for (k = 0; k < 3; k++) s += At[k * 3 + i] * Vt[k * 3 + j];   /* U[i][k] = At[k][i] */
            R[i * 3 + j] = s;
        }
    double rx = R[7] - R[5], ry = R[2] - R[6], rz = R[3] - R[1];
    double s = sqrt((rx * rx + ry * ry + rz * rz) * 0.25);
    double c = (R[0] + R[4] + R[8] - 1) * 0.5;
    c = c > 1. ? 1. : c < -1. ? -1. : c;
    double theta = acos(c);
    if (s < 1e-5) {
        if (c > 0) { rx = ry = rz = 0; }
        else {
            double t;
            t = (R[0] + 1) * 0.5; rx = sqrt(t > 0. ? t : 0.);
            t = (R[4] + 1) * 0.5; ry = sqrt(t > 0. ? t : 0.) * (R[1] < 0 ? -1. : 1.);
            t = (R[8] + 1) * 0.5; rz = sqrt(t > 0. ? t : 0.) * (R[2] < 0 ? -1. : 1.);
            if (fabs(rx) < fabs(ry) && fabs(rx) < fabs(rz) && (R[5] > 0) != (ry * rz > 0)) rz = -rz;
            theta /= sqrt(rx * rx + ry * ry + rz * rz);
            rx *= theta; ry *= theta; rz *= theta;
        }
    } else {
        double vth = 1 / (2 * s);
        vth *= theta;
        rx *= vth; ry *= vth; rz *= vth;
    }
    r[0] = rx; r[1] = ry; r[2] = rz;
}

/* ------------------------------------------------------------------------------------------
 * Gating + accumulation (reference src/tracking.cpp:305-329, 440-463)
 * ---------------------------------------------------------------------------------------- */
static void mat4_mul(const double A[16], const double B[16], double C[16])
{
    int i, j, k;
    for (i = 0; i < 4; i++)
        for (j = 0; j < 4; j++) {
            double s = 0;
            for (k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 4 + j];
            C[i * 4 + j] = s;
        }
}

int orc_gate_and_accumulate(const double R[9], const double t[3], double min_t2, double max_t2,
                            double pose[16], double Tinv_out[16])
{
    /* rotationMatrixToEulerAngles: sy, x, y, z are floats holding double expressions */
    float sy = (float)sqrt(R[0] * R[0] + R[3] * R[3]);
    int singular = sy < 1e-6;
    float ex, ey, ez;
    if (!singular) {
        ex = (float)atan2(R[7], R[8]);
        ey = (float)atan2(-R[6], (double)sy);
        ez = (float)atan2(R[3], R[0]);
    } else {
        ex = (float)atan2(-R[5], R[4]);
        ey = (float)atan2(-R[6], (double)sy);
        ez = 0;
    }
    /* "abs(rotation_euler[k]) < 0.1": float |.| compared against the double literal 0.1 */
    if (!((double)fabsf(ey) < 0.1 && (double)fabsf(ex) < 0.1 && (double)fabsf(ez) < 0.1)) return -4;
    double n2 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];   /* std::pow(x,2) == x*x exactly */
    if (!(n2 < max_t2 && n2 > min_t2)) return -5;
    /* T = [R t; 0 1]; T.inv() (cv::Mat::inv, DECOMP_LU).  CANONICAL: the rigid-body closed form
     * [R^T, -R^T t] is used instead of a 4x4 LU; R comes out of Rodrigues and is orthonormal to
     * ~1e-16, so the two agree to ~1e-15 relative. */
    double Ti[16] = {R[0], R[3], R[6], 0, R[1], R[4], R[7], 0, R[2], R[5], R[8], 0, 0, 0, 0, 1};
    Ti[3] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
    Ti[7] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
    Ti[11] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
    double P[16];
    mat4_mul(pose, Ti, P);
    memcpy(pose, P, sizeof(P));
    if (Tinv_out) memcpy(Tinv_out, Ti, sizeof(Ti));
    return 1;
}

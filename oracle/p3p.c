/*
 * oracle/p3p.c -- CPU restatement of OpenCV 3's P3P kernel (modules/calib3d/src/p3p.cpp, polynom_solver.cpp: Gao, Hou,
 * Tang, Cheng, "Complete solution classification for the perspective-three-point problem", the implementation cv::solvePnP
 * runs for SOLVEPNP_P3P).  cv::solvePnPRansac switches to it when it is handed exactly FOUR points (solvepnp.cpp: "else if
 * (npoints == 4) { model_points = 4; ransac_kernel_method = SOLVEPNP_P3P; }"), which the reference reaches when
 * num_features_tracking is lowered to 4 and a pair keeps exactly four tracks (src/tracking.cpp:274, 485).
 * TEST INFRASTRUCTURE ONLY.  Restated from memory like the rest of oracle/ (no OpenCV in this image): the known-answer tests
 * (tests/test_oracle_geometry.py: planted poses recovered, the quartic solver against numpy.roots) are what pins it.
 *
 * p3p::solve(R, t, 4 points): the three-point solver on points 0..2 (up to four poses), then the pose whose projection of
 * point 3 lies closest to its image point.
 */
#include "svo_oracle.h"
#include <math.h>
#include <string.h>

#define ORC_PI 3.1415926535897932384626433832795

/* polynom_solver.cpp */
static int solve_deg2(double a, double b, double c, double *x1, double *x2)
{
    double delta = b * b - 4 * a * c;
    if (delta < 0) return 0;
    double inv_2a = 0.5 / a;
    if (delta == 0) { *x1 = -b * inv_2a; *x2 = *x1; return 1; }
    double sqrt_delta = sqrt(delta);
    *x1 = (-b + sqrt_delta) * inv_2a;
    *x2 = (-b - sqrt_delta) * inv_2a;
    return 2;
}

static int solve_deg3(double a, double b, double c, double d, double *x0, double *x1, double *x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2(b, c, d, x0, x1);
    }
    double inv_a = 1. / a;
    double b_a = inv_a * b, b_a2 = b_a * b_a;
    double c_a = inv_a * c;
    double d_a = inv_a * d;
    double Q = (3 * c_a - b_a2) / 9;
    double R = (9 * b_a * c_a - 27 * d_a - 2 * b_a * b_a2) / 54;
    double Q3 = Q * Q * Q;
    double D = Q3 + R * R;
    double b_a_3 = (1. / 3.) * b_a;
    if (Q == 0) {
        if (R == 0) { *x0 = *x1 = *x2 = -b_a_3; return 3; }
        *x0 = pow(2 * R, 1 / 3.0) - b_a_3;
        return 1;
    }
    if (D <= 0) {
        double theta = acos(R / sqrt(-Q3));
        double sqrt_Q = sqrt(-Q);
        *x0 = 2 * sqrt_Q * cos(theta / 3.0) - b_a_3;
        *x1 = 2 * sqrt_Q * cos((theta + 2 * ORC_PI) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * ORC_PI) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

int orc_solve_deg4(double a, double b, double c, double d, double e, double x[4])
{
    if (a == 0) { x[3] = 0; return solve_deg3(b, c, d, e, &x[0], &x[1], &x[2]); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = solve_deg3(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
    if (n == 0) return 0;
    double R2 = 0.25 * b2 - c + r0, R;
    if (R2 < 0) return 0;
    R = sqrt(R2);
    double inv_R = 1. / R;
    int nb_real_roots = 0;
    double D2, E2;
    if (R < 10E-12) {
        double temp = r0 * r0 - 4 * e;
        if (temp < 0) D2 = E2 = -1;
        else {
            double sqrt_temp = sqrt(temp);
            D2 = 0.75 * b2 - 2 * c + 2 * sqrt_temp;
            E2 = D2 - 4 * sqrt_temp;
        }
    } else {
        double u = 0.75 * b2 - 2 * c - R2, v = 0.25 * inv_R * (4 * bc - 8 * d - b3);
        D2 = u + v;
        E2 = u - v;
    }
    double b_4 = 0.25 * b, R_2 = 0.5 * R;
    if (D2 >= 0) {
        double D = sqrt(D2);
        nb_real_roots = 2;
        double D_2 = 0.5 * D;
        x[0] = R_2 + D_2 - b_4;
        x[1] = x[0] - D;
    }
    if (E2 >= 0) {
        double E = sqrt(E2);
        double E_2 = 0.5 * E;
        if (nb_real_roots == 0) {
            x[0] = -R_2 + E_2 - b_4;
            x[1] = x[0] - E;
            nb_real_roots = 2;
        } else {
            x[2] = -R_2 + E_2 - b_4;
            x[3] = x[2] - E;
            nb_real_roots = 4;
        }
    }
    return nb_real_roots;
}

/* p3p::jacobi_4x4: eigenvalues D and eigenvectors (columns of U) of the symmetric 4 x 4 matrix A (Numerical Recipes' sweep) */
static int jacobi_4x4(double *A, double *D, double *U)
{
    double B[4], Z[4];
    int i, j, k, iter;
    for (i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1. : 0.;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    memcpy(D, B, sizeof(B));
    memset(Z, 0, sizeof(Z));
    for (iter = 0; iter < 50; iter++) {
        double sum = fabs(A[1]) + fabs(A[2]) + fabs(A[3]) + fabs(A[6]) + fabs(A[7]) + fabs(A[11]);
        if (sum == 0.0) return 1;
        double tresh = (iter < 3) ? 0.2 * sum / 16. : 0.0;
        for (i = 0; i < 3; i++) {
            double *pAij = A + 5 * i + 1;
            for (j = i + 1; j < 4; j++) {
                double Aij = *pAij;
                double eps_machine = 100.0 * fabs(Aij);
                if (iter > 3 && fabs(D[i]) + eps_machine == fabs(D[i]) && fabs(D[j]) + eps_machine == fabs(D[j]))
                    *pAij = 0.0;
                else if (fabs(Aij) > tresh) {
                    double hh = D[j] - D[i], t;
                    if (fabs(hh) + eps_machine == fabs(hh))
                        t = Aij / hh;
                    else {
                        double theta = 0.5 * hh / Aij;
                        t = 1.0 / (fabs(theta) + sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    hh = t * Aij;
                    Z[i] -= hh; Z[j] += hh;
                    D[i] -= hh; D[j] += hh;
                    *pAij = 0.0;
                    double c = 1.0 / sqrt(1 + t * t);
                    double s = t * c;
                    double tau = s / (1.0 + c);
                    for (k = 0; k <= i - 1; k++) {
                        double g = A[k * 4 + i], h = A[k * 4 + j];
                        A[k * 4 + i] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (k = i + 1; k <= j - 1; k++) {
                        double g = A[i * 4 + k], h = A[k * 4 + j];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[k * 4 + j] = h + s * (g - h * tau);
                    }
                    for (k = j + 1; k < 4; k++) {
                        double g = A[i * 4 + k], h = A[j * 4 + k];
                        A[i * 4 + k] = g - s * (h + g * tau);
                        A[j * 4 + k] = h + s * (g - h * tau);
                    }
                    for (k = 0; k < 4; k++) {
                        double g = U[k * 4 + i], h = U[k * 4 + j];
                        U[k * 4 + i] = g - s * (h + g * tau);
                        U[k * 4 + j] = h + s * (g - h * tau);
                    }
                }
                pAij++;
            }
        }
        for (i = 0; i < 4; i++) B[i] += Z[i];
        memcpy(D, B, sizeof(B));
        memset(Z, 0, sizeof(Z));
    }
    return 0;
}

/* p3p::align: the rigid motion that takes the three object points to M_end (Horn's quaternion method) */
static int p3p_align(double M_end[3][3], const double Xw[9], double R[3][3], double T[3])
{
    const double X0 = Xw[0], Y0 = Xw[1], Z0 = Xw[2], X1 = Xw[3], Y1 = Xw[4], Z1 = Xw[5], X2 = Xw[6], Y2 = Xw[7], Z2 = Xw[8];
    double C_start[3], C_end[3];
    int i, j;
    for (i = 0; i < 3; i++) C_end[i] = (M_end[0][i] + M_end[1][i] + M_end[2][i]) / 3;
    C_start[0] = (X0 + X1 + X2) / 3;
    C_start[1] = (Y0 + Y1 + Y2) / 3;
    C_start[2] = (Z0 + Z1 + Z2) / 3;
    double s[9];
    for (j = 0; j < 3; j++) {
        s[0 * 3 + j] = (X0 * M_end[0][j] + X1 * M_end[1][j] + X2 * M_end[2][j]) / 3 - C_end[j] * C_start[0];
        s[1 * 3 + j] = (Y0 * M_end[0][j] + Y1 * M_end[1][j] + Y2 * M_end[2][j]) / 3 - C_end[j] * C_start[1];
        s[2 * 3 + j] = (Z0 * M_end[0][j] + Z1 * M_end[1][j] + Z2 * M_end[2][j]) / 3 - C_end[j] * C_start[2];
    }
    double Qs[16], evs[4], U[16];
    Qs[0 * 4 + 0] = s[0 * 3 + 0] + s[1 * 3 + 1] + s[2 * 3 + 2];
    Qs[1 * 4 + 1] = s[0 * 3 + 0] - s[1 * 3 + 1] - s[2 * 3 + 2];
    Qs[2 * 4 + 2] = s[1 * 3 + 1] - s[2 * 3 + 2] - s[0 * 3 + 0];
    Qs[3 * 4 + 3] = s[2 * 3 + 2] - s[0 * 3 + 0] - s[1 * 3 + 1];
    Qs[1 * 4 + 0] = Qs[0 * 4 + 1] = s[1 * 3 + 2] - s[2 * 3 + 1];
    Qs[2 * 4 + 0] = Qs[0 * 4 + 2] = s[2 * 3 + 0] - s[0 * 3 + 2];
    Qs[3 * 4 + 0] = Qs[0 * 4 + 3] = s[0 * 3 + 1] - s[1 * 3 + 0];
    Qs[2 * 4 + 1] = Qs[1 * 4 + 2] = s[1 * 3 + 0] + s[0 * 3 + 1];
    Qs[3 * 4 + 1] = Qs[1 * 4 + 3] = s[2 * 3 + 0] + s[0 * 3 + 2];
    Qs[3 * 4 + 2] = Qs[2 * 4 + 3] = s[2 * 3 + 1] + s[1 * 3 + 2];
    jacobi_4x4(Qs, evs, U);
    int i_ev = 0;
    double ev_max = evs[i_ev];
    for (i = 1; i < 4; i++) if (evs[i] > ev_max) ev_max = evs[i_ev = i];
    double q[4];
    for (i = 0; i < 4; i++) q[i] = U[i * 4 + i_ev];
    double q02 = q[0] * q[0], q12 = q[1] * q[1], q22 = q[2] * q[2], q32 = q[3] * q[3];
    double q0_1 = q[0] * q[1], q0_2 = q[0] * q[2], q0_3 = q[0] * q[3];
    double q1_2 = q[1] * q[2], q1_3 = q[1] * q[3];
    double q2_3 = q[2] * q[3];
    R[0][0] = q02 + q12 - q22 - q32;
    R[0][1] = 2. * (q1_2 - q0_3);
    R[0][2] = 2. * (q1_3 + q0_2);
    R[1][0] = 2. * (q1_2 + q0_3);
    R[1][1] = q02 + q22 - q12 - q32;
    R[1][2] = 2. * (q2_3 - q0_1);
    R[2][0] = 2. * (q1_3 - q0_2);
    R[2][1] = 2. * (q2_3 + q0_1);
    R[2][2] = q02 + q32 - q12 - q22;
    for (i = 0; i < 3; i++) T[i] = C_end[i] - (R[i][0] * C_start[0] + R[i][1] * C_start[1] + R[i][2] * C_start[2]);
    return 1;
}

/* p3p::solve_for_lengths: the distances of the three points from the camera centre, up to four solutions.
 * distances = {|P1 P2|, |P0 P2|, |P0 P1|}, cosines of the angles between the viewing rays (1,2), (0,2), (0,1). */
static int solve_for_lengths(double lengths[4][3], const double distances[3], const double cosines[3])
{
    double p = cosines[0] * 2, q = cosines[1] * 2, r = cosines[2] * 2;
    double inv_d22 = 1. / (distances[2] * distances[2]);
    double a = inv_d22 * (distances[0] * distances[0]);
    double b = inv_d22 * (distances[1] * distances[1]);
    double a2 = a * a, b2 = b * b, p2 = p * p, q2 = q * q, r2 = r * r;
    double pr = p * r, pqr = q * pr;
    if (p2 + q2 + r2 - pqr - 1 == 0) return 0;                  /* the four points must not be coplanar with the centre */
    double ab = a * b, a_2 = 2 * a;
    double A = -2 * b + b2 + a2 + 1 + ab * (2 - r2) - a_2;
    if (A == 0) return 0;
    double a_4 = 4 * a;
    double B = q * (-2 * (ab + a2 + 1 - b) + r2 * ab + a_4) + pr * (b - b2 + ab);
    double C = q2 + b2 * (r2 + p2 - 2) - b * (p2 + pqr) - ab * (r2 + pqr) + (a2 - a_2) * (2 + q2) + 2;
    double D = pr * (ab - b2 + b) + q * ((p2 - 2) * b + 2 * (ab - a2) + a_4 - 2);
    double E = 1 + 2 * (b - a - ab) + b2 - b * p2 + a2;
    double temp = (p2 * (a - 1 + b) + r2 * (a - 1 - b) + pqr - a * pqr);
    double b0 = b * temp * temp;
    if (b0 == 0) return 0;
    double real_roots[4];
    int n = orc_solve_deg4(A, B, C, D, E, real_roots), i;
    if (n == 0) return 0;
    int nb_solutions = 0;
    double r3 = r2 * r, pr2 = p * r2, r3q = r3 * q;
    double inv_b0 = 1. / b0;
    for (i = 0; i < n; i++) {
        double x = real_roots[i];
        if (x <= 0) continue;
        double x2 = x * x;
        double b1 =
            ((1 - a - b) * x2 + (q * a - q) * x + 1 - a + b) *
            (((r3 * (a2 + ab * (2 - r2) - a_2 + b2 - 2 * b + 1)) * x +
              (r3q * (2 * (b - a2) + a_4 + ab * (r2 - 2) - 2) + pr2 * (1 + a2 + 2 * (ab - a - b) + r2 * (b - b2) + b2))) * x2 +
             (r3 * (q2 * (1 - 2 * a + a2) + r2 * (b2 - ab) - a_4 + 2 * (a2 - b2) + 2) + r * p2 * (b2 + 2 * (ab - b - a) + 1 + a2) +
              pr2 * q * (a_4 + 2 * (b - ab - a2) - 2 - r2 * b)) * x +
             2 * r3q * (a_2 - b - a2 + ab - 1) + pr2 * (q2 - a_4 + 2 * (a2 - b2) + r2 * b + q2 * (a2 - a_2) + 2) +
             p2 * (p * (2 * (ab - a - b) + a2 + b2 + 1) + 2 * q * r * (b + a_2 - a2 - ab - 1)));
        if (b1 <= 0) continue;
        double y = inv_b0 * b1;
        double v = x2 + y * y - x * y * r;
        if (v <= 0) continue;
        double Z = distances[2] / sqrt(v);
        double X = x * Z;
        double Y = y * Z;
        lengths[nb_solutions][0] = X;
        lengths[nb_solutions][1] = Y;
        lengths[nb_solutions][2] = Z;
        nb_solutions++;
    }
    return nb_solutions;
}

/* p3p::solve(R[4], t[4], three points): image points in pixels (mu, mv), object points Xw[9] */
static int p3p_solve3(double R[4][3][3], double t[4][3], const double mu[3], const double mv[3], const double Xw[9],
                      double fx, double fy, double cx, double cy)
{
    const double inv_fx = 1. / fx, inv_fy = 1. / fy, cx_fx = cx / fx, cy_fy = cy / fy;
    double u[3], v[3], k[3];
    int i, j;
    for (i = 0; i < 3; i++) {
        u[i] = inv_fx * mu[i] - cx_fx;
        v[i] = inv_fy * mv[i] - cy_fy;
        double norm = sqrt(u[i] * u[i] + v[i] * v[i] + 1);
        k[i] = 1. / norm;
        u[i] *= k[i];
        v[i] *= k[i];
    }
    const double X0 = Xw[0], Y0 = Xw[1], Z0 = Xw[2], X1 = Xw[3], Y1 = Xw[4], Z1 = Xw[5], X2 = Xw[6], Y2 = Xw[7], Z2 = Xw[8];
    double distances[3];
    distances[0] = sqrt((X1 - X2) * (X1 - X2) + (Y1 - Y2) * (Y1 - Y2) + (Z1 - Z2) * (Z1 - Z2));
    distances[1] = sqrt((X0 - X2) * (X0 - X2) + (Y0 - Y2) * (Y0 - Y2) + (Z0 - Z2) * (Z0 - Z2));
    distances[2] = sqrt((X0 - X1) * (X0 - X1) + (Y0 - Y1) * (Y0 - Y1) + (Z0 - Z1) * (Z0 - Z1));
    double cosines[3];
    cosines[0] = u[1] * u[2] + v[1] * v[2] + k[1] * k[2];
    cosines[1] = u[0] * u[2] + v[0] * v[2] + k[0] * k[2];
    cosines[2] = u[0] * u[1] + v[0] * v[1] + k[0] * k[1];
    double lengths[4][3];
    int n = solve_for_lengths(lengths, distances, cosines);
    int nb_solutions = 0;
    for (i = 0; i < n; i++) {
        double M_orig[3][3];
        for (j = 0; j < 3; j++) {
            M_orig[j][0] = lengths[i][j] * u[j];
            M_orig[j][1] = lengths[i][j] * v[j];
            M_orig[j][2] = lengths[i][j] * k[j];
        }
        if (!p3p_align(M_orig, Xw, R[nb_solutions], t[nb_solutions])) continue;
        nb_solutions++;
    }
    return nb_solutions;
}

/* p3p::solve with four points: pws = X0 Y0 Z0 .. X3 Y3 Z3, us = u0 v0 .. u3 v3 (pixels).  Returns 1 and R (row-major), t, or 0. */
int orc_p3p4(const double pws[12], const double us[8], double fx, double fy, double cx, double cy, double R[9], double t[3])
{
    double Rs[4][3][3], ts[4][3];
    const double mu[3] = {us[0], us[2], us[4]}, mv[3] = {us[1], us[3], us[5]};
    int n = p3p_solve3(Rs, ts, mu, mv, pws, fx, fy, cx, cy), i, j, ns = 0;
    if (n == 0) return 0;
    const double X3 = pws[9], Y3 = pws[10], Z3 = pws[11], mu3 = us[6], mv3 = us[7];
    double min_reproj = 0;
    for (i = 0; i < n; i++) {
        double X3p = Rs[i][0][0] * X3 + Rs[i][0][1] * Y3 + Rs[i][0][2] * Z3 + ts[i][0];
        double Y3p = Rs[i][1][0] * X3 + Rs[i][1][1] * Y3 + Rs[i][1][2] * Z3 + ts[i][1];
        double Z3p = Rs[i][2][0] * X3 + Rs[i][2][1] * Y3 + Rs[i][2][2] * Z3 + ts[i][2];
        double mu3p = cx + fx * X3p / Z3p;
        double mv3p = cy + fy * Y3p / Z3p;
        double reproj = (mu3p - mu3) * (mu3p - mu3) + (mv3p - mv3) * (mv3p - mv3);
        if (i == 0 || min_reproj > reproj) { ns = i; min_reproj = reproj; }
    }
    for (i = 0; i < 3; i++) {
        for (j = 0; j < 3; j++) R[3 * i + j] = Rs[ns][i][j];
        t[i] = ts[ns][i];
    }
    return 1;
}

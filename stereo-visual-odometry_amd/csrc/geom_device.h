// geom_device.h -- double-precision device routines of the pose solver: one-sided Jacobi SVD and
// the solves built on it, EPnP, Rodrigues, the pinhole projection with its Jacobian.
//
// These restate the OpenCV 3.4 algorithms behind cv::triangulatePoints / cv::solvePnPRansac /
// cv::Rodrigues (reference call sites src/tracking.cpp:292-294, 485-488) for per-thread execution
// on gfx950.  Every floating-point expression keeps the association order of the published
// algorithms (and of oracle/geom.c, oracle/pnp.c), the library is built with -ffp-contract=off,
// and f64 +,-,*,/,sqrt are IEEE-exact on CDNA4, so hypotheses, inlier masks and triangulated
// points are bit-identical to the CPU oracle; only sin/cos/acos/log/pow/atan2 (device libm) may
// differ in the last ulp, which the Levenberg-Marquardt refit absorbs.
//
// Matrices are addressed through (pointer, element stride) so the same code runs on private
// arrays (stride 1) and on the lane-interleaved LDS image used for the 12x12 EPnP eigenproblem
// (stride 64: element e of lane t at base[e*64 + t], bank-conflict-free ds_read_b64).
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

namespace svo {

#define SVO_DBL_EPS 2.220446049250313e-16
#define SVO_DBL_MIN 2.2250738585072014e-308

// The rotation of one Jacobi pair.  JacobiSVDImpl_ branches on the sign of beta:
//   beta < 0:  s = sqrt(((gamma - beta) * 0.5) / gamma),  c = p / (gamma * s * 2)
//   else:      c = sqrt((gamma + beta) / (gamma * 2)),    s = p / (gamma * c * 2)
// Both arms are one division, one square root and one more division on different operands; selecting
// the operands first runs ONE div-sqrt-div sequence for all lanes (64 lanes = 64 different matrices
// practically always disagree on the sign, and then the branch form executes both arms: 2 sqrt + 4
// div of ~15 instructions each).  Same operations on the same values per lane, so the same bits.
__device__ __forceinline__ void jacobi_cs_d(double p, double beta, double gamma, double &c, double &s)
{
    const bool neg = beta < 0;
    const double num = neg ? (gamma - beta) * 0.5 : gamma + beta;
    const double den = neg ? gamma : gamma * 2;
    const double r = sqrt(num / den);
    const double q = p / (gamma * r * 2);
    s = neg ? r : q;
    c = neg ? q : r;
}

// JacobiSVDImpl_<double>: At is n rows of length m, element (i,k) at At[(i*m+k)*as].
// Vt (n x n, element stride vs) may be null; `sort_rows` makes the rows of At follow the
// descending sort and get normalised even without Vt (what cv::SVD does when U is requested).
// M, N are compile-time so the k-loops unroll: their LDS / scratch loads are then issued together
// instead of one dependent load per multiply (the run-time-bound version was latency-bound).
// -- the three parts of JacobiSVDImpl_: column norms (+ V = I), the rotation of one pair, the epilogue
template <int M, int N>
__device__ inline void jacobi_init_d(double *At, int as, double *W, int ws, double *Vt, int vs)
{
    constexpr int m = M, n = N;
    for (int i = 0; i < n; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < m; k++) { double t = At[(i * m + k) * as]; sd += t * t; }
        W[i * ws] = sd;
        if (Vt) {
            for (int k = 0; k < n; k++) Vt[(i * n + k) * vs] = 0;
            Vt[(i * n + i) * vs] = 1;
        }
    }
}

// rows i, j of At (and of Vt); returns whether the pair was rotated
template <int M, int N>
__device__ __forceinline__ bool jacobi_pair_d(double *At, int as, double *W, int ws, double *Vt, int vs, int i, int j)
{
    constexpr int m = M, n = N;
    const double eps = SVO_DBL_EPS * 10;
    double *Ai = At + (i * m) * as, *Aj = At + (j * m) * as;
    double a = W[i * ws], p = 0, b = W[j * ws], c, s;
    {
        double xi[M], xj[M];
#pragma unroll
        for (int k = 0; k < m; k++) { xi[k] = Ai[k * as]; xj[k] = Aj[k * as]; }
#pragma unroll
        for (int k = 0; k < m; k++) p += xi[k] * xj[k];
    }
    if (fabs(p) <= eps * sqrt(a * b)) return false;
    p *= 2;
    double beta = a - b, gamma = sqrt(p * p + beta * beta);
    jacobi_cs_d(p, beta, gamma, c, s);
    a = b = 0;
#pragma unroll
    for (int k = 0; k < m; k++) {
        double x = Ai[k * as], y = Aj[k * as];
        double t0 = c * x + s * y;
        double t1 = -s * x + c * y;
        Ai[k * as] = t0; Aj[k * as] = t1;
        a += t0 * t0; b += t1 * t1;
    }
    W[i * ws] = a; W[j * ws] = b;
    if (Vt) {
        double *Vi = Vt + (i * n) * vs, *Vj = Vt + (j * n) * vs;
#pragma unroll
        for (int k = 0; k < n; k++) {
            double x = Vi[k * vs], y = Vj[k * vs];
            double t0 = c * x + s * y;
            double t1 = -s * x + c * y;
            Vi[k * vs] = t0; Vj[k * vs] = t1;
        }
    }
    return true;
}

template <int M, int N>
__device__ inline void jacobi_finish_d(double *At, int as, double *W, double *Vt, int vs, bool sort_rows)
{
    constexpr int m = M, n = N;
    const double minval = SVO_DBL_MIN;
    for (int i = 0; i < n; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < m; k++) { double t = At[(i * m + k) * as]; sd += t * t; }
        W[i] = sqrt(sd);
    }
    for (int i = 0; i < n - 1; i++) {
        int j = i;
        for (int k = i + 1; k < n; k++) if (W[j] < W[k]) j = k;
        if (i != j) {
            double t = W[i]; W[i] = W[j]; W[j] = t;
            if (Vt || sort_rows)
                for (int k = 0; k < m; k++) {
                    t = At[(i * m + k) * as]; At[(i * m + k) * as] = At[(j * m + k) * as]; At[(j * m + k) * as] = t;
                }
            if (Vt)
                for (int k = 0; k < n; k++) {
                    t = Vt[(i * n + k) * vs]; Vt[(i * n + k) * vs] = Vt[(j * n + k) * vs]; Vt[(j * n + k) * vs] = t;
                }
        }
    }
    if (!Vt && !sort_rows) return;
    for (int i = 0; i < n; i++) {
        double sd = W[i];
        double s = sd > minval ? 1 / sd : 0.;
        for (int k = 0; k < m; k++) At[(i * m + k) * as] *= s;
    }
}

template <int M, int N>
__device__ inline void jacobi_svd_d(double *At, int as, double *W, double *Vt, int vs, bool sort_rows)
{
    constexpr int m = M, n = N;
    const int max_iter = m > 30 ? m : 30;
    jacobi_init_d<M, N>(At, as, W, 1, Vt, vs);
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int i = 0; i < n - 1; i++)
            for (int j = i + 1; j < n; j++)
                if (jacobi_pair_d<M, N>(At, as, W, 1, Vt, vs, i, j)) changed = true;
        if (!changed) break;
    }
    jacobi_finish_d<M, N>(At, as, W, Vt, vs, sort_rows);
}

// NP disjoint pairs (i_q, j_q) of At in ONE instruction stream (no Vt).  A lone wave issues a dependent f64
// chain at one instruction per ~8 cycles, and a rotation is mostly such chains (the dot product, then
// sqrt - div - sqrt - div of ~15 dependent instructions each): two independent pairs interleave into the
// gaps, so a second pair costs a fraction of the first.  The arithmetic of a pair is exactly
// jacobi_pair_d's; lanes whose pair needs no rotation compute on and store nothing; the whole call is
// skipped when no lane of the wave rotates anything.
template <int M, int N, int NP>
__device__ __forceinline__ bool jacobi_pairs_d(double *At, int as, double *W, int ws, const int (&pi)[NP], const int (&pj)[NP])
{
    constexpr int m = M;
    const double eps = SVO_DBL_EPS * 10;
    double x[NP][M], y[NP][M], a[NP], b[NP], p[NP];
    bool rot[NP], any = false;
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const double *Ai = At + (pi[q] * m) * as, *Aj = At + (pj[q] * m) * as;
#pragma unroll
        for (int k = 0; k < m; k++) { x[q][k] = Ai[k * as]; y[q][k] = Aj[k * as]; }
        a[q] = W[pi[q] * ws]; b[q] = W[pj[q] * ws];
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        double pp = 0;
#pragma unroll
        for (int k = 0; k < m; k++) pp += x[q][k] * y[q][k];
        p[q] = pp;
        rot[q] = !(fabs(pp) <= eps * sqrt(a[q] * b[q]));
        any = any || rot[q];
    }
    if (!__any(any)) return false;
    // (starting gamma's square root beside the convergence test's, ahead of this branch, measured 2 % slower)
    double c[NP], sn[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const double p2 = p[q] * 2;
        const double beta = a[q] - b[q], gamma = sqrt(p2 * p2 + beta * beta);
        jacobi_cs_d(p2, beta, gamma, c[q], sn[q]);
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        double na = 0, nb = 0;
#pragma unroll
        for (int k = 0; k < m; k++) {
            const double t0 = c[q] * x[q][k] + sn[q] * y[q][k];
            const double t1 = -sn[q] * x[q][k] + c[q] * y[q][k];
            x[q][k] = t0; y[q][k] = t1;
            na += t0 * t0; nb += t1 * t1;
        }
        a[q] = na; b[q] = nb;
    }
#pragma unroll
    for (int q = 0; q < NP; q++) {
        if (rot[q]) {
            double *Ai = At + (pi[q] * m) * as, *Aj = At + (pj[q] * m) * as;
#pragma unroll
            for (int k = 0; k < m; k++) { Ai[k * as] = x[q][k]; Aj[k * as] = y[q][k]; }
            W[pi[q] * ws] = a[q]; W[pj[q] * ws] = b[q];
        }
    }
    return any;
}

// The sweeps of jacobi_svd_d by the NW waves of a workgroup: lane = matrix (the lane-interleaved LDS
// image; W in LDS too), wave = one of the pairs that can be rotated at the same time.  A rotation touches rows i, j and W[i], W[j] only, so rotations of disjoint pairs commute
// exactly; the cyclic-by-rows order (0,1),(0,2),...,(10,11) is therefore equivalent -- bit for bit --
// to any order that keeps the relative order of pairs SHARING a row.  Stage s = i + j does: two pairs
// that share a row are ordered by their sum in the cyclic order (same i: by j; same j: by i; (a,b)
// before (b,d): a < d; (c,a) before (a,b): c < b), and the pairs of one stage are disjoint.  For 12
// columns 21 stages of up to 6 pairs replace 66 sequential pairs (6 columns: 9 stages of up to 3
// replace 15); one workgroup barrier per stage.  21 is also the depth of the dependency graph (the
// chain (0,1) ... (0,11), (1,11) ... (10,11)), so no schedule has fewer stages.  A stage with more
// pairs than waves (five stages of the 12-column problem with four waves) gives its first waves TWO
// pairs each, rotated in one interleaved instruction stream (jacobi_pairs_d) instead of one after the
// other.  Every wave of the workgroup must call this.  A matrix that has
// converged keeps being swept while others have not (a sweep without rotation changes nothing).
template <int M, int N, int NW>
__device__ inline void jacobi_sweeps_coop(double *At, int as, double *W, int ws, double *Vt, int vs, int wave)
{
    constexpr int max_iter = M > 30 ? M : 30;
    constexpr bool kPairs = N / 2 <= 2 * NW;                 // a stage never has more than two pairs per wave
#ifdef SVO_PNP_DIAG
    long long t_pair = 0, t_bar = 0; int n_sweeps = 0;
#endif
    for (int iter = 0; iter < max_iter; iter++) {
        bool changed = false;
        for (int s = 1; s <= 2 * N - 3; s++) {
            const int i_lo = s > N - 1 ? s - (N - 1) : 0, cnt = (s - 1) / 2 - i_lo + 1;
#ifdef SVO_PNP_DIAG
            const long long t0 = clock64();
#endif
            if (kPairs && Vt == nullptr) {
                if (wave + NW < cnt) {                                  // this wave has two pairs in the stage
                    const int i0 = i_lo + wave, i1 = i0 + NW;
                    const int pi[2] = {i0, i1}, pj[2] = {s - i0, s - i1};
                    if (jacobi_pairs_d<M, N, 2>(At, as, W, ws, pi, pj)) changed = true;
                } else if (wave < cnt) {
                    const int pi[1] = {i_lo + wave}, pj[1] = {s - i_lo - wave};
                    if (jacobi_pairs_d<M, N, 1>(At, as, W, ws, pi, pj)) changed = true;
                }
            } else {
                for (int q = wave; q < cnt; q += NW) {
                    const int i = i_lo + q;
                    if (jacobi_pair_d<M, N>(At, as, W, ws, Vt, vs, i, s - i)) changed = true;
                }
            }
#ifdef SVO_PNP_DIAG
            const long long t1 = clock64();
#endif
            __syncthreads();
#ifdef SVO_PNP_DIAG
            t_pair += t1 - t0; t_bar += clock64() - t1;
#endif
        }
#ifdef SVO_PNP_DIAG
        n_sweeps++;
#endif
        if (!__syncthreads_or(changed)) break;
    }
#ifdef SVO_PNP_DIAG
    if (N == 12 && blockIdx.x == 0 && blockIdx.y == 0 && (threadIdx.x & 63) == 0)
        printf("pnp_hyp diag: wave %d sweeps %d  in pairs %lld  at barriers %lld cycles\n", wave, n_sweeps, t_pair, t_bar);
#endif
}

// The right-singular vector of the SMALLEST singular value of a 4x4 matrix (cv::SVD::compute +
// "last row of V^T", what cv::triangulatePoints needs) with everything in registers.  Same rotation
// sequence, same expressions and the same selection sort as jacobi_svd_d<4, 4> -- so the same bits
// -- but the pair loops are fully unrolled and the sort moves (value, row index) pairs instead of
// rows, so no array is ever indexed with a run-time value: the generic routine's row swaps forced
// At / Vt into scratch memory, and every rotation then paid a memory round trip (0.37 ms per 256 ORB
// pairs for a few hundred points each).  At: the transposed input (row i = column i of A).
__device__ inline void jacobi_null4_d(double (&At)[16], double (&X)[4])
{
    const double eps = SVO_DBL_EPS * 10;
    double W[4], Vt[16];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[i * 4 + k] * At[i * 4 + k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < 4; k++) Vt[i * 4 + k] = (i == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = i + 1; j < 4; j++) {
                double a = W[i], p = 0, b = W[j], c, s;
#pragma unroll
                for (int k = 0; k < 4; k++) p += At[i * 4 + k] * At[j * 4 + k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                double beta = a - b, gamma = sqrt(p * p + beta * beta);
                jacobi_cs_d(p, beta, gamma, c, s);
                a = b = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double x = At[i * 4 + k], y = At[j * 4 + k];
                    const double t0 = c * x + s * y, t1 = -s * x + c * y;
                    At[i * 4 + k] = t0; At[j * 4 + k] = t1;
                    a += t0 * t0; b += t1 * t1;
                }
                W[i] = a; W[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const double x = Vt[i * 4 + k], y = Vt[j * 4 + k];
                    Vt[i * 4 + k] = c * x + s * y; Vt[j * 4 + k] = -s * x + c * y;
                }
            }
        if (!changed) break;
    }
    int row[4] = {0, 1, 2, 3};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sd += At[i * 4 + k] * At[i * 4 + k];
        W[i] = sqrt(sd);
    }
    // selection sort, descending ("j = i; for k > i: if (W[j] < W[k]) j = k; swap(i, j)") on (W, row)
#pragma unroll
    for (int i = 0; i < 3; i++) {
        double best = W[i];
        int bj = i;
#pragma unroll
        for (int k = i + 1; k < 4; k++) if (best < W[k]) { best = W[k]; bj = k; }
#pragma unroll
        for (int k = i + 1; k < 4; k++)
            if (bj == k) {
                const double tw = W[i]; W[i] = W[k]; W[k] = tw;
                const int tr = row[i]; row[i] = row[k]; row[k] = tr;
            }
    }
    const int r = row[3];
#pragma unroll
    for (int k = 0; k < 4; k++) X[k] = r == 0 ? Vt[k] : r == 1 ? Vt[4 + k] : r == 2 ? Vt[8 + k] : Vt[12 + k];
}

// cv::solve(A, b, x, DECOMP_SVD), A m x n row-major (m >= n, n <= 6, m <= 6), one right-hand side.
// ws: workspace of >= 72 elements with element stride st (the lane-interleaved LDS image when the
// caller has one: private arrays with run-time indexing live in scratch memory, which is what
// made the first version of the pose kernel latency-bound).
template <int M, int N>
__device__ inline void svd_solve_d(const double *A, const double *b, double *x, double *ws, int st)
{
    constexpr int m = M, n = N;
    double W[6];
    double *At = ws, *Vt = ws + (M * N) * st;           // workspace: M*N + N*N doubles
    for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) At[(j * m + i) * st] = A[i * n + j];
    jacobi_svd_d<M, N>(At, st, W, Vt, st, false);
    double threshold = 0;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= SVO_DBL_EPS * 2;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += At[(i * m + j) * st] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt[(i * n + j) * st];
    }
}

// svd_solve_d by the NW waves of a workgroup (all of them call it, with identical arguments in the
// lanes that share a matrix copy); Wl: n doubles of LDS per lane copy.  x is valid on wave 0.
template <int M, int N, int NW>
__device__ inline void svd_solve_coop_d(const double *A, const double *b, double *x, double *ws, int st, double *Wl, int wave)
{
    constexpr int m = M, n = N;
    double *At = ws, *Vt = ws + (M * N) * st;           // workspace: M*N + N*N doubles
    if (wave == 0) {
        for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) At[(j * m + i) * st] = A[i * n + j];
        jacobi_init_d<M, N>(At, st, Wl, st, Vt, st);
    }
    __syncthreads();
    jacobi_sweeps_coop<M, N, NW>(At, st, Wl, st, Vt, st, wave);
    if (wave) return;
    double W[6];
    jacobi_finish_d<M, N>(At, st, W, Vt, st, false);
    double threshold = 0;
    for (int i = 0; i < n; i++) x[i] = 0;
    for (int i = 0; i < n; i++) threshold += W[i];
    threshold *= SVO_DBL_EPS * 2;
    for (int i = 0; i < n; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        double s = 0;
        for (int j = 0; j < m; j++) s += At[(i * m + j) * st] * b[j];
        s *= wi;
        for (int j = 0; j < n; j++) x[j] = x[j] + s * Vt[(i * n + j) * st];
    }
}

// A x = b for a symmetric positive definite 6x6 A by Cholesky, everything in registers.  Returns false
// -- and leaves x undefined -- when a pivot is not safely positive (relative to its diagonal entry):
// the caller then takes the SVD route, which is what cv::solve(DECOMP_SVD) would do for a
// rank-deficient system.
__device__ inline bool chol_solve6_d(const double (&A)[36], const double (&b)[6], double (&x)[6])
{
    double L[21];                                   // row-major lower triangle
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 6; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) {
            double s = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; k++) s -= L[i * (i + 1) / 2 + k] * L[j * (j + 1) / 2 + k];
            if (i == j) {
                ok = ok && s > 1e-10 * A[i * 6 + i];
                L[i * (i + 1) / 2 + i] = sqrt(s);
            } else {
                L[i * (i + 1) / 2 + j] = s / L[j * (j + 1) / 2 + j];
            }
        }
    }
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; k++) s -= L[i * (i + 1) / 2 + k] * y[k];
        y[i] = s / L[i * (i + 1) / 2 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s -= L[k * (k + 1) / 2 + i] * x[k];
        x[i] = s / L[i * (i + 1) / 2 + i];
    }
    return ok;
}

// cv::invert(A, Ainv, DECOMP_SVD) for 3x3
__device__ inline void svd_invert3_d(const double *A, double *Ainv, double *ws, int st)
{
    double W[3], buffer[3], threshold = 0;
    double *At = ws, *Vt = ws + 9 * st;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[(j * 3 + i) * st] = A[i * 3 + j];
    jacobi_svd_d<3, 3>(At, st, W, Vt, st, false);
    for (int i = 0; i < 9; i++) Ainv[i] = 0;
    for (int i = 0; i < 3; i++) threshold += W[i];
    threshold *= SVO_DBL_EPS * 2;
    for (int i = 0; i < 3; i++) {
        double wi = W[i];
        if (fabs(wi) <= threshold) continue;
        wi = 1 / wi;
        for (int j = 0; j < 3; j++) buffer[j] = At[(i * 3 + j) * st] * wi;
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) Ainv[j * 3 + k] = Ainv[j * 3 + k] + Vt[(i * 3 + j) * st] * buffer[k];
    }
}

__device__ inline double dot3_d(const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ inline double dist2_d(const double *a, const double *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

// epnp::qr_solve on the 6x4 Gauss-Newton system (column-max scan quirk of the published code kept)
__device__ inline void epnp_qr_solve_d(double *A, double *b, double *X)
{
    const int nr = 6, nc = 4;
    double A1[4], A2[4];
    for (int k = 0; k < nc; k++) {
        double eta = fabs(A[k * nc + k]);
        for (int i = k + 1; i < nr; i++) {
            double elt = fabs(A[(i - 1) * nc + k]);
            if (eta < elt) eta = elt;
        }
        if (eta == 0) return;
        double sum2 = 0.0, inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) {
            A[i * nc + k] *= inv_eta;
            sum2 += A[i * nc + k] * A[i * nc + k];
        }
        double sigma = sqrt(sum2);
        if (A[k * nc + k] < 0) sigma = -sigma;
        A[k * nc + k] += sigma;
        A1[k] = sigma * A[k * nc + k];
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
            double sum = 0;
            for (int i = k; i < nr; i++) sum += A[i * nc + k] * A[i * nc + j];
            double tau = sum / A1[k];
            for (int i = k; i < nr; i++) A[i * nc + j] -= tau * A[i * nc + k];
        }
    }
    for (int j = 0; j < nc; j++) {
        double tau = 0;
        for (int i = j; i < nr; i++) tau += A[i * nc + j] * b[i];
        tau /= A1[j];
        for (int i = j; i < nr; i++) b[i] -= tau * A[i * nc + j];
    }
    X[nc - 1] = b[nc - 1] / A2[nc - 1];
    for (int i = nc - 2; i >= 0; i--) {
        double sum = 0;
        for (int j = i + 1; j < nc; j++) sum += A[i * nc + j] * X[j];
        X[i] = (b[i] - sum) / A2[i];
    }
}

struct Epnp5 {
    double pws[15], us[10], alphas[20], pcs[15];
    double cws[4][3], ccs[4][3];
    double fu, fv, uc, vc;
};

// compute_ccs + compute_pcs + solve_for_sign + estimate_R_and_t + reprojection_error
__device__ inline double epnp_R_and_t_d(Epnp5 &e, const double *v /* 4 x 12: ut rows 11,10,9,8 */,
                                        const double *betas, double R[9], double t[3], double *ws, int st)
{
    const int n = 5;
    for (int i = 0; i < 4; i++) e.ccs[i][0] = e.ccs[i][1] = e.ccs[i][2] = 0.0;
    for (int i = 0; i < 4; i++) {
        const double *vi = v + 12 * i;
        for (int j = 0; j < 4; j++) for (int k = 0; k < 3; k++) e.ccs[j][k] += betas[i] * vi[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
        const double *a = e.alphas + 4 * i;
        for (int j = 0; j < 3; j++)
            e.pcs[3 * i + j] = a[0] * e.ccs[0][j] + a[1] * e.ccs[1][j] + a[2] * e.ccs[2][j] + a[3] * e.ccs[3][j];
    }
    if (e.pcs[2] < 0.0) {
        for (int i = 0; i < 4; i++) for (int j = 0; j < 3; j++) e.ccs[i][j] = -e.ccs[i][j];
        for (int i = 0; i < 3 * n; i++) e.pcs[i] = -e.pcs[i];
    }
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) { pc0[j] += e.pcs[3 * i + j]; pw0[j] += e.pws[3 * i + j]; }
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, abt_d[3];
    double *abt_ut = ws, *abt_vt = ws + 9 * st;
    for (int i = 0; i < n; i++) {
        const double *pc = e.pcs + 3 * i, *pw = e.pws + 3 * i;
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) abt_ut[(j * 3 + i) * st] = abt[i * 3 + j];
    jacobi_svd_d<3, 3>(abt_ut, st, abt_d, abt_vt, st, false);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = abt_ut[(0 * 3 + i) * st] * abt_vt[(0 * 3 + j) * st] + abt_ut[(1 * 3 + i) * st] * abt_vt[(1 * 3 + j) * st] +
                           abt_ut[(2 * 3 + i) * st] * abt_vt[(2 * 3 + j) * st];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] -
                       R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - dot3_d(R, pw0);
    t[1] = pc0[1] - dot3_d(R + 3, pw0);
    t[2] = pc0[2] - dot3_d(R + 6, pw0);
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
        const double *pw = e.pws + 3 * i;
        double Xc = dot3_d(R, pw) + t[0], Yc = dot3_d(R + 3, pw) + t[1];
        double inv_Zc = 1.0 / (dot3_d(R + 6, pw) + t[2]);
        double ue = e.uc + e.fu * Xc * inv_Zc, ve = e.vc + e.fv * Yc * inv_Zc;
        double u = e.us[2 * i], vv = e.us[2 * i + 1];
        sum2 += sqrt((u - ue) * (u - ue) + (vv - ve) * (vv - ve));
    }
    return sum2 / n;
}

// epnp::compute_pose for the 5-point minimal sample.  `big` is this lane's 144-double matrix
// region (element stride `bs`), used for the 12x12 eigenproblem of M^T M.
// -- part 1: control points, barycentric coordinates, M^T M into `big`
__device__ inline void epnp5_front_d(Epnp5 &e, double *big, int bs)
{
    const int n = 5;
    // ---- choose_control_points
    e.cws[0][0] = e.cws[0][1] = e.cws[0][2] = 0;
    for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) e.cws[0][j] += e.pws[3 * i + j];
    for (int j = 0; j < 3; j++) e.cws[0][j] /= n;
    {
        double PW0[15], c3[9], dc[3];
        double *uct = big, *vt3 = big + 9 * bs;
        for (int i = 0; i < n; i++) for (int j = 0; j < 3; j++) PW0[3 * i + j] = e.pws[3 * i + j] - e.cws[0][j];
        for (int i = 0; i < 3; i++)
            for (int j = i; j < 3; j++) {
                double s = 0;
                for (int k = 0; k < n; k++) s += PW0[k * 3 + i] * PW0[k * 3 + j];
                c3[i * 3 + j] = s;
            }
        for (int i = 0; i < 3; i++) for (int j = 0; j < i; j++) c3[i * 3 + j] = c3[j * 3 + i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) uct[(j * 3 + i) * bs] = c3[i * 3 + j];
        jacobi_svd_d<3, 3>(uct, bs, dc, vt3, bs, false);
        for (int i = 1; i < 4; i++) {
            double k = sqrt(dc[i - 1] / n);
            for (int j = 0; j < 3; j++) e.cws[i][j] = e.cws[0][j] + k * uct[(3 * (i - 1) + j) * bs];
        }
    }
    // ---- compute_barycentric_coordinates
    {
        double cc[9], ci[9];
        for (int i = 0; i < 3; i++) for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = e.cws[j][i] - e.cws[0][i];
        svd_invert3_d(cc, ci, big, bs);
        for (int i = 0; i < n; i++) {
            const double *pi = e.pws + 3 * i;
            double *a = e.alphas + 4 * i;
            for (int j = 0; j < 3; j++)
                a[1 + j] = ci[3 * j] * (pi[0] - e.cws[0][0]) + ci[3 * j + 1] * (pi[1] - e.cws[0][1]) +
                           ci[3 * j + 2] * (pi[2] - e.cws[0][2]);
            a[0] = 1.0 - a[1] - a[2] - a[3];
        }
    }
    // ---- M^T M accumulated row by row (same per-entry summation order as cvMulTransposed),
    //      directly into the transposed (== itself, symmetric) Jacobi work matrix
    for (int i = 0; i < 144; i++) big[i * bs] = 0;
    for (int r = 0; r < 2 * n; r++) {
        const int p = r >> 1;
        const double *as = e.alphas + 4 * p;
        double row[12];
        for (int j = 0; j < 4; j++) {
            if ((r & 1) == 0) { row[3 * j] = as[j] * e.fu; row[3 * j + 1] = 0.0; row[3 * j + 2] = as[j] * (e.uc - e.us[2 * p]); }
            else { row[3 * j] = 0.0; row[3 * j + 1] = as[j] * e.fv; row[3 * j + 2] = as[j] * (e.vc - e.us[2 * p + 1]); }
        }
        for (int i = 0; i < 12; i++)
            for (int j = i; j < 12; j++) big[(i * 12 + j) * bs] += row[i] * row[j];
    }
    for (int i = 0; i < 12; i++) for (int j = 0; j < i; j++) big[(i * 12 + j) * bs] = big[(j * 12 + i) * bs];
}

// -- part 2 (after the SVD of `big`: rows = left singular vectors, sorted): the four null-space vectors
__device__ inline void epnp5_load_v_d(const double *big, int bs, double (&v)[48])   // ut rows 11, 10, 9, 8
{
    for (int i = 0; i < 4; i++) for (int k = 0; k < 12; k++) v[i * 12 + k] = big[((11 - i) * 12 + k) * bs];
}

// -- part 3: betas, R, t.  NSEL = 0: all of epnp's three beta approximations, the best one wins
// (compute_pose's rule); 1..3: only that approximation -- its pose and its reprojection error, for a
// caller that runs the three side by side and applies the rule itself (pnp_hyp_kernel).
// `big` is workspace only here (64 doubles per lane).
template <int NSEL = 0>
__device__ inline double epnp5_back_d(Epnp5 &e, const double (&v)[48], double *big, int bs, double Rout[9], double tout[3])
{

    // ---- compute_L_6x10, compute_rho
    double L[60], rho[6];
    {
        double dv[4][6][3];
        for (int i = 0; i < 4; i++) {
            int a = 0, b = 1;
            for (int j = 0; j < 6; j++) {
                dv[i][j][0] = v[i * 12 + 3 * a] - v[i * 12 + 3 * b];
                dv[i][j][1] = v[i * 12 + 3 * a + 1] - v[i * 12 + 3 * b + 1];
                dv[i][j][2] = v[i * 12 + 3 * a + 2] - v[i * 12 + 3 * b + 2];
                b++;
                if (b > 3) { a++; b = a + 1; }
            }
        }
        for (int i = 0; i < 6; i++) {
            double *row = L + 10 * i;
            row[0] = dot3_d(dv[0][i], dv[0][i]);
            row[1] = 2.0 * dot3_d(dv[0][i], dv[1][i]);
            row[2] = dot3_d(dv[1][i], dv[1][i]);
            row[3] = 2.0 * dot3_d(dv[0][i], dv[2][i]);
            row[4] = 2.0 * dot3_d(dv[1][i], dv[2][i]);
            row[5] = dot3_d(dv[2][i], dv[2][i]);
            row[6] = 2.0 * dot3_d(dv[0][i], dv[3][i]);
            row[7] = 2.0 * dot3_d(dv[1][i], dv[3][i]);
            row[8] = 2.0 * dot3_d(dv[2][i], dv[3][i]);
            row[9] = dot3_d(dv[3][i], dv[3][i]);
        }
    }
    rho[0] = dist2_d(e.cws[0], e.cws[1]); rho[1] = dist2_d(e.cws[0], e.cws[2]);
    rho[2] = dist2_d(e.cws[0], e.cws[3]); rho[3] = dist2_d(e.cws[1], e.cws[2]);
    rho[4] = dist2_d(e.cws[1], e.cws[3]); rho[5] = dist2_d(e.cws[2], e.cws[3]);

    double best_rep = 0;
    for (int N = 1; N <= 3; N++) {
        if (NSEL && N != NSEL) continue;
        // ---- find_betas_approx_N
        double betas[4], Lr[30], bb[5];
        const int nc = N == 1 ? 4 : (N == 2 ? 3 : 5);
        for (int i = 0; i < 6; i++)
            for (int j = 0; j < nc; j++) {
                int col = N == 1 ? (j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 3 : 6) : j;
                Lr[i * nc + j] = L[10 * i + col];
            }
        // the 12x12 image is free again (v[] holds what is needed)
        if (N == 1) svd_solve_d<6, 4>(Lr, rho, bb, big, bs);
        else if (N == 2) svd_solve_d<6, 3>(Lr, rho, bb, big, bs);
        else svd_solve_d<6, 5>(Lr, rho, bb, big, bs);
        if (N == 1) {
            if (bb[0] < 0) {
                betas[0] = sqrt(-bb[0]);
                betas[1] = -bb[1] / betas[0]; betas[2] = -bb[2] / betas[0]; betas[3] = -bb[3] / betas[0];
            } else {
                betas[0] = sqrt(bb[0]);
                betas[1] = bb[1] / betas[0]; betas[2] = bb[2] / betas[0]; betas[3] = bb[3] / betas[0];
            }
        } else {
            if (bb[0] < 0) {
                betas[0] = sqrt(-bb[0]);
                betas[1] = (bb[2] < 0) ? sqrt(-bb[2]) : 0.0;
            } else {
                betas[0] = sqrt(bb[0]);
                betas[1] = (bb[2] > 0) ? sqrt(bb[2]) : 0.0;
            }
            if (bb[1] < 0) betas[0] = -betas[0];
            betas[2] = N == 3 ? bb[3] / betas[0] : 0.0;
            betas[3] = 0.0;
        }
        // ---- gauss_newton: 5 iterations
        for (int it = 0; it < 5; it++) {
            double A[24], b[6], x[4] = {0, 0, 0, 0};
            for (int i = 0; i < 6; i++) {
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
            epnp_qr_solve_d(A, b, x);
            for (int i = 0; i < 4; i++) betas[i] += x[i];
        }
        double Rn[9], tn[3];
        double rep = epnp_R_and_t_d(e, v, betas, Rn, tn, big, bs);
        // "N = 1; if (rep[2] < rep[1]) N = 2; if (rep[3] < rep[N]) N = 3;"
        if (N == 1 || NSEL || rep < best_rep) {
            best_rep = rep;
            for (int i = 0; i < 9; i++) Rout[i] = Rn[i];
            for (int i = 0; i < 3; i++) tout[i] = tn[i];
        }
    }
    return best_rep;
}

__device__ inline void epnp5_d(Epnp5 &e, double *big, int bs, double Rout[9], double tout[3])
{
    epnp5_front_d(e, big, bs);
    double d12[12];
    jacobi_svd_d<12, 12>(big, bs, d12, nullptr, 0, true);
    double v[48];
    epnp5_load_v_d(big, bs, v);
    epnp5_back_d<0>(e, v, big, bs, Rout, tout);
}

// ---- the same back part for a caller that keeps the hypothesis' state in MEMORY between the phases ----
// (pnp_hyp_kernel: the front runs on one wave, the 12x12 SVD on four, the three beta approximations on
// three; carried in registers across all of that, pws / us / alphas / cws / v / L were 300+ live f64
// values per lane and most of the kernel ran out of scratch memory).  `hand` is the hand-over record of
// the lane's hypothesis, element e at hand[e * hs]:
//   0..14 pws   15..24 us   25..44 alphas   45..56 cws   57..104 v (ut rows 11, 10, 9, 8)
// Every value is loaded where it is used and dies there; expressions and summation orders are those of
// epnp5_back_d / epnp_R_and_t_d above, so the bits are the same.
constexpr int kEpnpHandPws = 0, kEpnpHandUs = 15, kEpnpHandAlphas = 25, kEpnpHandCws = 45, kEpnpHandV = 57, kEpnpHandDoubles = 105;

// rows [row0, row0 + nrows) of compute_L_6x10 from the sorted 12x12 image (rows 11..8 = v) into Lm (element e at
// Lm[e * ls]); the three beta approximations share one L, so the rows are computed once, by different waves
template <int ROW0, int NROWS>
__device__ inline void epnp5_L_rows_d(const double *big, int bs, double *Lm, int ls)
{
#pragma unroll
    for (int i = ROW0; i < ROW0 + NROWS; i++) {
        const int a = i < 3 ? 0 : i < 5 ? 1 : 2, b = i < 3 ? i + 1 : i < 5 ? i - 1 : 3;
        double dv[4][3];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) dv[q][c] = big[((11 - q) * 12 + 3 * a + c) * bs] - big[((11 - q) * 12 + 3 * b + c) * bs];
        double *row = Lm + 10 * i * ls;
        row[0 * ls] = dot3_d(dv[0], dv[0]);
        row[1 * ls] = 2.0 * dot3_d(dv[0], dv[1]);
        row[2 * ls] = dot3_d(dv[1], dv[1]);
        row[3 * ls] = 2.0 * dot3_d(dv[0], dv[2]);
        row[4 * ls] = 2.0 * dot3_d(dv[1], dv[2]);
        row[5 * ls] = dot3_d(dv[2], dv[2]);
        row[6 * ls] = 2.0 * dot3_d(dv[0], dv[3]);
        row[7 * ls] = 2.0 * dot3_d(dv[1], dv[3]);
        row[8 * ls] = 2.0 * dot3_d(dv[2], dv[3]);
        row[9 * ls] = dot3_d(dv[3], dv[3]);
    }
}
// the same rows from the HAND-OVER record (v = ut rows 11, 10, 9, 8 at kEpnpHandV): the back kernel of the split
// hypothesis launch has no image in LDS
template <int ROW0, int NROWS>
__device__ inline void epnp5_L_rows_hand_d(const double *hand, int hs, double *Lm, int ls)
{
#pragma unroll
    for (int i = ROW0; i < ROW0 + NROWS; i++) {
        const int a = i < 3 ? 0 : i < 5 ? 1 : 2, b = i < 3 ? i + 1 : i < 5 ? i - 1 : 3;
        double dv[4][3];
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int c = 0; c < 3; c++)
                dv[q][c] = hand[(kEpnpHandV + q * 12 + 3 * a + c) * hs] - hand[(kEpnpHandV + q * 12 + 3 * b + c) * hs];
        double *row = Lm + 10 * i * ls;
        row[0 * ls] = dot3_d(dv[0], dv[0]);
        row[1 * ls] = 2.0 * dot3_d(dv[0], dv[1]);
        row[2 * ls] = dot3_d(dv[1], dv[1]);
        row[3 * ls] = 2.0 * dot3_d(dv[0], dv[2]);
        row[4 * ls] = 2.0 * dot3_d(dv[1], dv[2]);
        row[5 * ls] = dot3_d(dv[2], dv[2]);
        row[6 * ls] = 2.0 * dot3_d(dv[0], dv[3]);
        row[7 * ls] = 2.0 * dot3_d(dv[1], dv[3]);
        row[8 * ls] = 2.0 * dot3_d(dv[2], dv[3]);
        row[9 * ls] = dot3_d(dv[3], dv[3]);
    }
}
// compute_rho from the control points of the hand-over record into rho_m[0..5]
__device__ inline void epnp5_rho_d(const double *hand, int hs, double *rho_m, int rs)
{
    double cws[4][3];
#pragma unroll
    for (int i = 0; i < 12; i++) cws[i / 3][i % 3] = hand[(kEpnpHandCws + i) * hs];
    rho_m[0 * rs] = dist2_d(cws[0], cws[1]); rho_m[1 * rs] = dist2_d(cws[0], cws[2]);
    rho_m[2 * rs] = dist2_d(cws[0], cws[3]); rho_m[3 * rs] = dist2_d(cws[1], cws[2]);
    rho_m[4 * rs] = dist2_d(cws[1], cws[3]); rho_m[5 * rs] = dist2_d(cws[2], cws[3]);
}

// compute_ccs + compute_pcs + solve_for_sign + estimate_R_and_t + reprojection_error, operands from `hand`
__device__ inline double epnp_R_and_t_mem_d(const double *hand, int hs, double fu, double fv, double uc, double vc,
                                            const double (&betas)[4], double R[9], double t[3], double *ws, int st)
{
    const int n = 5;
    double ccs[4][3], pcs[15];
#pragma unroll
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * hand[(kEpnpHandV + 12 * i + 3 * j + k) * hs];
#pragma unroll
    for (int i = 0; i < n; i++) {
        double a[4];
#pragma unroll
        for (int q = 0; q < 4; q++) a[q] = hand[(kEpnpHandAlphas + 4 * i + q) * hs];
#pragma unroll
        for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[2] < 0.0) {
#pragma unroll
        for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];          // (the ccs are not used after this point)
    }
    double pws[15];
#pragma unroll
    for (int i = 0; i < 15; i++) pws[i] = hand[(kEpnpHandPws + i) * hs];
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { pc0[j] += pcs[3 * i + j]; pw0[j] += pws[3 * i + j]; }
#pragma unroll
    for (int j = 0; j < 3; j++) { pc0[j] /= n; pw0[j] /= n; }
    double abt[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, abt_d[3];
    double *abt_ut = ws, *abt_vt = ws + 9 * st;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pc = pcs + 3 * i, *pw = pws + 3 * i;
#pragma unroll
        for (int j = 0; j < 3; j++) {
            abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
            abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
            abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) abt_ut[(j * 3 + i) * st] = abt[i * 3 + j];
    jacobi_svd_d<3, 3>(abt_ut, st, abt_d, abt_vt, st, false);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            R[i * 3 + j] = abt_ut[(0 * 3 + i) * st] * abt_vt[(0 * 3 + j) * st] + abt_ut[(1 * 3 + i) * st] * abt_vt[(1 * 3 + j) * st] +
                           abt_ut[(2 * 3 + i) * st] * abt_vt[(2 * 3 + j) * st];
    const double det = R[0] * R[4] * R[8] + R[1] * R[5] * R[6] + R[2] * R[3] * R[7] -
                       R[2] * R[4] * R[6] - R[1] * R[3] * R[8] - R[0] * R[5] * R[7];
    if (det < 0) { R[6] = -R[6]; R[7] = -R[7]; R[8] = -R[8]; }
    t[0] = pc0[0] - dot3_d(R, pw0);
    t[1] = pc0[1] - dot3_d(R + 3, pw0);
    t[2] = pc0[2] - dot3_d(R + 6, pw0);
    double sum2 = 0.0;
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double *pw = pws + 3 * i;
        double Xc = dot3_d(R, pw) + t[0], Yc = dot3_d(R + 3, pw) + t[1];
        double inv_Zc = 1.0 / (dot3_d(R + 6, pw) + t[2]);
        double ue = uc + fu * Xc * inv_Zc, ve = vc + fv * Yc * inv_Zc;
        double u = hand[(kEpnpHandUs + 2 * i) * hs], vv = hand[(kEpnpHandUs + 2 * i + 1) * hs];
        sum2 += sqrt((u - ue) * (u - ue) + (vv - ve) * (vv - ve));
    }
    return sum2 / n;
}

// find_betas_approx_N + gauss_newton + R, t, reprojection error of ONE approximation N = 1, 2, 3
// (L and rho in memory, element e at Lm[e * ls] / rho_m[e * ls]: read where they are used)
template <int N>
__device__ inline double epnp5_betas_pose_d(const double *Lm, const double *rho_m, int ls, const double *hand, int hs, double fu,
                                            double fv, double uc, double vc, double *ws, int st, double Rout[9], double tout[3])
{
    double betas[4], bb[5];
    {
        double rho[6];
#pragma unroll
        for (int i = 0; i < 6; i++) rho[i] = rho_m[i * ls];
        constexpr int nc = N == 1 ? 4 : (N == 2 ? 3 : 5);
        double Lr[6 * nc];
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = 0; j < nc; j++) {
                const int col = N == 1 ? (j == 0 ? 0 : j == 1 ? 1 : j == 2 ? 3 : 6) : j;
                Lr[i * nc + j] = Lm[(10 * i + col) * ls];
            }
        svd_solve_d<6, nc>(Lr, rho, bb, ws, st);
    }
    if (N == 1) {
        if (bb[0] < 0) {
            betas[0] = sqrt(-bb[0]);
            betas[1] = -bb[1] / betas[0]; betas[2] = -bb[2] / betas[0]; betas[3] = -bb[3] / betas[0];
        } else {
            betas[0] = sqrt(bb[0]);
            betas[1] = bb[1] / betas[0]; betas[2] = bb[2] / betas[0]; betas[3] = bb[3] / betas[0];
        }
    } else {
        if (bb[0] < 0) {
            betas[0] = sqrt(-bb[0]);
            betas[1] = (bb[2] < 0) ? sqrt(-bb[2]) : 0.0;
        } else {
            betas[0] = sqrt(bb[0]);
            betas[1] = (bb[2] > 0) ? sqrt(bb[2]) : 0.0;
        }
        if (bb[1] < 0) betas[0] = -betas[0];
        betas[2] = N == 3 ? bb[3] / betas[0] : 0.0;
        betas[3] = 0.0;
    }
    for (int it = 0; it < 5; it++) {
        double A[24], b[6], x[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 6; i++) {
            double r[10];
#pragma unroll
            for (int q = 0; q < 10; q++) r[q] = Lm[(10 * i + q) * ls];
            double *ra = A + i * 4;
            ra[0] = 2 * r[0] * betas[0] + r[1] * betas[1] + r[3] * betas[2] + r[6] * betas[3];
            ra[1] = r[1] * betas[0] + 2 * r[2] * betas[1] + r[4] * betas[2] + r[7] * betas[3];
            ra[2] = r[3] * betas[0] + r[4] * betas[1] + 2 * r[5] * betas[2] + r[8] * betas[3];
            ra[3] = r[6] * betas[0] + r[7] * betas[1] + r[8] * betas[2] + 2 * r[9] * betas[3];
            b[i] = rho_m[i * ls] - (r[0] * betas[0] * betas[0] + r[1] * betas[0] * betas[1] +
                                    r[2] * betas[1] * betas[1] + r[3] * betas[0] * betas[2] +
                                    r[4] * betas[1] * betas[2] + r[5] * betas[2] * betas[2] +
                                    r[6] * betas[0] * betas[3] + r[7] * betas[1] * betas[3] +
                                    r[8] * betas[2] * betas[3] + r[9] * betas[3] * betas[3]);
        }
        epnp_qr_solve_d(A, b, x);
        for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
    return epnp_R_and_t_mem_d(hand, hs, fu, fv, uc, vc, betas, Rout, tout, ws, st);
}

// cv::Rodrigues vector -> matrix (+ 3x9 Jacobian)
__device__ inline void rodrigues_vec2mat_d(const double r[3], double R[9], double *J)
{
    double theta = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (theta < SVO_DBL_EPS) {
        for (int i = 0; i < 9; i++) R[i] = 0;
        R[0] = R[4] = R[8] = 1;
        if (J) {
            for (int i = 0; i < 27; i++) J[i] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = theta ? 1. / theta : 0.;
    double rx = r[0] * itheta, ry = r[1] * itheta, rz = r[2] * itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double r_x[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; k++) R[k] = c * I[k] + c1 * rrt[k] + s * r_x[k];
    if (J) {
        double drrt[27] = {rx + rx, ry, rz, ry, 0, 0, rz, 0, 0,
                           0, rx, 0, rx, ry + ry, rz, 0, rz, 0,
                           0, 0, rx, 0, 0, ry, rx, ry, rz + rz};
        const double d_r_x_[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0,
                                   0, 0, 1, 0, 0, 0, -1, 0, 0,
                                   0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++) {
            double ri = i == 0 ? rx : i == 1 ? ry : rz;
            double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; k++)
                J[i * 9 + k] = a0 * I[k] + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * r_x[k] + a4 * d_r_x_[i * 9 + k];
        }
    }
}

// cv::Rodrigues matrix -> vector
__device__ inline void rodrigues_mat2vec_d(const double Rin[9], double r[3], double *ws, int st)
{
    double W[3], R[9];
    double *At = ws, *Vt = ws + 9 * st;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) At[(j * 3 + i) * st] = Rin[i * 3 + j];
    jacobi_svd_d<3, 3>(At, st, W, Vt, st, false);
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += At[(k * 3 + i) * st] * Vt[(k * 3 + j) * st];
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

// PnPRansacCallback::computeError for one point (cvProjectPoints2 in double -> float -> float L2^2)
__device__ inline float reproj_err2_d(const double R[9], const double t[3], double fx, double fy, double cx,
                                      double cy, float Px, float Py, float Pz, float mx, float my)
{
    double X = Px, Y = Py, Z = Pz;
    double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    z = z ? 1. / z : 1;
    x *= z; y *= z;
    float px = (float)(x * fx + cx), py = (float)(y * fy + cy);
    float dx = mx - px, dy = my - py;
    float s = 0.f;
    s += dx * dx;
    s += dy * dy;
    return s;
}

// RANSACUpdateNumIters
__device__ inline int ransac_update_iters_d(double p, double ep, int model_points, int max_iters)
{
    p = p > 0. ? p : 0.; p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.; ep = ep < 1. ? ep : 1.;
    double num = 1. - p > SVO_DBL_MIN ? 1. - p : SVO_DBL_MIN;
    double denom = 1. - pow(1. - ep, (double)model_points);
    if (denom < SVO_DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)rint(num / denom);
}

__device__ inline uint32_t rng_next_d(uint64_t &state)
{
    state = (uint64_t)(uint32_t)state * 4164903690u + (uint32_t)(state >> 32);
    return (uint32_t)state;
}


// ---- P3P (round 5): cv::solvePnP(SOLVEPNP_P3P) on exactly four points -- what cv::solvePnPRansac runs when it is handed
// npoints == 4 (reachable with num_features_tracking = 4, reference src/tracking.cpp:274, 485).  The same arithmetic, in the
// same order, as oracle/p3p.c (p3p.cpp + polynom_solver.cpp as restated there); pow / acos / cos are the device library's, so
// the pose can differ from the oracle's in the last ulps -- it only seeds the LM refit on the four points.
/* polynom_solver.cpp */
__device__ inline int solve_deg2_d(double a, double b, double c, double *x1, double *x2)
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

__device__ inline int solve_deg3_d(double a, double b, double c, double d, double *x0, double *x1, double *x2)
{
    if (a == 0) {
        if (b == 0) {
            if (c == 0) return 0;
            *x0 = -d / c;
            return 1;
        }
        *x2 = 0;
        return solve_deg2_d(b, c, d, x0, x1);
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
        *x1 = 2 * sqrt_Q * cos((theta + 2 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        *x2 = 2 * sqrt_Q * cos((theta + 4 * 3.1415926535897932384626433832795) / 3.0) - b_a_3;
        return 3;
    }
    double AD = pow(fabs(R) + sqrt(D), 1.0 / 3.0) * (R > 0 ? 1 : (R < 0 ? -1 : 0));
    double BD = (AD == 0) ? 0 : -Q / AD;
    *x0 = AD + BD - b_a_3;
    return 1;
}

__device__ inline int solve_deg4_d(double a, double b, double c, double d, double e, double x[4])
{
    if (a == 0) { x[3] = 0; return solve_deg3_d(b, c, d, e, &x[0], &x[1], &x[2]); }
    double inv_a = 1. / a;
    b *= inv_a; c *= inv_a; d *= inv_a; e *= inv_a;
    double b2 = b * b, bc = b * c, b3 = b2 * b;
    double r0, r1, r2;
    int n = solve_deg3_d(1, -c, d * b - 4 * e, 4 * c * e - d * d - b2 * e, &r0, &r1, &r2);
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
__device__ inline int jacobi_4x4_d(double *A, double *D, double *U)
{
    double B[4], Z[4];
    int i, j, k, iter;
    for (i = 0; i < 16; i++) U[i] = (i % 5 == 0) ? 1. : 0.;
    B[0] = A[0]; B[1] = A[5]; B[2] = A[10]; B[3] = A[15];
    for (int q_ = 0; q_ < 4; q_++) D[q_] = B[q_];
    for (int q_ = 0; q_ < 4; q_++) Z[q_] = 0.;
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
        for (int q_ = 0; q_ < 4; q_++) D[q_] = B[q_];
        for (int q_ = 0; q_ < 4; q_++) Z[q_] = 0.;
    }
    return 0;
}

/* p3p::align: the rigid motion that takes the three object points to M_end (Horn's quaternion method) */
__device__ inline int p3p_align_d(double M_end[3][3], const double Xw[9], double R[3][3], double T[3])
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
    jacobi_4x4_d(Qs, evs, U);
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
__device__ inline int solve_for_lengths_d(double lengths[4][3], const double distances[3], const double cosines[3])
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
    int n = solve_deg4_d(A, B, C, D, E, real_roots), i;
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
__device__ inline int p3p_solve3_d(double R[4][3][3], double t[4][3], const double mu[3], const double mv[3], const double Xw[9],
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
    int n = solve_for_lengths_d(lengths, distances, cosines);
    int nb_solutions = 0;
    for (i = 0; i < n; i++) {
        double M_orig[3][3];
        for (j = 0; j < 3; j++) {
            M_orig[j][0] = lengths[i][j] * u[j];
            M_orig[j][1] = lengths[i][j] * v[j];
            M_orig[j][2] = lengths[i][j] * k[j];
        }
        if (!p3p_align_d(M_orig, Xw, R[nb_solutions], t[nb_solutions])) continue;
        nb_solutions++;
    }
    return nb_solutions;
}

/* p3p::solve with four points: pws = X0 Y0 Z0 .. X3 Y3 Z3, us = u0 v0 .. u3 v3 (pixels).  Returns 1 and R (row-major), t, or 0. */
__device__ inline int p3p4_d(const double pws[12], const double us[8], double fx, double fy, double cx, double cy, double R[9], double t[3])
{
    double Rs[4][3][3], ts[4][3];
    const double mu[3] = {us[0], us[2], us[4]}, mv[3] = {us[1], us[3], us[5]};
    int n = p3p_solve3_d(Rs, ts, mu, mv, pws, fx, fy, cx, cy), i, j, ns = 0;
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

}  // namespace svo

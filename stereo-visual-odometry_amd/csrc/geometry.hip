// geometry.hip -- triangulation, RANSAC-EPnP + Levenberg-Marquardt pose solver, motion gates and
// pose chaining for gfx950; replaces cv::triangulatePoints / cv::convertPointsFromHomogeneous
// (reference src/tracking.cpp:292-294), cv::solvePnPRansac + cv::Rodrigues (:485-488) and the
// gating / accumulation code (:305-329, 440-463).  "g2o" in the reference is link-only; the live
// solver is the one restated here.
//
// Kernels
//   triangulate_kernel : one thread per point, 4x4 DLT + one-sided Jacobi SVD in f64.
//   pnp_begin / pnp_hyp / pnp_score / pnp_select / pnp_refit : solvePnPRansac as a pipeline of
//       launches -- EPnP hypotheses 64 per workgroup (one per lane, 12x12 eigenproblem in LDS), all
//       blocks of a phase concurrently; inlier counts over many workgroups; OpenCV's sequential
//       model-selection rule replayed over the counts in order (exactly the serial algorithm's
//       winner, adaptive stop and mask; the cv::RNG multiply-with-carry sequence, seed 2^64-1, is
//       drawn serially); LM refit on the inliers with 256 threads per pair.  Details above
//       struct PnpRecord.
//   finalize_kernel    : per pair failure staging, Euler / translation gates, inv([R t;0 1]).
//   chain_kernel       : frame_pose_ *= T^-1 over the batch, skipping failed steps.
#include <cstring>
#include <cstdlib>
#include "svo_ctx.h"
#include "geom_device.h"

namespace svo {

// ------------------------------------------------------------------------------------------
struct PnpState;
// the keypoint counts (and capacity flags) of the frames of a pair, frozen for the pose stage's gates while the next batch's
// front end overwrites the live ones: snap[p] = n(previous), snap[n_pairs + p] = n(current), snap[2 n_pairs + p] = OR of the
// flags of the pair's image slots (ORB mode: per = 2 slots per frame).  Written by the pair's begin workgroup.
struct SnapArgs { const int *n, *ovf; int fp0, fc0, fstep, per, n_pairs; int *snap; };
struct PnpBeginArgs { PnpState *state; int *counts; int *subsets; int iterations, first_cap; const uint64_t *stream; SnapArgs snap; };
__device__ void pnp_begin_item(const PnpBeginArgs &a, int n, int b, int lane);

struct TriArgs {
    double P1[12], P2[12];
    const float2 *x1, *x2; float *out3; int64_t stride;    // item b at + b*stride points
    const int *n_pts; int n_fixed;
    int fuse_begin; PnpBeginArgs begin;                     // the batch pipeline: the last workgroup of an item starts its solvePnPRansac
};

// grid.x workgroups per item walk its points in strides (the launch is sized from the batch, not
// from the keypoint capacity: most items hold far fewer points than max_keypoints)
__global__ __launch_bounds__(64) void triangulate_kernel(TriArgs a)
{
    const int b = blockIdx.y;
    const int n = a.n_pts ? a.n_pts[b] : a.n_fixed;
    const int gx = gridDim.x - a.fuse_begin;
    if ((int)blockIdx.x == gx) {
        // solvePnPRansac's per-item state and first point subsets depend on the point COUNT only: prepared
        // here, beside the triangulation, instead of by a launch of their own (20 us of the online path)
        pnp_begin_item(a.begin, n, b, threadIdx.x);
        return;
    }
    for (int i = blockIdx.x * 64 + threadIdx.x; i < n; i += gx * 64) {
        const int64_t o = (int64_t)b * a.stride + i;
        const float2 p1 = a.x1[o], p2 = a.x2[o];
        const double xs[2] = {(double)p1.x, (double)p2.x}, ys[2] = {(double)p1.y, (double)p2.y};
        double At[16], V4[4];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const double *P = j == 0 ? a.P1 : a.P2;
#pragma unroll
            for (int k = 0; k < 4; k++) {                   // A row (2j) and (2j + 1), stored transposed
                At[k * 4 + 2 * j] = xs[j] * P[8 + k] - P[k];
                At[k * 4 + 2 * j + 1] = ys[j] * P[8 + k] - P[4 + k];
            }
        }
        jacobi_null4_d(At, V4);
        // 4 x N CV_32F homogeneous result, then convertPointsFromHomogeneous in float
        const float X = (float)V4[0], Y = (float)V4[1], Z = (float)V4[2], Wh = (float)V4[3];
        const float scale = Wh != 0.f ? 1.f / Wh : 1.f;
        a.out3[o * 3 + 0] = X * scale; a.out3[o * 3 + 1] = Y * scale; a.out3[o * 3 + 2] = Z * scale;
    }
}

// ------------------------------------------------------------------------------------------
// solvePnPRansac as a short pipeline of launches (all on one stream, no host round trip):
//
//   pnp_begin    : per item state (niters, best model, cv::RNG state) + the first 64 point subsets
//   per PHASE p (phase 0 = hypotheses 0..63, later phases up to 448 hypotheses each):
//     pnp_hyp    : one 64-lane workgroup per block of 64 hypotheses: lane h runs the 5-point EPnP of
//                  hypothesis base+h (12x12 eigenproblem in a lane-interleaved 72 KB LDS image) -- the
//                  blocks of a phase run CONCURRENTLY (the single-wave version ran them one after the
//                  other: 8 x the EPnP latency at iterationsCount 500 with few inliers);
//     pnp_score  : inlier counts of every hypothesis of the phase, points spread over many
//                  workgroups (lane = hypothesis, four waves x grid.z point chunks; integer counts
//                  added with atomics: order-free, exact);
//     pnp_select : replays OpenCV's sequential "better model -> shrink niters" rule over the counts
//                  in hypothesis order, so the winner, the adaptive stop and the iteration count are
//                  exactly those of the serial algorithm; then draws the next phase's subsets (the
//                  RNG is inherently serial: one lane) if the rule still wants more hypotheses.
//                  Hypotheses past the stop were computed speculatively and are ignored; workgroups
//                  of a phase that is not needed exit on their first instruction.
//   pnp_refit    : inlier mask of the winner + Levenberg-Marquardt on the inliers with 256 threads
//                  per item (block-wide reductions in a fixed order; every thread sees identical sums).
//
// Exactness: hypotheses, counts, winner, stop and mask are bit-identical to the oracle (same
// expressions, IEEE f64); only the LM sums are associated differently (observed pose error <= 1e-9).
struct PnpRecord {                 // device-side record of one solve
    double rvec[3], tvec[3], R[9];
    int n_inliers, ransac_iters, best_iter, lm_iters, ok, n;
};

constexpr int kHypBlock = 64;                  // hypotheses per EPnP workgroup (one per lane)
constexpr int kPhaseHyps = 512;                // capacity of a phase (hypotheses, counts, subsets, hand-over records per item)
// The FIRST phase is 64 hypotheses where the adaptive stop usually ends the search inside them (LK mode: ~20
// iterations at 70 % inliers -- anything more would be computed to be thrown away), and the whole iterationsCount
// (<= 512: one phase) where it never does (ORB mode: ~18 % inliers, always the full 500): the lone pair of the
// online path then pays ONE hypothesis launch instead of two in sequence (237 us each).  The subsets of the first
// phase are drawn by the begin workgroup, which runs beside the triangulation launch.
inline int pnp_first_cap(const svo_config &cfg) { return cfg.track_mode == SVO_MODE_ORB ? kPhaseHyps : 64; }
struct PnpState {                              // per item, lives across the launches of one solve
    double bestRt[12];
    uint64_t rng;
    int n, niters, max_good, best_iter, iters_done, next_base, phase_hyps, _pad;
};
struct PnpHyp { double R[9], t[3]; };

struct PnpArgs {
    const float *X3; const float2 *img; int64_t stride;     // points of item b at + b*stride
    const int *n_pts; int n_fixed;
    double fx, fy, cx, cy;
    int iterations; float reproj_err; double confidence;
    uint8_t *mask;                                          // stride bytes per item
    PnpRecord *out;
    PnpState *state; PnpHyp *hyp; int *counts; int *subsets;   // per item: 1, kPhaseHyps, kPhaseHyps, 2 x 5 * kPhaseHyps
    int phase_base, phase_cap, phase_index;                 // hypotheses [phase_base, phase_base + phase_cap) in this phase
    int score_chunk;                                        // points per scoring workgroup (grid.z chunks)
    int refit_svd;                                          // 1: always take the SVD route of the refit's solves (SVO_REFIT_SVD=1; tests)
    int first_cap;                                          // hypotheses of the first phase (pnp_first_cap)
    const uint64_t *stream;                                 // cv::RNG(-1)'s first kRngStream states (draw_subsets_stream)
    double *hand;                                           // EPnP hand-over records: per item kPhaseBlocks x 105 x 64 doubles (pnp_hyp_body)
};

__device__ inline double wave_allsum_f64(double v)
{
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v = v + __shfl_xor(v, m, 64);
    return v;
}

// cv::RNG draws for RANSACPointSetRegistrator::getSubset.  The generator is a serial chain
// (multiply-with-carry) and the number of draws per subset depends on the data (a duplicate index is
// redrawn), so subsets are drawn by ONE wave, and everything is kept wave-uniform: the compiler
// then runs the chain on the scalar unit (s_mul_i32 / s_mul_hi_u32, ~5x shorter latency per step
// than the vector unit; drawn by a single divergent lane 64 subsets took 41 us).  x % n is one
// multiply-high by floor(2^32 / n) and at most two corrections instead of a 32-bit division.
struct FastMod { uint32_t n, m; };
__device__ inline FastMod fastmod_make(uint32_t n) { FastMod f; f.n = n; f.m = n > 1 ? (uint32_t)(0x100000000ull / n) : 0u; return f; }
__device__ inline uint32_t fastmod(uint32_t x, FastMod f)
{
    uint32_t r = x - __umulhi(x, f.m) * f.n;             // quotient estimate is low by at most 2
    if (r >= f.n) r -= f.n;
    if (r >= f.n) r -= f.n;
    return r;
}
__device__ inline uint32_t sgpr(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// `count` consecutive subsets into dst[5 * count]; all lanes of the wave call this with the same
// arguments, lanes 0..4 store.  Returns the advanced state.
__device__ inline uint64_t draw_subsets(uint64_t rng_in, int n, int count, int *dst, int lane)
{
    uint32_t lo = sgpr((uint32_t)rng_in), hi = sgpr((uint32_t)(rng_in >> 32));
    const FastMod fm = fastmod_make(sgpr((uint32_t)n));
    if (fm.n <= 1) return rng_in;
    for (int h = 0; h < count; h++) {
        uint32_t idx[5];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            for (;;) {
                // state = (uint32)state * 4164903690 + (state >> 32)
                const uint64_t st = (uint64_t)lo * 4164903690u + hi;
                lo = sgpr((uint32_t)st); hi = sgpr((uint32_t)(st >> 32));
                idx[i] = sgpr(fastmod(lo, fm));
                bool dup = false;
#pragma unroll
                for (int j = 0; j < i; j++) dup = dup || idx[j] == idx[i];
                if (!dup) break;
            }
        }
        const uint32_t mine = lane == 0 ? idx[0] : lane == 1 ? idx[1] : lane == 2 ? idx[2] : lane == 3 ? idx[3] : idx[4];
        if (lane < 5) dst[h * 5 + lane] = (int)mine;
    }
    return ((uint64_t)hi << 32) | lo;
}

// The FIRST subsets of a solve, drawn by the 64 lanes together.  RANSACPointSetRegistrator::run seeds its cv::RNG with
// (uint64)-1 on every call, so the raw 32-bit draws of a solve are a CONSTANT sequence: the context holds the first
// kRngStream generator states (geom_workspace_init), and only what depends on the data is computed here --
//   (1) idx[p] = draw p modulo the point count, for the whole stretch at once (LDS);
//   (2) a subset that starts at position p consumes 5 draws unless it meets a duplicate index: one flag per position
//       (each lane plays getSubset from its position), gathered by residue of p modulo 5 into bit masks;
//   (3) the chain "subset k+1 starts where subset k ended" is then a walk over the FLAGGED subsets only (a few per
//       hundred at a few hundred points): between two of them the starts are an arithmetic progression of step 5,
//       found with one find-first-set per 64 subsets; a flagged subset is replayed (wave-uniform) for its length;
//   (4) every subset's five indices are read off the stream from its start, a lane per subset.
// The result is the serial generator's, draw for draw; when the stretch runs out (tiny point counts redraw a lot)
// the caller continues with draw_subsets from the state reached.  512 subsets: ~25 us instead of 130 (the scalar
// chain is ~100 cycles per draw), on the critical path of the online pair in ORB mode.
constexpr int kRngStream = 4096;
struct DrawLds { uint16_t idx[kRngStream + 8]; uint16_t pos[kPhaseHyps]; };
__device__ inline int draw_subsets_stream(const uint64_t *stream, int n, int count, int *dst, int lane, uint64_t *state_out)
{
    __shared__ DrawLds s;
    int T = 5 * count + (count >> 1) + 64;
    T = T < kRngStream ? T : kRngStream;
    const FastMod fm = fastmod_make((uint32_t)n);
    for (int p0 = 0; p0 < T; p0 += 512) {                       // (eight loads in flight)
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = stream[min(p0 + 64 * u + lane, kRngStream - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (p0 + 64 * u + lane < T) s.idx[p0 + 64 * u + lane] = (uint16_t)fastmod((uint32_t)v[u], fm);
    }
    wave_lds_fence();
    // five consecutive draws: all different?  (the common case: a subset of exactly five draws)
    auto five = [&](int p, uint32_t id[5]) -> bool {
#pragma unroll
        for (int i = 0; i < 5; i++) id[i] = s.idx[p + i];       // (idx has 8 entries of slack past the stretch)
        return id[0] != id[1] && id[0] != id[2] && id[0] != id[3] && id[0] != id[4] && id[1] != id[2] && id[1] != id[3] &&
               id[1] != id[4] && id[2] != id[3] && id[2] != id[4] && id[3] != id[4];
    };
    // getSubset from position p in general: the position after its fifth distinct index, or -1 when the stretch ends first
    auto play = [&](int p, uint32_t id[5]) -> int {
        int pp = p;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            for (;;) {
                if (pp >= T) return -1;
                id[i] = s.idx[pp++];
                bool dup = false;
#pragma unroll
                for (int j = 0; j < i; j++) dup = dup || id[j] == id[i];
                if (!dup) break;
            }
        }
        return pp;
    };
    // flags by position: word c (positions 64 c + lane) in lane c of (flo, fhi); the last positions of the stretch are flagged
    const int W = (T + 63) / 64;
    uint32_t flo = 0, fhi = 0;
    for (int c = 0; c < W; c++) {
        const int p = 64 * c + lane;
        uint32_t id[5];
        const bool distinct = five(min(p, T), id);
        const unsigned long long m = __ballot(p + 5 > T || !distinct);
        if (lane == c) { flo = (uint32_t)m; fhi = (uint32_t)(m >> 32); }
    }
    // the walk (wave-uniform): from a subset start p, the next flagged start among p, p + 5, p + 10, ... -- the positions
    // congruent to p modulo 5 are bits (p + c) % 5, + 5, + 10, ... of word c (64 = 4 modulo 5)
    int k = 0, p = 0;
    while (k < count && p < T) {
        const int r = p % 5;
        int pn = -1, bit = p & 63;
        for (int c = p >> 6; c < W; c++, bit = 0) {
            unsigned long long w = ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)fhi, c) << 32) |
                                   (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)flo, c);
            w &= (0x1084210842108421ull << ((r + c) % 5)) & (~0ull << bit);
            if (w) { pn = c * 64 + __ffsll((long long)w) - 1; break; }
        }
        if (pn < 0) break;                                      // (cannot happen: every class has a flagged position at the end)
        int take = (pn - p) / 5;
        take = take < count - k ? take : count - k;
        for (int j = lane; j < take; j += 64) s.pos[k + j] = (uint16_t)(p + 5 * j);
        k += take; p += 5 * take;
        if (k >= count || p >= T) break;
        // a subset with a redraw (or the end of the stretch): replay it for its length
        uint32_t id[5];
        const int pe = play(p, id);                             // (uniform: every lane replays the same subset)
        if (pe < 0) break;
        if (lane == 0) s.pos[k] = (uint16_t)p;
        k++; p = sgpr((uint32_t)pe);
    }
    wave_lds_fence();
    for (int kk = lane; kk < k; kk += 64) {
        uint32_t id[5];
        const int ps = s.pos[kk];
        if (ps + 5 > T || !five(ps, id)) play(ps, id);
#pragma unroll
        for (int i = 0; i < 5; i++) dst[kk * 5 + i] = (int)id[i];
    }
    *state_out = p == 0 ? ~0ull : stream[p - 1];
    return k;
}

// Subsets are double-buffered by phase parity: while the EPnP blocks of phase p run, one extra
// workgroup of the same launch draws the subsets phase p+1 would need (they depend on the RNG state
// only, not on the selection), so the serial drawing never sits on the critical path.
__device__ void pnp_begin_item(const PnpBeginArgs &a, int n, int b, int lane)
{
    PnpState *st = a.state + b;
    int *counts = a.counts + (int64_t)b * kPhaseHyps;
    if (a.snap.snap && lane == 0) {
        const SnapArgs &q = a.snap;
        const int sp = q.per * (q.fp0 + b * q.fstep), sc = q.per * (q.fc0 + b * q.fstep);
        q.snap[b] = q.n[sp];
        q.snap[q.n_pairs + b] = q.n[sc];
        if (q.ovf) q.snap[2 * q.n_pairs + b] = q.per == 2 ? (q.ovf[sp] | q.ovf[sp + 1] | q.ovf[sc] | q.ovf[sc + 1]) : (q.ovf[sp] | q.ovf[sc]);
    }
    for (int i = lane; i < kPhaseHyps; i += 64) counts[i] = 0;
    const int niters = a.iterations > 1 ? a.iterations : 1;
    uint64_t rng = ~0ull;
    int *sub = a.subsets + (int64_t)b * 2 * kPhaseHyps * 5;         // phase 0 -> buffer 0
    int hyps = 0;
    if (n >= 5) {
        hyps = niters < a.first_cap ? niters : a.first_cap;
        if (n > 5) {
            int done = 0;
            if (a.stream && n <= 65535) done = draw_subsets_stream(a.stream, n, hyps, sub, lane, &rng);
            if (done < hyps) rng = draw_subsets(rng, n, hyps - done, sub + 5 * done, lane);
        } else { hyps = 1; if (lane < 5) sub[lane] = lane; }          // npoints == model_points: one solve, all inliers
    } else if (n == 4) {
        // "if (npoints == 4) { model_points = 4; ransac_kernel_method = SOLVEPNP_P3P; }": one P3P solve on the four points
        hyps = 1;
        if (lane < 4) sub[lane] = lane;
    }
    if (lane != 0) return;
    st->n = n; st->niters = niters;
    st->max_good = 0; st->best_iter = -1; st->iters_done = 0; st->next_base = 0;
    st->phase_hyps = hyps;
    st->rng = rng;
}
__global__ __launch_bounds__(64) void pnp_begin_kernel(PnpArgs a)
{
    const PnpBeginArgs ba{a.state, a.counts, a.subsets, a.iterations, a.first_cap, a.stream, SnapArgs{}};
    pnp_begin_item(ba, a.n_pts ? a.n_pts[blockIdx.x] : a.n_fixed, blockIdx.x, threadIdx.x);
}

// One workgroup = 64 hypotheses (lane = hypothesis) and FOUR waves.  A lane-private EPnP is VALU-issue
// bound on its SIMD (f64 instructions issue every ~4.5 cycles; two thirds of them are the Jacobi SVD
// of the 12x12 M^T M, 66 pairs per sweep), so the block is spread over the CU's four SIMDs:
//   * wave 0 runs the lane-private front part (control points, barycentric coordinates, M^T M);
//   * all four waves rotate DISJOINT pairs of the 12x12 problem at the same time
//     (jacobi_sweeps_coop: 21 barrier-separated stages instead of 66 sequential pairs, same bits; a
//     stage with five or six pairs gives a wave two of them in one interleaved instruction stream);
//   * waves 0..2 run epnp's three beta approximations (N = 1, 2, 3) side by side; wave 0 applies
//     compute_pose's selection rule.
// Between the phases the hypothesis' state lives in MEMORY, not in registers: the front writes pws / us /
// alphas / cws (57 doubles) to a lane-interleaved hand-over record in the workspace (global memory: the
// LDS is full of the 12x12 image), wave 0 adds the four null-space vectors after the SVD, and the back part
// loads what it needs where it needs it (epnp5_L_rho_d, epnp5_betas_pose_d).  Carried in registers across
// the whole kernel the same state was 300+ live f64 values per lane: 1 122 spilled VGPRs in the
// 256-register build, 243 in the 512-register one.  No wave leaves before the last barrier.
// LDS: the lane-interleaved 12x12 image + the singular values (156 doubles per lane); after the SVD the same
// area holds L and rho (shared by the three approximations: 66 doubles) and the three waves' workspaces:
// 188 doubles per lane, 94 KB -- one workgroup per CU, which is also what the launch beside the next
// batch's front end wants (two would lock every LDS-staged front-end kernel out of the CU).
// ONE build, one wave per SIMD (324 registers): with the state in memory the body no longer needs the 512 the
// round-2 kernel took for the online path, and a 256-register build for batches -- kept then so that two waves
// fit a SIMD beside the next batch's front end -- ran its back part out of scratch memory at half the speed
// (ORB mode: 27.3 k pairs/s with it, 28.4 k with this one; the front end's waves still find 188 registers
// per SIMD beside a hypothesis wave).
constexpr int kPnpLdsDoubles = 66 + 40 + 27 + 55;          // per lane: L + rho + the three solve workspaces (> the 144 + 12 of the SVD)
constexpr size_t kPnpLdsBytes = (size_t)(kPnpLdsDoubles * 64) * sizeof(double);   // 94 KB: ONE workgroup per CU
constexpr int kPhaseBlocks = kPhaseHyps / kHypBlock;       // hand-over records: one per block of a phase
// The hypothesis launch is TWO kernels:
//   pnp_hyp_front_kernel: the lane-private front (wave 0), the 12x12 eigenproblem by staged Jacobi (four waves), the
//       sort, the four null-space vectors into the hand-over record.  LDS: the lane-interleaved image + singular
//       values = 78 KB, at most 256 registers: TWO blocks share a CU, i.e. two waves per SIMD -- the sweeps are one
//       dependent f64 chain per stage, which a lone wave issues at 8 cycles per instruction and two waves at 4 each
//       (profiles/r02_valu_roof.txt).  The sweeps are 62 % of a hypothesis' cycles.
//   pnp_hyp_back_kernel: L and rho (shared by the three approximations), the three beta approximations side by side
//       on three waves, compute_pose's rule.  94 KB of LDS, 300+ registers: one block per CU, as the single kernel was.
// Everything a hypothesis carries from one to the other is in its hand-over record (global memory, L2-resident).
constexpr size_t kPnpFrontLdsBytes = (size_t)((144 + 12) * 64) * sizeof(double);          // 78 KB
__device__ __forceinline__ void pnp_hyp_front_body(const PnpArgs &a, double *pnp_smem_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    PnpState *st = a.state + b;
    const int h = blockIdx.x * kHypBlock + lane;             // hypothesis index inside the phase
    if (st->next_base != a.phase_base) return;               // phase not needed (uniform for the block)
    if (blockIdx.x == gridDim.x - 1) {
        // the drawer: subsets of the NEXT phase into the other buffer (see pnp_begin_kernel); this block
        // has no barrier
        const int next = a.phase_base + a.phase_cap, niters = a.iterations > 1 ? a.iterations : 1;
        const int n = st->n;
        if (wave == 0 && next < niters && n > 5 && st->phase_hyps > 0) {
            const int more = niters - next < kPhaseHyps ? niters - next : kPhaseHyps;
            const uint64_t rng = draw_subsets(st->rng, n, more, a.subsets + ((int64_t)b * 2 + (a.phase_index + 1) % 2) * kPhaseHyps * 5, lane);
            if (lane == 0) st->rng = rng;
        }
        return;
    }
    if (blockIdx.x * kHypBlock >= st->phase_hyps) return;    // block beyond the phase (uniform)
    if (st->n == 4) {
        // exactly four points: cv::solvePnPRansac's kernel is P3P (geom_device.h p3p4_d), ONE model from all four; a solve
        // without a solution leaves the marker -1 in the hypothesis' inlier count (the selection rule then finds no model)
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const float *X3 = a.X3 + (int64_t)b * a.stride * 3;
            const float2 *img = a.img + (int64_t)b * a.stride;
            const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
            double pws[12], us[8];
            for (int i = 0; i < 4; i++) {
                pws[3 * i] = (double)X3[3 * i]; pws[3 * i + 1] = (double)X3[3 * i + 1]; pws[3 * i + 2] = (double)X3[3 * i + 2];
                const float2 m = img[i];
                const float xn = (float)(((double)m.x - cx) * (1. / fx));       // undistortPoints: float out
                const float yn = (float)(((double)m.y - cy) * (1. / fy));
                us[2 * i] = (double)xn * fx + cx;                                // p3p::extract_points: back to pixels
                us[2 * i + 1] = (double)yn * fy + cy;
            }
            PnpHyp out;
            if (p3p4_d(pws, us, fx, fy, cx, cy, out.R, out.t)) a.hyp[(int64_t)b * kPhaseHyps] = out;
            else a.counts[(int64_t)b * kPhaseHyps] = -1;
        }
        return;
    }
    const bool active = h < st->phase_hyps;                  // idle lanes solve points 0..4 (they take part in the barriers)
    double *big = pnp_smem_w + lane, *W = pnp_smem_w + 144 * 64 + lane;
    double *hand = a.hand + ((int64_t)b * kPhaseBlocks + blockIdx.x) * (kEpnpHandDoubles * 64) + lane;   // element e at hand[e * 64]
    if (wave == 0) {
        Epnp5 e;
        const float *X3 = a.X3 + (int64_t)b * a.stride * 3;
        const float2 *img = a.img + (int64_t)b * a.stride;
        const int *my = a.subsets + (((int64_t)b * 2 + a.phase_index % 2) * kPhaseHyps + h) * 5;
        const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
        e.fu = fx; e.fv = fy; e.uc = cx; e.vc = cy;
        for (int i = 0; i < 5; i++) {
            const int s = active ? my[i] : i;
            e.pws[3 * i] = (double)X3[3 * s]; e.pws[3 * i + 1] = (double)X3[3 * s + 1]; e.pws[3 * i + 2] = (double)X3[3 * s + 2];
            const float2 m = img[s];
            const float xn = (float)(((double)m.x - cx) * (1. / fx));
            const float yn = (float)(((double)m.y - cy) * (1. / fy));
            e.us[2 * i] = (double)xn * fx + cx;
            e.us[2 * i + 1] = (double)yn * fy + cy;
        }
        epnp5_front_d(e, big, 64);
        for (int i = 0; i < 15; i++) hand[(kEpnpHandPws + i) * 64] = e.pws[i];
        for (int i = 0; i < 10; i++) hand[(kEpnpHandUs + i) * 64] = e.us[i];
        for (int i = 0; i < 20; i++) hand[(kEpnpHandAlphas + i) * 64] = e.alphas[i];
        for (int i = 0; i < 12; i++) hand[(kEpnpHandCws + i) * 64] = e.cws[i / 3][i % 3];
        jacobi_init_d<12, 12>(big, 64, W, 64, nullptr, 0);
    }
    __syncthreads();
    jacobi_sweeps_coop<12, 12, 4>(big, 64, W, 64, nullptr, 0, wave);
    if (wave == 0) {
        double d12[12];
        jacobi_finish_d<12, 12>(big, 64, d12, nullptr, 0, true);
        for (int i = 0; i < 4; i++)                          // ut rows 11, 10, 9, 8
            for (int k = 0; k < 12; k++) hand[(kEpnpHandV + i * 12 + k) * 64] = big[((11 - i) * 12 + k) * 64];
    }
}
// Two builds.  WPE = 2: 256 registers a lane (82 spilled, all in the straight-line front) -- the lone online pair, whose eight
// blocks never share a CU with anything, and LK mode.  WPE = 3: 168 registers (338 spilled): the ORB-mode batch build.  Two blocks still share a CU
// (78 KB of LDS each), but they now hold 336 of a SIMD's 512 registers instead of all of them, so the next batch's extraction
// kernels run BESIDE the hypothesis blocks instead of waiting for block turnover: the pose stage alone 1.63 -> 1.69 ms, the
// overlapped ORB step 5.92 -> 5.81 ms (same box; 128 registers: 1.87 ms alone, the next matcher waits for it, 6.05 ms).
// (amdgpu_num_vgpr is not honoured for this kernel: the register budget comes in the steps waves_per_eu allows.)
template <int WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void pnp_hyp_front_kernel(PnpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double pnp_smem[];
    pnp_hyp_front_body(a, pnp_smem);
}

__device__ __forceinline__ void pnp_hyp_back_body(const PnpArgs &a, double *pnp_smem_w)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    const PnpState *st = a.state + b;
    const int h = blockIdx.x * kHypBlock + lane;
    if (st->next_base != a.phase_base) return;               // phase not needed
    if (blockIdx.x * kHypBlock >= st->phase_hyps) return;    // block beyond the phase
    if (st->n == 4) return;                                  // a P3P item: the front kernel wrote its one model
    const bool active = h < st->phase_hyps;
    double *big = pnp_smem_w + lane;
    const double *hand = a.hand + ((int64_t)b * kPhaseBlocks + blockIdx.x) * (kEpnpHandDoubles * 64) + lane;
    // L (6 x 10) and rho are the same for the three approximations: computed once, two rows per wave and rho by the fourth
    double *Lm = big, *rho_m = big + 60 * 64;
    if (wave == 0) epnp5_L_rows_hand_d<0, 2>(hand, 64, Lm, 64);
    else if (wave == 1) epnp5_L_rows_hand_d<2, 2>(hand, 64, Lm, 64);
    else if (wave == 2) epnp5_L_rows_hand_d<4, 2>(hand, 64, Lm, 64);
    else epnp5_rho_d(hand, 64, rho_m, 64);
    __syncthreads();
    PnpHyp out;
    double rep = 0;
    // workspaces behind L and rho: svd_solve of 6 x 4 / 6 x 3 / 6 x 5 (M*N + N*N doubles: 40, 27, 55); a wave's
    // results go to the start of its own workspace
    double *ws = big + (66 + (wave == 0 ? 0 : wave == 1 ? 40 : 67)) * 64;
    if (wave == 0) rep = epnp5_betas_pose_d<1>(Lm, rho_m, 64, hand, 64, a.fx, a.fy, a.cx, a.cy, ws, 64, out.R, out.t);
    else if (wave == 1) rep = epnp5_betas_pose_d<2>(Lm, rho_m, 64, hand, 64, a.fx, a.fy, a.cx, a.cy, ws, 64, out.R, out.t);
    else if (wave == 2) rep = epnp5_betas_pose_d<3>(Lm, rho_m, 64, hand, 64, a.fx, a.fy, a.cx, a.cy, ws, 64, out.R, out.t);
    if (wave == 1 || wave == 2) {
        double *dst = ws;
        for (int i = 0; i < 9; i++) dst[i * 64] = out.R[i];
        for (int i = 0; i < 3; i++) dst[(9 + i) * 64] = out.t[i];
        dst[12 * 64] = rep;
    }
    __syncthreads();
    if (wave == 0) {
        // epnp::compute_pose: "N = 1; if (rep[2] < rep[1]) N = 2; if (rep[3] < rep[N]) N = 3"
#pragma unroll
        for (int w = 1; w <= 2; w++) {
            const double *src = big + (66 + (w == 1 ? 40 : 67)) * 64;
            if (src[12 * 64] < rep) {
                rep = src[12 * 64];
                for (int i = 0; i < 9; i++) out.R[i] = src[i * 64];
                for (int i = 0; i < 3; i++) out.t[i] = src[(9 + i) * 64];
            }
        }
        if (active) a.hyp[(int64_t)b * kPhaseHyps + h] = out;
    }
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void pnp_hyp_back_kernel(PnpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double pnp_smem[];
    pnp_hyp_back_body(a, pnp_smem);
}

// findInliers for the hypotheses of the phase: grid (blocks of 64 hypotheses, items, point chunks),
// 256 threads; lane = hypothesis, the workgroup's points are staged in LDS 256 at a time and every
// wave takes a quarter of them (LDS broadcast reads).
constexpr int kScoreChunk = 1024;              // points per workgroup (grid.z chunks)
__global__ __launch_bounds__(256) void pnp_score_kernel(PnpArgs a)
{
    __shared__ float sP[256 * 5];
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const PnpState *st = a.state + b;
    const int n = st->n;
    const int h = blockIdx.x * kHypBlock + lane;
    if (st->next_base != a.phase_base || blockIdx.x * kHypBlock >= st->phase_hyps || n <= 5) return;
    const int i0 = blockIdx.z * a.score_chunk, i1 = min(n, i0 + a.score_chunk);
    if (i0 >= n) return;
    const float *X3 = a.X3 + (int64_t)b * a.stride * 3;
    const float2 *img = a.img + (int64_t)b * a.stride;
    const bool active = h < st->phase_hyps;
    PnpHyp H;
    if (active) H = a.hyp[(int64_t)b * kPhaseHyps + h];
    else { for (int q = 0; q < 9; q++) H.R[q] = 0; H.t[0] = H.t[1] = H.t[2] = 0; }
    const float thr2 = (float)((double)a.reproj_err * (double)a.reproj_err);
    int good = 0;
    for (int c0 = i0; c0 < i1; c0 += 256) {
        __syncthreads();
        const int i = c0 + threadIdx.x;
        if (i < i1) {
            const float2 m = img[i];
            sP[threadIdx.x * 5] = X3[3 * i]; sP[threadIdx.x * 5 + 1] = X3[3 * i + 1]; sP[threadIdx.x * 5 + 2] = X3[3 * i + 2];
            sP[threadIdx.x * 5 + 3] = m.x; sP[threadIdx.x * 5 + 4] = m.y;
        }
        __syncthreads();
        const int cnt = min(256, i1 - c0);
        for (int k = wave; k < cnt; k += 4) {
            const float *q = sP + k * 5;
            good += reproj_err2_d(H.R, H.t, a.fx, a.fy, a.cx, a.cy, q[0], q[1], q[2], q[3], q[4]) <= thr2;
        }
    }
    if (active && good) atomicAdd(a.counts + (int64_t)b * kPhaseHyps + h, good);
}

// inclusive prefix maximum over the wave's lanes on the DPP network (row shifts, then the two row broadcasts)
__device__ __forceinline__ int wave_incl_max(int v)
{
    const int lowest = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(lowest, v, 0x143, 0xc, 0xf, false));
    return v;
}

// OpenCV's loop "for (iter = 0; iter < niters; iter++) { count inliers; if better than the best so far: keep, shrink niters }"
// replayed over the phase's counts.  Only the hypotheses that BEAT every earlier one change anything (a handful in 500:
// the running maximum is a prefix maximum, found by the 64 lanes together), and between two of them niters is constant,
// so the replay is a walk over those few in order -- winner, stop and iteration count are the serial loop's.  (One lane
// reading 500 counts one after the other was 62 us of the online pair in ORB mode.)
__global__ __launch_bounds__(64) void pnp_select_kernel(PnpArgs a)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    PnpState *st = a.state + b;
    if (st->next_base != a.phase_base || st->phase_hyps <= 0) return;
    int *counts = a.counts + (int64_t)b * kPhaseHyps;
    const int n = st->n, model_points = n == 4 ? 4 : 5;
    const int hyps = st->phase_hyps;
    {
        int niters = st->niters, max_good = st->max_good, best_iter = st->best_iter, iters_done = st->iters_done, owner = -1;
        if (n == model_points) {
            // one hypothesis on exactly five (EPnP) or four (P3P) points: all inliers, one iteration; a P3P solve without a
            // solution (count marker -1) is no model at all
            if (a.phase_base < niters) {
                iters_done = a.phase_base + 1; niters = a.phase_base + 1;
                if (counts[0] >= 0) { max_good = n; best_iter = a.phase_base; owner = 0; }
            }
        } else if (a.phase_base < niters) {
            constexpr int kChunks = kPhaseHyps / 64;
            int g[kChunks];
#pragma unroll
            for (int c = 0; c < kChunks; c++) g[c] = c * 64 + lane < hyps ? counts[c * 64 + lane] : (int)0x80000000;
            int run_max = max_good > model_points - 1 ? max_good : model_points - 1;     // what a hypothesis has to beat
            int last_event = -1;
            bool stopped = false;
#pragma unroll
            for (int c = 0; c < kChunks; c++) {
                if (stopped || c * 64 >= hyps) continue;
                const int incl = wave_incl_max(g[c]);
                int before = __builtin_amdgcn_update_dpp((int)0x80000000, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1
                before = max(before, run_max);
                unsigned long long m = __ballot(g[c] > before);
                while (m) {
                    const int l = __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const int it = a.phase_base + c * 64 + l;
                    if (it >= niters) { stopped = true; break; }
                    max_good = __builtin_amdgcn_readlane(g[c], l); best_iter = it; owner = c * 64 + l; last_event = it;
                    niters = ransac_update_iters_d(a.confidence, (double)(n - max_good) / n, model_points, niters);
                }
                run_max = max(run_max, __builtin_amdgcn_readlane(incl, 63));
            }
            // the last iteration the loop ran: up to the end of the phase or of niters, and at least the last better model's
            const int upto = a.phase_base + hyps < niters ? a.phase_base + hyps : niters;
            iters_done = upto > last_event + 1 ? upto : last_event + 1;
        }
        if (lane == 0) {
            if (owner >= 0) {
                const PnpHyp *H = a.hyp + (int64_t)b * kPhaseHyps + owner;
                for (int i = 0; i < 9; i++) st->bestRt[i] = H->R[i];
                for (int i = 0; i < 3; i++) st->bestRt[9 + i] = H->t[i];
            }
            st->niters = niters; st->max_good = max_good; st->best_iter = best_iter; st->iters_done = iters_done;
            // the next phase, if the rule still wants hypotheses beyond this one (its subsets were drawn
            // beside this phase's EPnP blocks)
            const int next = a.phase_base + a.phase_cap;
            int more = 0;
            if (next < niters && n > model_points) more = niters - next < kPhaseHyps ? niters - next : kPhaseHyps;
            st->phase_hyps = more;
            st->next_base = more > 0 ? next : -1;
        }
    }
    __syncthreads();
    for (int i = lane; i < kPhaseHyps; i += 64) counts[i] = 0;          // ready for the next phase's atomics
}

// One LM evaluation over the masked points by a 256-thread workgroup: err norm^2, optionally
// J^T J (upper, 21) and J^T e (6).  Partial sums per thread (points tid, tid + 256, ...), wave
// butterfly, then the four wave sums added in wave order by every thread: all threads hold
// bit-identical totals and take the same branches.
constexpr int kRefitThreads = 256;
__device__ inline double lm_eval(const double param[6], const float *X3, const float2 *img,
                                 const uint8_t *mask, int n, double fx, double fy, double cx, double cy,
                                 double *red /* LDS: 4 x 28 */, double *JtJ, double *JtErr)
{
    double R[9], dRdr[27];
    rodrigues_vec2mat_d(param, R, JtJ ? dRdr : nullptr);
    const double *t = param + 3;
    double acc[28];
    const int nacc = JtJ ? 28 : 1;
    for (int k = 0; k < nacc; k++) acc[k] = 0;
    for (int i = threadIdx.x; i < n; i += kRefitThreads) {
        if (!mask[i]) continue;
        double X = X3[3 * i], Y = X3[3 * i + 1], Z = X3[3 * i + 2];
        double x = R[0] * X + R[1] * Y + R[2] * Z + t[0];
        double y = R[3] * X + R[4] * Y + R[5] * Z + t[1];
        double z = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        z = z ? 1. / z : 1;
        x *= z; y *= z;
        const float2 m = img[i];
        double ex = (x * fx + cx) - (double)m.x, ey = (y * fy + cy) - (double)m.y;
        acc[0] += ex * ex + ey * ey;
        if (JtJ) {
            double Jx[6], Jy[6];
            double dxdt[3] = {z, 0, -x * z}, dydt[3] = {0, z, -y * z};
            double dx0dr[3] = {X * dRdr[0] + Y * dRdr[1] + Z * dRdr[2], X * dRdr[9] + Y * dRdr[10] + Z * dRdr[11],
                               X * dRdr[18] + Y * dRdr[19] + Z * dRdr[20]};
            double dy0dr[3] = {X * dRdr[3] + Y * dRdr[4] + Z * dRdr[5], X * dRdr[12] + Y * dRdr[13] + Z * dRdr[14],
                               X * dRdr[21] + Y * dRdr[22] + Z * dRdr[23]};
            double dz0dr[3] = {X * dRdr[6] + Y * dRdr[7] + Z * dRdr[8], X * dRdr[15] + Y * dRdr[16] + Z * dRdr[17],
                               X * dRdr[24] + Y * dRdr[25] + Z * dRdr[26]};
            for (int j = 0; j < 3; j++) {
                double dxdr = z * (dx0dr[j] - x * dz0dr[j]);
                double dydr = z * (dy0dr[j] - y * dz0dr[j]);
                Jx[j] = fx * dxdr; Jy[j] = fy * dydr;
                Jx[3 + j] = fx * dxdt[j]; Jy[3 + j] = fy * dydt[j];
            }
            int q = 1;
            for (int r = 0; r < 6; r++)
                for (int c = r; c < 6; c++) acc[q++] += Jx[r] * Jx[c] + Jy[r] * Jy[c];
            for (int r = 0; r < 6; r++) acc[q++] += Jx[r] * ex + Jy[r] * ey;
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                                        // `red` may still be read from the previous call
    for (int k = 0; k < nacc; k++) {
        const double w = wave_allsum_f64(acc[k]);
        if (lane == 0) red[wave * 28 + k] = w;
    }
    __syncthreads();
    for (int k = 0; k < nacc; k++) acc[k] = ((red[k] + red[28 + k]) + red[56 + k]) + red[84 + k];
    if (JtJ) {
        int q = 1;
        for (int r = 0; r < 6; r++)
            for (int c = r; c < 6; c++) { JtJ[r * 6 + c] = acc[q]; JtJ[c * 6 + r] = acc[q]; q++; }
        for (int r = 0; r < 6; r++) JtErr[r] = acc[q++];
    }
    return acc[0];
}

constexpr size_t kRefitLdsBytes = (size_t)(72 * 64 + 4 * 28 + 8 + 6 * 64) * sizeof(double);
__global__ __launch_bounds__(kRefitThreads) void pnp_refit_kernel(PnpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double pnp_smem[];
    double *ws = pnp_smem;                    // lane-interleaved 6x6 SVD workspace of wave 0 (36,864 B)
    double *red = pnp_smem + 72 * 64;         // 4 x 28 wave sums
    double *bcast = red + 4 * 28;             // 6 parameters + flags handed from wave 0 to the others
    double *wl = bcast + 8;                   // lane-interleaved singular values of the cooperative SVD
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const PnpState *st = a.state + b;
    const int n = st->n;
    const float *X3 = a.X3 + (int64_t)b * a.stride * 3;
    const float2 *img = a.img + (int64_t)b * a.stride;
    uint8_t *mask = a.mask + (int64_t)b * a.stride;
    PnpRecord *out = a.out + b;
    const int model_points = n == 4 ? 4 : 5;
    const int max_good = st->max_good;
    if (n < model_points || max_good <= 0) {
        // fewer than four points: cv::solvePnPRansac would assert; no solution.  max_good == 0:
        // solvePnPRansac returns false, rvec/tvec stay at the caller's zeros, no inliers
        if (tid == 0) {
            for (int i = 0; i < 3; i++) { out->rvec[i] = 0; out->tvec[i] = 0; }
            for (int i = 0; i < 9; i++) out->R[i] = (i % 4 == 0) ? 1.0 : 0.0;
            out->n_inliers = 0; out->ransac_iters = n < model_points ? 0 : st->iters_done; out->best_iter = -1;
            out->lm_iters = 0; out->ok = 0; out->n = n;
        }
        for (int i = tid; i < n; i += kRefitThreads) mask[i] = 0;
        return;
    }
    const double fx = a.fx, fy = a.fy, cx = a.cx, cy = a.cy;
    const float thr2 = (float)((double)a.reproj_err * (double)a.reproj_err);
    double bR[9], bt[3];
    for (int i = 0; i < 9; i++) bR[i] = st->bestRt[i];
    for (int i = 0; i < 3; i++) bt[i] = st->bestRt[9 + i];
    for (int i = tid; i < n; i += kRefitThreads) {
        const float2 m = img[i];
        mask[i] = (n == model_points) ? 1
                  : (uint8_t)(reproj_err2_d(bR, bt, fx, fy, cx, cy, X3[3 * i], X3[3 * i + 1], X3[3 * i + 2], m.x, m.y) <= thr2);
    }
    __threadfence_block();
    __syncthreads();

    // ---- refit on the inliers: solvePnP(ITERATIVE, useExtrinsicGuess) == CvLevMarq on 6 params.
    // Every thread carries the same scalars; the 6x6 SVD solves run on wave 0 (lane-interleaved LDS
    // workspace, all its lanes redundantly) and the new parameters reach the other waves through LDS.
    double param[6], prevParam[6], JtJ[36], JtErr[6];
    if (wave == 0) {
        rodrigues_mat2vec_d(bR, param, ws + lane, 64);
        if (lane == 0) for (int i = 0; i < 3; i++) bcast[i] = param[i];
    }
    __syncthreads();
    for (int i = 0; i < 3; i++) param[i] = bcast[i];
    param[3] = bt[0]; param[4] = bt[1]; param[5] = bt[2];
    const double POW10[33] = {1e-16, 1e-15, 1e-14, 1e-13, 1e-12, 1e-11, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6,
                              1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8,
                              1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16};
    const int max_iter = 20;
    const double epsilon = 1.1920928955078125e-07;    // FLT_EPSILON
    double prevErrNorm = 1.7976931348623157e308, errNorm;
    int lambdaLg10 = -3, iters = 0;
    for (;;) {
        double e2 = lm_eval(param, X3, img, mask, n, fx, fy, cx, cy, red, JtJ, JtErr);
        for (int i = 0; i < 6; i++) prevParam[i] = param[i];
        if (iters == 0) prevErrNorm = sqrt(e2);
        bool done = false;
        for (;;) {
            const double lambda = POW10[lambdaLg10 + 16];
            {
                // The damped normal equations.  CvLevMarq solves them with cv::solve(DECOMP_SVD); J^T J (1 +
                // lambda on the diagonal) is symmetric positive definite unless the inliers are degenerate,
                // and then a register-only Cholesky solve (every thread the same values, no LDS, no
                // barrier) gives the same step to within cond(A) * 2^-52 -- far inside what the LM
                // iteration corrects anyway (the refit is compared with the oracle at 1e-9, DESIGN.md
                // section 5).  The Jacobi SVD route (~80 k cycles per solve, two thirds of this kernel)
                // remains for the degenerate case, where its singular-value threshold matters.
                double A[36], x[6];
                for (int i = 0; i < 36; i++) A[i] = JtJ[i];
                for (int i = 0; i < 6; i++) A[i * 6 + i] *= 1. + lambda;
                if (!a.refit_svd && chol_solve6_d(A, JtErr, x)) {
                    for (int i = 0; i < 6; i++) param[i] = prevParam[i] - x[i];
                } else {
                    __syncthreads();                         // bcast is free again
                    svd_solve_coop_d<6, 6, 4>(A, JtErr, x, ws + lane, 64, wl + lane, wave);
                    if (tid == 0) for (int i = 0; i < 6; i++) bcast[i] = prevParam[i] - x[i];
                    __syncthreads();
                    for (int i = 0; i < 6; i++) param[i] = bcast[i];
                }
            }
            errNorm = sqrt(lm_eval(param, X3, img, mask, n, fx, fy, cx, cy, red, nullptr, nullptr));
            if (errNorm > prevErrNorm) {
                if (++lambdaLg10 <= 16) continue;
            }
            lambdaLg10 = lambdaLg10 - 1 > -16 ? lambdaLg10 - 1 : -16;
            double dn = 0, pn = 0;
            for (int i = 0; i < 6; i++) {
                dn += (param[i] - prevParam[i]) * (param[i] - prevParam[i]);
                pn += prevParam[i] * prevParam[i];
            }
            if (++iters >= max_iter || sqrt(dn) / sqrt(pn) < epsilon) done = true;
            prevErrNorm = errNorm;
            break;
        }
        if (done) break;
    }
    if (tid == 0) {
        double Rf[9];
        rodrigues_vec2mat_d(param, Rf, nullptr);
        for (int i = 0; i < 3; i++) { out->rvec[i] = param[i]; out->tvec[i] = param[3 + i]; }
        for (int i = 0; i < 9; i++) out->R[i] = Rf[i];
        out->n_inliers = max_good; out->ransac_iters = st->iters_done; out->best_iter = st->best_iter;
        out->lm_iters = iters; out->ok = 1; out->n = n;
    }
}

// ------------------------------------------------------------------------------------------
struct FinalizeArgs {
    const PnpRecord *pnp; const int *n_prev, *n_cur, *n_tracked;   // per pair
    const int *ovf;                // per pair (ORB mode): an image of the pair overflowed a capacity; may be null
    int n_pairs;
    int mode;                      // SVO_MODE_LK checks "< 30 FAST corners" first (src/tracking.cpp:261)
    int cap;                       // max_keypoints: a frame with more corners was truncated -> SVO_FAIL_CAPACITY
    int num_features_tracking; double inlier_rate, min_move2, max_move2;
    svo_step_result *res;
};

__device__ inline void finalize_pair(const FinalizeArgs &a, int p)
{
    svo_step_result r;
    const PnpRecord &q = a.pnp[p];
    r.n_prev_kps = a.n_prev[p]; r.n_cur_kps = a.n_cur[p];
    r.n_tracked = 0; r.n_inliers = 0; r.ransac_iters = 0; r.lm_iters = 0;
    for (int i = 0; i < 3; i++) { r.rvec[i] = 0; r.tvec[i] = 0; }
    for (int i = 0; i < 9; i++) r.R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 16; i++) { r.T_rel_inv[i] = (i % 5 == 0) ? 1.0 : 0.0; r.pose[i] = 0; }
    int fail = 0;
    if (r.n_prev_kps > a.cap || r.n_cur_kps > a.cap || (a.ovf && a.ovf[p])) fail = SVO_FAIL_CAPACITY;   // never silently track a truncated set
    else if (a.mode == SVO_MODE_LK && r.n_cur_kps < 30) fail = SVO_FAIL_FEW_KEYPOINTS;     // src/tracking.cpp:261
    else {
        const int m = a.n_tracked[p];
        r.n_tracked = m;
        if (m < a.num_features_tracking) fail = SVO_FAIL_FEW_TRACKS;         // :274
        else {
            r.n_inliers = q.n_inliers; r.ransac_iters = q.ransac_iters; r.lm_iters = q.lm_iters;
            for (int i = 0; i < 3; i++) { r.rvec[i] = q.rvec[i]; r.tvec[i] = q.tvec[i]; }
            for (int i = 0; i < 9; i++) r.R[i] = q.R[i];
            if ((double)q.n_inliers / (double)m < a.inlier_rate) fail = SVO_FAIL_INLIER_RATIO;   // :491
            else {
                // rotationMatrixToEulerAngles (:440-463): floats holding double expressions
                const double *R = q.R;
                const float sy = (float)sqrt(R[0] * R[0] + R[3] * R[3]);
                const bool singular = (double)sy < 1e-6;
                float ex, ey, ez;
                if (!singular) {
                    ex = (float)atan2(R[7], R[8]); ey = (float)atan2(-R[6], (double)sy); ez = (float)atan2(R[3], R[0]);
                } else {
                    ex = (float)atan2(-R[5], R[4]); ey = (float)atan2(-R[6], (double)sy); ez = 0.f;
                }
                if (!((double)fabsf(ey) < 0.1 && (double)fabsf(ex) < 0.1 && (double)fabsf(ez) < 0.1))
                    fail = SVO_FAIL_ROTATION_GATE;                           // :308
                else {
                    const double *t = q.tvec;
                    const double n2 = t[0] * t[0] + t[1] * t[1] + t[2] * t[2];
                    if (!(n2 < a.max_move2 && n2 > a.min_move2)) fail = SVO_FAIL_TRANSL_GATE;   // :311
                    else {
                        double *Ti = r.T_rel_inv;      // inv([R t; 0 1]) in closed form
                        Ti[0] = R[0]; Ti[1] = R[3]; Ti[2] = R[6];
                        Ti[4] = R[1]; Ti[5] = R[4]; Ti[6] = R[7];
                        Ti[8] = R[2]; Ti[9] = R[5]; Ti[10] = R[8];
                        Ti[3] = -(R[0] * t[0] + R[3] * t[1] + R[6] * t[2]);
                        Ti[7] = -(R[1] * t[0] + R[4] * t[1] + R[7] * t[2]);
                        Ti[11] = -(R[2] * t[0] + R[5] * t[1] + R[8] * t[2]);
                        Ti[12] = 0; Ti[13] = 0; Ti[14] = 0; Ti[15] = 1;
                    }
                }
            }
        }
    }
    r.fail_stage = fail;
    r.ok = fail == 0;
    a.res[p] = r;
}

// frame_pose_ = frame_pose_ * T^-1 over consecutive pairs; failed steps are skipped (:59-68).
// The seed pose travels BY VALUE in the kernel argument block: a launch queued behind many others
// (callers of svo_track_batch with device results never synchronise) keeps the seed it was given.
// One wave: lanes (i, j) hold P[i][j], row elements travel by quad DPP broadcast (see
// chain_relative_kernel below: same association order -- k ascending, separate multiply and add --
// as a plain triple loop, so chunked and whole-sequence chaining agree bit for bit).
struct Pose16 { double m[16]; };
template <int K>
__device__ __forceinline__ double quad_bcast_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), K * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), K * 0x55, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// The gates of every pair, then the chain, in ONE launch of one workgroup (the pose stage ends with it: two launches were
// 10 us of a lone pair's 750).  ONE WAVE (round 6): the launch is queued on the side stream while the NEXT batch's LK grid
// fills the chip; behind lk_sse2_kernel's single-wave workgroups (two a SIMD, all 160 KB of LDS) wave slots come free one at
// a time, and a four-wave workgroup never found four at once until that grid drained -- the poses of a batch arrived up to a
// whole LK launch (20 ms) late (profiles/r06_sse2_timeline_before.csv).  A lone wave takes the first slot that frees; the 256
// pairs of a batch are four trips of finalize_pair instead of one.
constexpr int kFinThreads = 64;
__global__ __launch_bounds__(kFinThreads) void finalize_chain_kernel(FinalizeArgs a, Pose16 pose0, const double *seed_dev)
{
    __shared__ double sT[64 * 16];
    __shared__ int sOk[64];
    for (int p = threadIdx.x; p < a.n_pairs; p += kFinThreads) finalize_pair(a, p);
    __threadfence_block();
    svo_step_result *res = a.res;
    const int n_pairs = a.n_pairs;
    const int lane = threadIdx.x & 63, i = (lane >> 2) & 3, j = lane & 3;
    const bool first = true;
    double P = seed_dev ? seed_dev[i * 4 + j] : pose0.m[i * 4 + j];     // device seed: the previous batch's last pose
    for (int base = 0; base < n_pairs; base += 64) {
        // the records of 64 pairs are fetched together (one dependent global load per pair made the
        // serial product cost a memory round trip per step)
        const int cnt = min(64, n_pairs - base);
        __syncthreads();
        if (first) {
            for (int e = lane; e < cnt * 16; e += 64) sT[e] = res[base + (e >> 4)].T_rel_inv[e & 15];
            if (lane < cnt) sOk[lane] = res[base + lane].ok;
        }
        __syncthreads();
        if (!first) continue;
        for (int p = 0; p < cnt; p++) {
            const double p0 = quad_bcast_f64<0>(P), p1 = quad_bcast_f64<1>(P), p2 = quad_bcast_f64<2>(P), p3 = quad_bcast_f64<3>(P);
            if (sOk[p]) {
                const double *t = sT + p * 16 + j;
                double s = 0;
                s += p0 * t[0]; s += p1 * t[4]; s += p2 * t[8]; s += p3 * t[12];
                P = s;
            }
            if (lane < 16) res[base + p].pose[lane] = P;
        }
    }
}

// The same product for relative motions gathered from independently tracked chunks (other
// launches, contexts or GPUs): lanes (i, j) of one wave hold P[i][j]; row elements travel by quad
// broadcast, the T matrices are staged through LDS 64 pairs at a time.  Same association order
// as chain_kernel (k ascending, separate multiply and add).

__global__ __launch_bounds__(64) void chain_relative_kernel(const double *T, const int *ok, int n, Pose16 pose0,
                                                            double *out)
{
    __shared__ double sT[64 * 16];
    __shared__ int sOk[64];
    const int lane = threadIdx.x, i = (lane >> 2) & 3, j = lane & 3;
    double P = pose0.m[i * 4 + j];
    for (int base = 0; base < n; base += 64) {
        const int cnt = min(64, n - base);
        __syncthreads();
        for (int e = lane; e < cnt * 16; e += 64) sT[e] = T[(int64_t)base * 16 + e];
        if (lane < cnt) sOk[lane] = ok[base + lane];
        __syncthreads();
        for (int p = 0; p < cnt; p++) {
            const double p0 = quad_bcast_f64<0>(P), p1 = quad_bcast_f64<1>(P), p2 = quad_bcast_f64<2>(P), p3 = quad_bcast_f64<3>(P);
            if (sOk[p]) {
                const double *t = sT + p * 16 + j;
                double s = 0;
                s += p0 * t[0]; s += p1 * t[4]; s += p2 * t[8]; s += p3 * t[12];
                P = s;
            }
            if (lane < 16) out[(int64_t)(base + p) * 16 + lane] = P;
        }
    }
}

int stage_chain_relative(svo_ctx *ctx, const double *T, const int32_t *ok, int n, const double *pose0_host, double *out,
                         int mem)
{
    SVO_ARG(T && ok && out && n >= 0, "null pointer / negative count");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    if (n == 0) return SVO_OK;
    Pose16 p0;                                     // travels as a kernel argument: no copy, no allocation
    for (int q = 0; q < 16; q++) p0.m[q] = pose0_host ? pose0_host[q] : (q % 5 == 0 ? 1.0 : 0.0);
    if (mem == SVO_MEM_DEVICE) {                   // stream-ordered, nothing else to do
        hipLaunchKernelGGL(chain_relative_kernel, dim3(1), dim3(64), 0, ctx->stream, T, (const int *)ok, n, p0, out);
        SVO_HIP(hipGetLastError());
        return SVO_OK;
    }
    // host operands: temporaries released on every exit path
    struct DevBuf {
        void *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } bT, bOut, bOk;
    SVO_HIP(hipMalloc(&bT.p, sizeof(double) * 16 * (size_t)n));
    SVO_HIP(hipMalloc(&bOut.p, sizeof(double) * 16 * (size_t)n));
    SVO_HIP(hipMalloc(&bOk.p, sizeof(int) * (size_t)n));
    SVO_HIP(hipMemcpyAsync(bT.p, T, sizeof(double) * 16 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipMemcpyAsync(bOk.p, ok, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(chain_relative_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double *)bT.p, (const int *)bOk.p, n, p0,
                       (double *)bOut.p);
    SVO_HIP(hipGetLastError());
    SVO_HIP(hipMemcpyAsync(out, bOut.p, sizeof(double) * 16 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

// ------------------------------------------------------------------------------------------
// workspace layout inside ctx->pnp_ws (n = max_batch items):
//   [PnpRecord x n][mask bytes x n*cap][PnpState x n][PnpHyp x n*512][counts x n*512][subsets x n*2*512*5]
//   [EPnP hand-over records x n * 8 blocks * 105 * 64 doubles][cv::RNG(-1) states x kRngStream]
// (kPhaseHyps = 512 hypotheses per phase in kPhaseBlocks = 8 blocks of 64.)  The hand-over records are the large part:
// 430 KB per item, 110 MB for a 256-pair context, in BOTH modes -- a later phase of an LK-mode solve launches all eight
// blocks too.  INTEGRATION.md lists a context's device memory.
static size_t al256(size_t v) { return (v + 255) / 256 * 256; }
static size_t ws_off_mask(int n_items) { return al256(sizeof(PnpRecord) * (size_t)n_items); }
static size_t ws_off_state(const svo_config &cfg, int n_items) { return al256(ws_off_mask(n_items) + (size_t)n_items * cfg.max_keypoints); }
static size_t ws_off_hyp(const svo_config &cfg, int n_items) { return al256(ws_off_state(cfg, n_items) + sizeof(PnpState) * (size_t)n_items); }
static size_t ws_off_counts(const svo_config &cfg, int n_items) { return al256(ws_off_hyp(cfg, n_items) + sizeof(PnpHyp) * (size_t)n_items * kPhaseHyps); }
static size_t ws_off_subsets(const svo_config &cfg, int n_items) { return al256(ws_off_counts(cfg, n_items) + sizeof(int) * (size_t)n_items * kPhaseHyps); }
static size_t ws_off_hand(const svo_config &cfg, int n_items) { return al256(ws_off_subsets(cfg, n_items) + sizeof(int) * 2 * 5 * (size_t)n_items * kPhaseHyps); }
static size_t ws_off_stream(const svo_config &cfg, int n_items) { return al256(ws_off_hand(cfg, n_items) + sizeof(double) * (size_t)n_items * kPhaseBlocks * kEpnpHandDoubles * 64); }
static size_t ws_end(const svo_config &cfg, int n_items) { return al256(ws_off_stream(cfg, n_items) + sizeof(uint64_t) * kRngStream); }

// RANSAC inlier flags of batch item 0 (the online pair): max_keypoints bytes
const uint8_t *pnp_inlier_mask(const svo_ctx *ctx) { return (const uint8_t *)ctx->pnp_ws + ws_off_mask(ctx->cfg.max_batch); }

int geom_workspace_bytes(const svo_config &cfg, int n_items, size_t *bytes)
{
    // called once per context at creation: the EPnP and refit kernels need more dynamic LDS than the default limit
    if (hipFuncSetAttribute((const void *)pnp_hyp_front_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPnpFrontLdsBytes) != hipSuccess ||
        hipFuncSetAttribute((const void *)pnp_hyp_front_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPnpFrontLdsBytes) != hipSuccess)
        return SVO_ERR_HIP;
    if (hipFuncSetAttribute((const void *)pnp_hyp_back_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kPnpLdsBytes) != hipSuccess)
        return SVO_ERR_HIP;
    if (hipFuncSetAttribute((const void *)pnp_refit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)kRefitLdsBytes) != hipSuccess)
        return SVO_ERR_HIP;
    *bytes = ws_end(cfg, n_items) + 256;
    return SVO_OK;
}

// the generator states every solve starts with (cv::RNG(-1): state = (uint32)state * 4164903690 + (state >> 32)), once per context
int geom_workspace_init(svo_ctx *ctx)
{
    std::vector<uint64_t> st(kRngStream);
    uint64_t x = ~0ull;
    for (int i = 0; i < kRngStream; i++) { x = (uint64_t)(uint32_t)x * 4164903690u + (uint32_t)(x >> 32); st[i] = x; }
    if (hipMemcpy((char *)ctx->pnp_ws + ws_off_stream(ctx->cfg, ctx->cfg.max_batch), st.data(), sizeof(uint64_t) * kRngStream,
                  hipMemcpyHostToDevice) != hipSuccess)
        return SVO_ERR_HIP;
    return SVO_OK;
}

// The launch sequence of one solvePnPRansac for n_items items (see the kernel comments above).
// max_pts bounds the points of any item (grid.z of the scoring kernel).
static void launch_pnp_pipeline(svo_ctx *ctx, PnpArgs a, int n_items, int max_pts, hipStream_t st, bool begun = false)
{
    const svo_config &cfg = ctx->cfg;
    char *ws = (char *)ctx->pnp_ws;
    const int B = cfg.max_batch;
    a.out = (PnpRecord *)ws;
    a.state = (PnpState *)(ws + ws_off_state(cfg, B));
    a.hyp = (PnpHyp *)(ws + ws_off_hyp(cfg, B));
    a.counts = (int *)(ws + ws_off_counts(cfg, B));
    a.subsets = (int *)(ws + ws_off_subsets(cfg, B));
    a.hand = (double *)(ws + ws_off_hand(cfg, B));
    a.stream = (const uint64_t *)(ws + ws_off_stream(cfg, B));
    if (!begun) hipLaunchKernelGGL(pnp_begin_kernel, dim3(n_items), dim3(64), 0, st, a);
    const int niters = a.iterations > 1 ? a.iterations : 1;
    // 1024 points per scoring workgroup; a launch that leaves the chip mostly empty (the online path) takes
    // 256 (one staging pass, four times the workgroups: 36 -> 12 us for a pair's 2.4 k points)
    a.score_chunk = n_items <= 4 ? 256 : kScoreChunk;
    const int zchunks = max_pts > 0 ? (max_pts + a.score_chunk - 1) / a.score_chunk : 1;
    int phase = 0;
    for (int base = 0; base < niters; phase++) {
        const int cap = base == 0 ? a.first_cap : kPhaseHyps;
        const int hyps = niters - base < cap ? niters - base : cap;
        const int blocks = (hyps + kHypBlock - 1) / kHypBlock;
        a.phase_base = base; a.phase_cap = cap; a.phase_index = phase;
        // one block per CU (94 KB of LDS, one wave per SIMD); beside the next batch's front end that leaves the
        // LDS-staged front-end kernels room on the CU (two 78 KB blocks locked them out: ORB mode, 1.7 ms per step)
        // (ORB-mode batches only: beside lk_kernel the lighter build costs the LK launch 0.1 ms of shared vector-unit time)
        if (n_items >= 16 && ctx->cfg.track_mode == SVO_MODE_ORB) hipLaunchKernelGGL(pnp_hyp_front_kernel<3>, dim3(blocks + 1, n_items), dim3(256), kPnpFrontLdsBytes, st, a);   // + the drawer
        else hipLaunchKernelGGL(pnp_hyp_front_kernel<2>, dim3(blocks + 1, n_items), dim3(256), kPnpFrontLdsBytes, st, a);
        hipLaunchKernelGGL(pnp_hyp_back_kernel, dim3(blocks, n_items), dim3(256), kPnpLdsBytes, st, a);
        hipLaunchKernelGGL(pnp_score_kernel, dim3(blocks, n_items, zchunks), dim3(256), 0, st, a);
        hipLaunchKernelGGL(pnp_select_kernel, dim3(n_items), dim3(64), 0, st, a);
        base += cap;
    }
    const char *force = getenv("SVO_REFIT_SVD");           // test hook: the rank-deficient route on ordinary data
    a.refit_svd = force && force[0] == '1';
    hipLaunchKernelGGL(pnp_refit_kernel, dim3(n_items), dim3(kRefitThreads), kRefitLdsBytes, st, a);
}

// The count snapshot alone (the part of pnp_begin that must run in FRONT-END order: the next batch's FAST overwrites the live
// counts): for a triangulation launched on the side stream.
__global__ __launch_bounds__(256) void snap_counts_kernel(SnapArgs q)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= q.n_pairs) return;
    const int sp = q.per * (q.fp0 + b * q.fstep), sc = q.per * (q.fc0 + b * q.fstep);
    q.snap[b] = q.n[sp];
    q.snap[q.n_pairs + b] = q.n[sc];
    if (q.ovf) q.snap[2 * q.n_pairs + b] = q.per == 2 ? (q.ovf[sp] | q.ovf[sp + 1] | q.ovf[sc] | q.ovf[sc + 1]) : (q.ovf[sp] | q.ovf[sc]);
}
void launch_snap_counts(svo_ctx *ctx, int n_items, const SnapSpec &snap, hipStream_t st)
{
    const SnapArgs q{snap.n, snap.ovf, snap.fp0, snap.fc0, snap.fstep, snap.per, n_items, ctx->kp_n_snap};
    hipLaunchKernelGGL(snap_counts_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, q);
}

void launch_triangulate_batch(svo_ctx *ctx, int n_items, int max_pts, const float2 *x1, const float2 *x2,
                              const int *n_pts, int n_fixed, const SnapSpec *snap, hipStream_t st)
{
    if (!st) st = ctx->stream;
    TriArgs a{};
    memcpy(a.P1, ctx->cfg.P1, sizeof(a.P1));
    memcpy(a.P2, ctx->cfg.P2, sizeof(a.P2));
    a.x1 = x1; a.x2 = x2; a.out3 = ctx->X3; a.stride = ctx->cfg.max_keypoints;
    a.n_pts = n_pts; a.n_fixed = n_fixed;
    int gx = (8192 + n_items - 1) / n_items;                 // ~8 k workgroups per launch
    gx = gx < 2 ? 2 : gx;
    gx = gx > (max_pts + 63) / 64 ? (max_pts + 63) / 64 : gx;
    gx = gx < 0 ? 0 : gx;
    // + one workgroup per item that prepares the item's solvePnPRansac (launch_pnp_batch follows on every path)
    char *ws = (char *)ctx->pnp_ws;
    const int B = ctx->cfg.max_batch;
    a.fuse_begin = 1;
    a.begin.state = (PnpState *)(ws + ws_off_state(ctx->cfg, B));
    a.begin.counts = (int *)(ws + ws_off_counts(ctx->cfg, B));
    a.begin.subsets = (int *)(ws + ws_off_subsets(ctx->cfg, B));
    a.begin.iterations = ctx->cfg.iterations;
    a.begin.first_cap = pnp_first_cap(ctx->cfg);
    a.begin.stream = (const uint64_t *)(ws + ws_off_stream(ctx->cfg, B));
    if (snap) a.begin.snap = SnapArgs{snap->n, snap->ovf, snap->fp0, snap->fc0, snap->fstep, snap->per, n_items, ctx->kp_n_snap};
    hipLaunchKernelGGL(triangulate_kernel, dim3(gx + 1, n_items), dim3(64), 0, st, a);
}

void launch_pnp_batch(svo_ctx *ctx, int n_items, const float2 *img, const int *n_pts, int n_fixed, hipStream_t st)
{
    PnpArgs a{};
    a.X3 = ctx->X3; a.img = img; a.stride = ctx->cfg.max_keypoints;
    a.n_pts = n_pts; a.n_fixed = n_fixed;
    // K = P1[:, :3] (src/tracking.cpp:476-477)
    a.fx = ctx->cfg.P1[0]; a.fy = ctx->cfg.P1[5]; a.cx = ctx->cfg.P1[2]; a.cy = ctx->cfg.P1[6];
    a.iterations = ctx->cfg.iterations; a.reproj_err = ctx->cfg.reproj_err;
    a.first_cap = pnp_first_cap(ctx->cfg);
    a.confidence = (double)ctx->cfg.confidence;
    a.mask = (uint8_t *)ctx->pnp_ws + ws_off_mask(ctx->cfg.max_batch);
    launch_pnp_pipeline(ctx, a, n_items, ctx->cfg.max_keypoints, st, /*begun by launch_triangulate_batch*/ true);
}

void launch_finalize_chain(svo_ctx *ctx, int n_pairs, const int *n_prev, const int *n_cur, const int *ovf,
                           const double *pose0_host, hipStream_t st)
{
    FinalizeArgs f{};
    f.pnp = (const PnpRecord *)ctx->pnp_ws; f.n_prev = n_prev; f.n_cur = n_cur; f.n_tracked = ctx->m_out; f.ovf = ovf;
    f.cap = ctx->cfg.max_keypoints;
    f.n_pairs = n_pairs; f.mode = ctx->cfg.track_mode; f.num_features_tracking = ctx->cfg.num_features_tracking;
    f.inlier_rate = ctx->cfg.inlier_rate; f.min_move2 = ctx->cfg.min_move2; f.max_move2 = ctx->cfg.max_move2;
    f.res = ctx->d_results;
    Pose16 p0;
    for (int i = 0; i < 16; i++) p0.m[i] = pose0_host ? pose0_host[i] : ((i % 5 == 0) ? 1.0 : 0.0);
    hipLaunchKernelGGL(finalize_chain_kernel, dim3(1), dim3(kFinThreads), 0, st, f, p0, ctx->seed_dev);
}

// ---- stage API ------------------------------------------------------------------------------
int stage_triangulate(svo_ctx *ctx, const double P1[12], const double P2[12], const svo_pt2f *x1,
                      const svo_pt2f *x2, int n, svo_pt3f *out, int mem)
{
    SVO_ARG(P1 && P2, "null projection matrix");
    SVO_ARG(n >= 0 && n <= ctx->cfg.max_keypoints, "n exceeds max_keypoints");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    if (n == 0) return SVO_OK;
    SVO_ARG(x1 && x2 && out, "null pointer");
    SVO_HIP(hipSetDevice(ctx->device));
    TriArgs a{};
    memcpy(a.P1, P1, sizeof(a.P1));
    memcpy(a.P2, P2, sizeof(a.P2));
    a.stride = 0; a.n_pts = nullptr; a.n_fixed = n;
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(ctx->cmp[0], x1, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(ctx->cmp[1], x2, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        a.x1 = ctx->cmp[0]; a.x2 = ctx->cmp[1]; a.out3 = ctx->X3;
    } else {
        a.x1 = (const float2 *)x1; a.x2 = (const float2 *)x2; a.out3 = (float *)out;
    }
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 63) / 64, 1), dim3(64), 0, ctx->stream, a);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(out, ctx->X3, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

int stage_pnp_ransac(svo_ctx *ctx, const svo_pt3f *obj, const svo_pt2f *img, int n, const double K[9],
                     int iterations, float reproj_err, double confidence, svo_pnp_result *res,
                     uint8_t *inlier_mask, int mem)
{
    SVO_ARG(K && res, "null pointer");
    SVO_ARG(n >= 0 && n <= ctx->cfg.max_keypoints, "n exceeds max_keypoints");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    SVO_ARG(n == 0 || (obj && img), "null pointer");
    SVO_HIP(hipSetDevice(ctx->device));
    PnpArgs a{};
    a.stride = 0; a.n_pts = nullptr; a.n_fixed = n;
    a.fx = K[0]; a.fy = K[4]; a.cx = K[2]; a.cy = K[5];
    a.iterations = iterations; a.reproj_err = reproj_err; a.confidence = confidence;
    a.first_cap = pnp_first_cap(ctx->cfg);                   // (the stage call of a context follows its track_mode too)
    uint8_t *ws_mask = (uint8_t *)ctx->pnp_ws + ws_off_mask(ctx->cfg.max_batch);
    if (mem == SVO_MEM_HOST) {
        if (n > 0) {
            SVO_HIP(hipMemcpyAsync(ctx->X3, obj, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
            SVO_HIP(hipMemcpyAsync(ctx->cmp[3], img, sizeof(float2) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
        }
        a.X3 = ctx->X3; a.img = ctx->cmp[3]; a.mask = ws_mask;
    } else {
        a.X3 = (const float *)obj; a.img = (const float2 *)img; a.mask = inlier_mask ? inlier_mask : ws_mask;
    }
    launch_pnp_pipeline(ctx, a, 1, n, ctx->stream);
    SVO_HIP(hipGetLastError());
    PnpRecord *h = (PnpRecord *)((char *)ctx->h_pinned + 256);
    SVO_HIP(hipMemcpyAsync(h, ctx->pnp_ws, sizeof(PnpRecord), hipMemcpyDeviceToHost, ctx->stream));
    if (mem == SVO_MEM_HOST && inlier_mask && n > 0)
        SVO_HIP(hipMemcpyAsync(inlier_mask, ws_mask, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(res->rvec, h->rvec, sizeof(res->rvec));
    memcpy(res->tvec, h->tvec, sizeof(res->tvec));
    memcpy(res->R, h->R, sizeof(res->R));
    res->n_inliers = h->n_inliers; res->ransac_iters = h->ransac_iters; res->best_iter = h->best_iter;
    res->lm_iters = h->lm_iters; res->ok = h->ok; res->_pad = 0;
    return SVO_OK;
}

}  // namespace svo

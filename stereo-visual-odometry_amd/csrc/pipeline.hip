// pipeline.hip -- the fused LK-mode frame step (Tracking::AddFrame -> LK_StereoF2F_PnP_Track,
// reference src/tracking.cpp:49-77, 258-344) as a fixed sequence of batched launches:
//
//   pyramids (L,R of every new frame) -> FAST on every left image -> circular LK chain per point
//   -> stable compaction -> triangulation -> RANSAC-EPnP + LM -> gates -> pose chain
//
// Every consecutive frame pair is independent, so a batch of F frames is
// F-1 pairs processed by ONE launch of each kernel (grid.y / grid.x = pair), with frame f's
// pyramids at bslots[2f], bslots[2f+1] and its keypoints at kp[f]; only the 4x4 pose product is
// sequential.  The online entry point (svo_add_frame) runs the same kernels with one pair on a
// two-frame ring.
#include <cstring>
#include "svo_ctx.h"

namespace svo {

void timing_mark(svo_ctx *ctx, const char *name)
{
    if (!ctx->timing) return;
    if (ctx->ev_used >= 16384) return;                 // bounded log
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t ev = nullptr;
        if (hipEventCreate(&ev) != hipSuccess) return;
        ctx->ev_pool.push_back(ev);
    }
    hipEvent_t ev = ctx->ev_pool[ctx->ev_used++];
    (void)hipEventRecord(ev, ctx->stream);
    ctx->marks.emplace_back(name, ev);
}
static inline void mark(svo_ctx *ctx, const char *name) { timing_mark(ctx, name); }

static const char *kTMatch = "orb_match";
static const char *kT0 = "start", *kTPyr = "pyramid", *kTFast = "fast", *kTLk = "lk", *kTCompact = "compact",
                  *kTTri = "triangulate", *kTPnp = "pnp", *kTFin = "finalize";

// Builds pyramids of `n_new` frames into frame slots [f0, f0+n_new) and runs FAST on their left
// images.  L/R: device pointers to the first new frame.
static int ingest_frames(svo_ctx *ctx, const uint8_t *L, const uint8_t *R, int pitch, int64_t frame_stride,
                         int f0, int n_new)
{
    if (ctx->cfg.track_mode == SVO_MODE_ORB) {
        // Detect_MyORBFeatures (src/tracking.cpp:502-532): ORBextractor on the left AND right image;
        // frame f -> feature slots 2f (left), 2f+1 (right)
        // (orb_extract_batch records its own stage marks: orb_pyramid, orb_cellfast, orb_quadtree, orb_describe)
        // (level 0 read in place: the frames belong to the caller's batch / the context's staging until the step is done)
        return orb_extract_batch(ctx, L, R, pitch, frame_stride, 2 * f0, 2 * n_new, ctx->stream, /*in_place*/ true);
    }
    const PyrGeom &g = ctx->geom;
    PyrArgs p{};
    // left and right images interleave into consecutive slots 2f, 2f+1: one launch set for both
    p.g = g; p.pitch = pitch; p.img_stride = frame_stride; p.slot_stride = g.slot_bytes;
    p.img = L; p.img2 = R; p.slots = ctx->bslots + (size_t)(2 * f0) * g.slot_bytes;
    launch_pyramid(p, 2 * n_new, ctx->stream);
    mark(ctx, kTPyr);
    FastArgs a{};
    a.img = L; a.pitch = pitch; a.img_stride = frame_stride;
    a.w = ctx->cfg.width; a.h = ctx->cfg.height; a.thr = ctx->cfg.fast_threshold; a.nms = 1;
    a.score = ctx->score + (size_t)f0 * ctx->score_stride; a.spitch = ctx->spitch; a.score_stride = ctx->score_stride;
    a.rowcount = ctx->rowcount + (size_t)f0 * ctx->rowcount_stride; a.rowcount_stride = ctx->rowcount_stride;
    a.kp_xy = ctx->kp_xy + (size_t)f0 * ctx->cfg.max_keypoints;
    a.kp_resp = ctx->kp_resp + (size_t)f0 * ctx->cfg.max_keypoints;
    a.kp_stride = ctx->cfg.max_keypoints;
    a.n_out = ctx->kp_n + f0; a.cap = ctx->cfg.max_keypoints;
    launch_fast(a, n_new, ctx->stream);
    launch_fast_keep_strongest(a, n_new, ctx->cfg.fast_keep_strongest, ctx->stream);
    mark(ctx, kTFast);
    return SVO_OK;
}

static int run_back(svo_ctx *ctx, int n_pairs, const double *pose0_host, svo_step_result *results_dev, bool triangulate_first = false);

// A micro-batch of a stream starts with the frame the previous one ended with: instead of building that frame's
// pyramids and detecting its features again, what the pair needs of it is carried from frame slot `last` to slot 0
// (LK mode: the two pyramid slots, the FAST keypoints + responses + count; ORB mode: both images' keypoints, descriptors,
// counts and capacity flags) -- ONE launch, 16 bytes per thread.
struct CarryArgs { uint8_t *dst[4]; const uint8_t *src[4]; size_t bytes[4]; int n; };
__global__ __launch_bounds__(256) void carry_frame_kernel(CarryArgs a)
{
    const int seg = blockIdx.y;
    if (seg >= a.n) return;
    const size_t n16 = a.bytes[seg] / 16, rest = a.bytes[seg] - n16 * 16;
    const uint4 *s = (const uint4 *)a.src[seg];
    uint4 *d = (uint4 *)a.dst[seg];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) d[i] = s[i];
    if (blockIdx.x == 0 && threadIdx.x < rest) a.dst[seg][n16 * 16 + threadIdx.x] = a.src[seg][n16 * 16 + threadIdx.x];
}

static void carry_last_frame(svo_ctx *ctx, int last)
{
    CarryArgs c{};
    const size_t cap = (size_t)ctx->cfg.max_keypoints;
    auto seg = [&](void *dst, const void *src, size_t bytes) {
        c.dst[c.n] = (uint8_t *)dst; c.src[c.n] = (const uint8_t *)src; c.bytes[c.n] = bytes; c.n++;
    };
    if (ctx->cfg.track_mode == SVO_MODE_ORB) {
        const size_t kcap = (size_t)ctx->orb_kp_cap;
        seg(ctx->orb_kps, (const svo_keypoint *)ctx->orb_kps + 2 * (size_t)last * kcap, 2 * kcap * sizeof(svo_keypoint));
        seg(ctx->orb_desc, ctx->orb_desc + 2 * (size_t)last * kcap * 32, 2 * kcap * 32);
        seg(ctx->orb_n, ctx->orb_n + 2 * last, 2 * sizeof(int));
        seg(ctx->orb_overflow, ctx->orb_overflow + 2 * last, 2 * sizeof(int));
    } else {
        seg(ctx->bslots, ctx->bslots + (size_t)(2 * last) * ctx->geom.slot_bytes, (size_t)2 * ctx->geom.slot_bytes);
        seg(ctx->kp_xy, ctx->kp_xy + (size_t)last * cap, cap * sizeof(float2));
        seg(ctx->kp_resp, ctx->kp_resp + (size_t)last * cap, cap * sizeof(float));
        seg(ctx->kp_n, ctx->kp_n + last, sizeof(int));
    }
    hipLaunchKernelGGL(carry_frame_kernel, dim3(64, c.n), dim3(256), 0, ctx->stream, c);
}

// Tracks `n_pairs` pairs; pair p = (frame slot fp0 + p*fstep, frame slot fc0 + p*fstep).
// Front half on the context's stream: circular LK, compaction, triangulation.  Back half (pose
// solver, gates, chain, optional copy of the records to `results_dev`) on `back_stream`, which is
// the context's stream, or -- overlap mode -- the side stream, ordered after the front by an event.
static int run_pairs(svo_ctx *ctx, int n_pairs, int fp0, int fc0, int fstep, const double *pose0_host,
                     svo_step_result *results_dev)
{
    const PyrGeom &g = ctx->geom;
    const int cap = ctx->cfg.max_keypoints;
    if (ctx->cfg.track_mode == SVO_MODE_ORB) {
        // ORB_StereoF2F_PnP_Track (src/tracking.cpp:168-249): Hamming matches + filter instead of LK
        if (ctx->back_pending) {
            SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_back, 0));
            ctx->back_pending = false;
        }
        orb_match_pairs(ctx, n_pairs, fp0, fc0, fstep, ctx->stream);
        mark(ctx, kTMatch);
        // + n_prev / n_cur = left keypoint counts of the two frames (feature slots 2f) and the capacity flags, frozen for the
        // pose stage by the pairs' begin workgroups
        const SnapSpec snap{(const int *)ctx->orb_n, (const int *)ctx->orb_overflow, fp0, fc0, fstep, 2};
        launch_triangulate_batch(ctx, n_pairs, cap, ctx->cmp[0], ctx->cmp[1], ctx->m_out, 0, &snap);
        mark(ctx, kTTri);
        return run_back(ctx, n_pairs, pose0_host, results_dev);
    }
    auto S = [&](int slot) { return ctx->bslots + (size_t)slot * g.slot_bytes; };
    LkArgs a{};
    a.g = g; a.ncalls = 4;
    a.slot_stride = (int64_t)fstep * 2 * g.slot_bytes;
    // L1 -> R1 -> R2 -> L2 -> L1'  (src/tracking.cpp:593-618)
    a.prev[0] = S(2 * fp0);     a.next[0] = S(2 * fp0 + 1);
    a.prev[1] = S(2 * fp0 + 1); a.next[1] = S(2 * fc0 + 1);
    a.prev[2] = S(2 * fc0 + 1); a.next[2] = S(2 * fc0);
    a.prev[3] = S(2 * fc0);     a.next[3] = S(2 * fp0);
    // matched_t1_left = the previous frame's FAST keypoints (:268-271), read in place.  Outputs use
    // the same per-item stride (po = b * pts_stride + idx); n_pts is indexed by the batch item, so
    // fstep must be 1 when n_pairs > 1.
    a.pts_in = ctx->kp_xy + (size_t)fp0 * cap; a.pts_stride = (int64_t)fstep * cap;
    a.n_pts = ctx->kp_n + fp0;
    a.n_fixed = 0; a.cap = cap;
    for (int i = 0; i < 4; i++) { a.pts_out[i] = ctx->pts_out[i]; a.status[i] = ctx->status[i]; }
    a.keep = ctx->keep;
    a.match_err = ctx->cfg.feature_match_error; a.match_err_f = (float)ctx->cfg.feature_match_error;
    a.accum = ctx->cfg.lk_accum;
    launch_lk(a, n_pairs, cap, ctx->stream);
    mark(ctx, kTLk);
    // the previous batch's pose stage (overlap mode) still reads the compacted lists / 3-D points
    if (ctx->back_pending) {
        SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_back, 0));
        ctx->back_pending = false;
    }
    CompactArgs c{};
    c.keep = ctx->keep; c.n_pts = ctx->kp_n + fp0; c.n_fixed = 0; c.pts_stride = (int64_t)fstep * cap; c.cap = cap;
    c.in[0] = a.pts_in; c.in[1] = ctx->pts_out[0]; c.in[2] = ctx->pts_out[1]; c.in[3] = ctx->pts_out[2];
    for (int i = 0; i < 4; i++) c.out[i] = ctx->cmp[i];
    c.m_out = ctx->m_out;
    launch_compact(c, n_pairs, ctx->stream);
    mark(ctx, kTCompact);
    // triangulatePoints(P1, P2, t1_left, t1_right) (:292-294); the pairs' begin workgroups also freeze the keypoint counts
    // of the frames involved for the pose stage (FAST of the next batch overwrites kp_n): snap[p] = n_prev of pair p,
    // snap[n_pairs + p] = n_cur of pair p
    const SnapSpec snap{ctx->kp_n, nullptr, fp0, fc0, fstep, 1};
    // Overlap mode: the triangulation belongs to the pose stage (nothing of the front end reads its points), so it goes to the
    // side stream with it -- 0.14 ms per 256 pairs, 50 us of a micro-batch's front-end chain --; only the count snapshot stays in
    // front-end order (SVO_TRI_SIDE=0: the round-5 order, for A/B runs)
    static const bool tri_side = !(getenv("SVO_TRI_SIDE") && getenv("SVO_TRI_SIDE")[0] == '0');
    if (tri_side && ctx->overlap && results_dev != nullptr) {
        launch_snap_counts(ctx, n_pairs, snap, ctx->stream);
        mark(ctx, kTTri);
        return run_back(ctx, n_pairs, pose0_host, results_dev, /*triangulate_first*/ true);
    }
    launch_triangulate_batch(ctx, n_pairs, cap, ctx->cmp[0], ctx->cmp[1], ctx->m_out, 0, &snap);
    mark(ctx, kTTri);
    return run_back(ctx, n_pairs, pose0_host, results_dev);
}

// Pose stage: solvePnPRansac(X, t2_left) (:299 / :200), gates, frame_pose_ chain, optional copy of the
// records; on the side stream in overlap mode.
static int run_back(svo_ctx *ctx, int n_pairs, const double *pose0_host, svo_step_result *results_dev, bool triangulate_first)
{
    hipStream_t bs = ctx->stream;
    const bool side = ctx->overlap && results_dev != nullptr;
    if (side) {
        SVO_HIP(hipEventRecord(ctx->ev_front, ctx->stream));
        SVO_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_front, 0));
        bs = ctx->side_stream;
    }
    if (triangulate_first)       // LK mode, overlap: triangulatePoints + the RANSAC start-up of every pair, on the pose stage's stream
        launch_triangulate_batch(ctx, n_pairs, ctx->cfg.max_keypoints, ctx->cmp[0], ctx->cmp[1], ctx->m_out, 0, nullptr, bs);
    launch_pnp_batch(ctx, n_pairs, ctx->cmp[3], ctx->m_out, 0, bs);
    if (!side) mark(ctx, kTPnp);
    launch_finalize_chain(ctx, n_pairs, ctx->kp_n_snap, ctx->kp_n_snap + n_pairs,
                          ctx->cfg.track_mode == SVO_MODE_ORB ? ctx->kp_n_snap + 2 * n_pairs : nullptr, pose0_host, bs);
    if (results_dev)
        SVO_HIP(hipMemcpyAsync(results_dev, ctx->d_results, sizeof(svo_step_result) * (size_t)n_pairs,
                               hipMemcpyDeviceToDevice, bs));
    if (side) {
        SVO_HIP(hipEventRecord(ctx->ev_back, ctx->side_stream));
        ctx->back_pending = true;
    } else {
        mark(ctx, kTFin);
    }
    return SVO_OK;
}

int pipeline_track_batch(svo_ctx *ctx, const uint8_t *left_frames, const uint8_t *right_frames, int pitch,
                         int64_t frame_stride, int n_frames, const double *pose0,
                         svo_step_result *results, int results_mem, int carry_first)
{
    // results == NULL with SVO_MEM_DEVICE: the records stay in the context (svo_collect_results)
    SVO_ARG(left_frames && right_frames && (results || results_mem == SVO_MEM_DEVICE), "null pointer");
    SVO_ARG(n_frames >= 2 && n_frames - 1 <= ctx->cfg.max_batch, "n_frames - 1 must be in [1, max_batch]");
    SVO_ARG(pitch >= ctx->cfg.width && frame_stride >= (int64_t)pitch * ctx->cfg.height, "bad pitch / frame_stride");
    SVO_ARG(results_mem == SVO_MEM_HOST || results_mem == SVO_MEM_DEVICE, "bad results_mem");
    SVO_HIP(hipSetDevice(ctx->device));
    const int n_pairs = n_frames - 1;
    // carry_first: frame 0 of this batch IS the last frame of the previous batch on this context (a stream's halo frame):
    // its features are carried over instead of being computed again
    SVO_ARG(!carry_first || ctx->carry_slot > 0, "SVO_CONTINUE_CARRY_FRAME: frame slot of the previous async batch's last frame is no longer valid "
                                                 "(no such batch, a failed launch, or svo_add_frame / a synchronous batch ran in between)");
    const int carry_from = carry_first ? ctx->carry_slot : 0;
    ctx->carry_slot = -1;                        // whatever happens below overwrites frame slots
    ctx->last_batch_pairs = n_pairs;
    mark(ctx, kT0);
    int rc;
    if (carry_from > 0) {
        carry_last_frame(ctx, carry_from);
        rc = ingest_frames(ctx, left_frames + frame_stride, right_frames + frame_stride, pitch, frame_stride, 1, n_frames - 1);
    } else {
        rc = ingest_frames(ctx, left_frames, right_frames, pitch, frame_stride, 0, n_frames);
    }
    if (rc) return rc;
    // the LK outputs use the keypoint stride (cap) per item: frame slots are consecutive (fstep 1)
    rc = run_pairs(ctx, n_pairs, 0, 1, 1, pose0, results_mem == SVO_MEM_DEVICE ? results : nullptr);
    if (rc) return rc;
    SVO_HIP(hipGetLastError());
    if (results_mem == SVO_MEM_DEVICE) return SVO_OK;
    svo_step_result *h = (svo_step_result *)((char *)ctx->h_pinned + 4096);
    SVO_HIP(hipMemcpyAsync(h, ctx->d_results, sizeof(svo_step_result) * (size_t)n_pairs, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(results, h, sizeof(svo_step_result) * (size_t)n_pairs);
    return SVO_OK;
}

// Host image -> device staging: the rows are gathered into a pinned mirror with the staging pitch
// and go over PCIe as ONE copy (a pitched 2-D copy from pageable memory is issued row by row and
// costs ~5 ms per KITTI pair).
int stage_host_image(svo_ctx *ctx, const uint8_t *img, int pitch, int stage_idx, const uint8_t **dptr, int *dpitch)
{
    const int w = ctx->cfg.width, h = ctx->cfg.height, sp = ctx->stage_pitch;
    const size_t bytes = (size_t)sp * h;
    uint8_t *hs = ctx->h_stage + (size_t)stage_idx * bytes;
    uint8_t *dst = ctx->stage_img + (size_t)stage_idx * bytes;
    if (ctx->h_stage_busy[stage_idx]) {
        SVO_HIP(hipEventSynchronize(ctx->ev_stage[stage_idx]));
        ctx->h_stage_busy[stage_idx] = false;
    }
    for (int y = 0; y < h; y++) memcpy(hs + (size_t)y * sp, img + (size_t)y * pitch, (size_t)w);
    SVO_HIP(hipMemcpyAsync(dst, hs, bytes, hipMemcpyHostToDevice, ctx->stream));
    SVO_HIP(hipEventRecord(ctx->ev_stage[stage_idx], ctx->stream));
    ctx->h_stage_busy[stage_idx] = true;
    *dptr = dst; *dpitch = sp;
    return SVO_OK;
}

int pipeline_add_frame(svo_ctx *ctx, const uint8_t *left, const uint8_t *right, int pitch, int mem,
                       svo_step_result *res)
{
    SVO_ARG(left && right && res, "null pointer");
    SVO_ARG(pitch >= ctx->cfg.width, "pitch < width");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    SVO_HIP(hipSetDevice(ctx->device));
    ctx->carry_slot = -1;                        // the online ring lives in frame slots 0 / 1
    const uint8_t *dL = left, *dR = right;
    int dp = pitch;
    if (mem == SVO_MEM_HOST) {
        int rcs = stage_host_image(ctx, left, pitch, 0, &dL, &dp);
        if (rcs) return rcs;
        rcs = stage_host_image(ctx, right, pitch, 1, &dR, &dp);
        if (rcs) return rcs;
    }
    // two-frame ring in frame slots 0 / 1
    const int cur = ctx->online_frames == 0 ? 0 : (ctx->online_cur ^ 1);
    const int prev = cur ^ 1;
    mark(ctx, kT0);
    int rc = ingest_frames(ctx, dL, dR, dp, 0, cur, 1);
    if (rc) return rc;
    memset(res, 0, sizeof(*res));
    if (ctx->online_frames == 0) {
        // StereoInit_f2f (:78-92): detect only
        int *h_n = (int *)ctx->h_pinned;
        const int *src_n = ctx->cfg.track_mode == SVO_MODE_ORB ? ctx->orb_n + 2 * cur : ctx->kp_n + cur;
        SVO_HIP(hipMemcpyAsync(h_n, src_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
        res->ok = 1; res->n_cur_kps = *h_n;
        for (int i = 0; i < 9; i++) res->R[i] = (i % 4 == 0) ? 1.0 : 0.0;
        for (int i = 0; i < 16; i++) { res->T_rel_inv[i] = (i % 5 == 0) ? 1.0 : 0.0; res->pose[i] = ctx->pose[i]; }
        ctx->online_frames = 1; ctx->online_cur = cur; ctx->online_tracked = 0;
        return SVO_OK;
    }
    rc = run_pairs(ctx, 1, prev, cur, 0, ctx->pose, nullptr);
    if (rc) return rc;
    SVO_HIP(hipGetLastError());
    svo_step_result *h = (svo_step_result *)((char *)ctx->h_pinned + 4096);
    SVO_HIP(hipMemcpyAsync(h, ctx->d_results, sizeof(svo_step_result), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    *res = *h;
    ctx->online_tracked = res->n_tracked;
    memcpy(ctx->pose, res->pose, sizeof(ctx->pose));
    ctx->online_frames++; ctx->online_cur = cur;       // last_frame_ = current_frame_ on both outcomes (:59-68)
    return res->ok ? SVO_OK : res->fail_stage;
}

}  // namespace svo

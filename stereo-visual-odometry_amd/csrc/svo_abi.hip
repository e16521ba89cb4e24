// svo_abi.hip -- host side of the C-ABI declared in include/svo_abi.h: context lifecycle, buffer
// ownership (ctx owns every device buffer; no allocation in the steady state) and the stage /
// fused entry points that launch the HIP kernels.  There is NO CPU fallback in this library.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <new>
#include <vector>
#include "svo_ctx.h"

using namespace svo;

namespace svo {
// geometry.hip / pipeline.hip
int geom_workspace_bytes(const svo_config &cfg, int n_items, size_t *bytes);
int geom_workspace_init(svo_ctx *ctx);
int stage_triangulate(svo_ctx *ctx, const double P1[12], const double P2[12], const svo_pt2f *x1,
                      const svo_pt2f *x2, int n, svo_pt3f *out, int mem);
int stage_pnp_ransac(svo_ctx *ctx, const svo_pt3f *obj, const svo_pt2f *img, int n, const double K[9],
                     int iterations, float reproj_err, double confidence, svo_pnp_result *res,
                     uint8_t *inlier_mask, int mem);
int pipeline_add_frame(svo_ctx *ctx, const uint8_t *left, const uint8_t *right, int pitch, int mem,
                       svo_step_result *res);
int pipeline_track_batch(svo_ctx *ctx, const uint8_t *left_frames, const uint8_t *right_frames, int pitch,
                         int64_t frame_stride, int n_frames, const double *pose0,
                         svo_step_result *results, int results_mem, int carry_first = 0);
}  // namespace svo

static int align_up(int v, int a) { return (v + a - 1) / a * a; }

namespace svo {
int dev_alloc_raw(svo_ctx *ctx, void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    if (ctx->arena.planning) {
        ctx->arena.plan.emplace_back(out, ctx->arena.planned);
        ctx->arena.planned += (bytes + 255) / 256 * 256;
        return SVO_OK;
    }
    SVO_HIP(hipMalloc(out, bytes));
    ctx->arena.extra.push_back(*out);
    return SVO_OK;
}

int dev_defer(svo_ctx *ctx, std::function<int()> fn)
{
    if (!ctx->arena.planning) return fn();
    ctx->arena.after.push_back(std::move(fn));
    return SVO_OK;
}

static int dev_commit(svo_ctx *ctx)
{
    DevArena &a = ctx->arena;
    a.planning = false;
    SVO_HIP(hipMalloc((void **)&a.base, a.planned + 256));
    for (auto &e : a.plan) *e.first = a.base + e.second;
    a.plan.clear();
    for (auto &fn : a.after) { const int rc = fn(); if (rc != SVO_OK) return rc; }
    a.after.clear();
    return SVO_OK;
}

static void dev_release(svo_ctx *ctx)
{
    for (void *p : ctx->arena.extra) if (p) (void)hipFree(p);
    ctx->arena.extra.clear();
    if (ctx->arena.base) (void)hipFree(ctx->arena.base);
    ctx->arena.base = nullptr;
}
}  // namespace svo

// Pyramid geometry: buildOpticalFlowPyramid stops adding levels once the next one would not
// exceed the 21-pixel window ("if (sz.width <= winSize.width || sz.height <= winSize.height)").
static void make_geom(int w, int h, PyrGeom *g)
{
    memset(g, 0, sizeof(*g));
    int64_t off = 0;
    int lw = w, lh = h;
    for (int l = 0; l < kMaxLevels; l++) {
        if (l > 0) {
            int nw = (lw + 1) / 2, nh = (lh + 1) / 2;
            if (nw <= kWin || nh <= kWin) break;
            lw = nw; lh = nh;
        }
        g->w[l] = lw; g->h[l] = lh;
        g->pitch[l] = align_up(lw + 2 * kPad, 64);
        g->origin[l] = off + (int64_t)kPad * g->pitch[l] + kPad;
        off += (int64_t)g->pitch[l] * (lh + 2 * kPad);
        off = (off + 255) / 256 * 256;
        g->nlevels = l + 1;
    }
    g->slot_bytes = off;
}

extern "C" int svo_wait_results(svo_ctx *ctx);
extern "C" int svo_abi_version(void) { return SVO_ABI_VERSION; }
extern "C" int svo_config_bytes(void) { return (int)sizeof(svo_config); }

extern "C" int svo_device_count(int *n)
{
    if (!n) return SVO_ERR_ARG;
    *n = 0;
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) return SVO_ERR_HIP;
    *n = c;
    return SVO_OK;
}

extern "C" void svo_default_config(svo_config *cfg, int width, int height)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->width = width; cfg->height = height;
    cfg->max_keypoints = 8192;
    cfg->max_batch = 1;
    cfg->num_slots = 4;
    cfg->fast_threshold = 20;               // src/tracking.cpp:99
    cfg->num_features_tracking = 5;         // config/default.yaml:69
    cfg->iterations = 500;                  // :80
    cfg->reproj_err = 0.5f;                 // :81
    cfg->confidence = 0.99f;                // :82
    cfg->feature_match_error = 3.0;         // :66
    cfg->inlier_rate = 0.01;                // :77
    cfg->min_move2 = 0.0005 * 0.0005;       // LK mode, src/tracking.cpp:311
    cfg->max_move2 = 100.0;
    const double fx = 718.856, fy = 718.856, cx = 607.193, cy = 185.216, tx = -0.537;   // default.yaml:33-47
    const double P1[12] = {fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0};
    const double P2[12] = {fx, 0, cx, fx * tx, 0, fy, cy, 0, 0, 0, 1, 0};
    memcpy(cfg->P1, P1, sizeof(P1));
    memcpy(cfg->P2, P2, sizeof(P2));
    cfg->track_mode = SVO_MODE_LK;
    cfg->orb_nfeatures = 2000;              // config/default.yaml:93
    cfg->orb_scale_factor = 1.2f;           // :92
    cfg->orb_nlevels = 8;                   // :91
    cfg->orb_ini_th = 20;                   // :90
    cfg->orb_min_th = 7;                    // :89
    cfg->fast_keep_strongest = 0;           // every cv::FAST corner, as the reference tracks them
    cfg->lk_accum = SVO_LK_ACCUM_EXACT;     // the canonical recipe; SVO_LK_ACCUM_SSE2 = an x86 OpenCV build's float order
}

static void free_all(svo_ctx *c)
{
    dev_release(c);                                   // every device buffer: the arena + the lazy extras
    if (c->ev_front) (void)hipEventDestroy(c->ev_front);
    if (c->ev_back) (void)hipEventDestroy(c->ev_back);
    if (c->ev_order) (void)hipEventDestroy(c->ev_order);
    if (c->side_stream) (void)hipStreamDestroy(c->side_stream);
    for (int k = 0; k < 2; k++) {
        if (c->ev_up[k]) (void)hipEventDestroy(c->ev_up[k]);
        if (c->ev_fb_free[k]) (void)hipEventDestroy(c->ev_fb_free[k]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    for (int k = 0; k < 2; k++)
        if (c->ev_async[k]) (void)hipEventDestroy(c->ev_async[k]);
    if (c->fetch_stream) (void)hipStreamDestroy(c->fetch_stream);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (int k = 0; k < 2; k++) if (c->ev_stage[k]) (void)hipEventDestroy(c->ev_stage[k]);
    for (auto &e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
}

// SVO_TIMING=1: milliseconds of svo_create's phases on stderr (where a short run's start-up goes)
static double now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
#define PHASE(name) do { if (timing_on) { const double t__ = now_ms(); fprintf(stderr, "[svo_create] %8.2f ms  %s\n", t__ - t_phase, name); t_phase = t__; } } while (0)

extern "C" int svo_create(const svo_config *cfg, int device, svo_ctx **out)
{
    if (!cfg || !out) return SVO_ERR_ARG;
    const bool timing_on = getenv("SVO_TIMING") != nullptr;
    double t_phase = now_ms();
    *out = nullptr;
    if (cfg->width < 32 || cfg->height < 32 || cfg->max_keypoints < 64 || cfg->max_batch < 1 ||
        cfg->num_slots < 4 || cfg->width > 16384 || cfg->height > 16384)
        return SVO_ERR_ARG;
    if (cfg->lk_accum != SVO_LK_ACCUM_EXACT && cfg->lk_accum != SVO_LK_ACCUM_SSE2 && cfg->lk_accum != SVO_LK_ACCUM_SIMD128 &&
        cfg->lk_accum != SVO_LK_ACCUM_SSE2_LEGACY) {
        fprintf(stderr, "svo_create: lk_accum = %d is none of SVO_LK_ACCUM_EXACT / _SSE2 / _SIMD128 / _SSE2_LEGACY\n", cfg->lk_accum);
        return SVO_ERR_ARG;
    }
    if (cfg->fast_keep_strongest < 0) return SVO_ERR_ARG;
    if (cfg->num_features_tracking < 4) {
        // a pair with exactly 4 tracks takes cv::solvePnPRansac's npoints == 4 branch (the P3P kernel: geometry.hip); with
        // fewer than 4 the reference's call asserts (CV_Assert(npoints >= 4)) and the process dies: refuse the configuration
        fprintf(stderr, "svo_create: num_features_tracking = %d < 4 is not supported (cv::solvePnPRansac asserts npoints >= 4)\n",
                cfg->num_features_tracking);
        return SVO_ERR_ARG;
    }
    svo_ctx *ctx = new (std::nothrow) svo_ctx();
    if (!ctx) return SVO_ERR_NOMEM;
    ctx->cfg = *cfg;
    ctx->device = device;
    for (int i = 0; i < 16; i++) ctx->pose[i] = (i % 5 == 0) ? 1.0 : 0.0;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        // fail loudly: this library has no CPU path
        fprintf(stderr, "svo_create: no usable HIP device (count=%d, requested %d)\n", ndev, device);
        delete ctx;
        return SVO_ERR_HIP;
    }
    PHASE("hipGetDeviceCount (HIP runtime start)");
    auto fail = [&](int code) { free_all(ctx); delete ctx; return code; };
#define CK(call) do { if ((call) != hipSuccess) { fprintf(stderr, "svo_create: %s failed\n", #call); return fail(SVO_ERR_HIP); } } while (0)
    CK(hipSetDevice(device));
    CK(hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
    ctx->stream = ctx->own_stream;
    PHASE("hipSetDevice + stream");
    make_geom(cfg->width, cfg->height, &ctx->geom);

    const int w = cfg->width, h = cfg->height, cap = cfg->max_keypoints, B = cfg->max_batch;
    const int n_img = B + 1;
    ctx->n_img = n_img;
    const bool orb = cfg->track_mode == SVO_MODE_ORB;
    if (!orb && cfg->track_mode != SVO_MODE_LK) return fail(SVO_ERR_ARG);
    // What only the LK pipeline touches is sized for ONE item in an ORB context (the stage API -- svo_fast_detect,
    // svo_lk_track, svo_circular_match -- stays usable there); the ORB buffers of an LK context are allocated on first use.
    const int n_fast = orb ? 1 : n_img, B_lk = orb ? 1 : B;
    ctx->arena.planning = true;
#define DA(ptr, bytes) do { if (dev_alloc(ctx, &(ptr), (bytes)) != SVO_OK) return fail(SVO_ERR_HIP); } while (0)
    DA(ctx->slots, (size_t)cfg->num_slots * ctx->geom.slot_bytes);
    dev_defer(ctx, [ctx]() { SVO_HIP(hipMemsetAsync(ctx->slots, 0, (size_t)ctx->cfg.num_slots * ctx->geom.slot_bytes, ctx->stream)); return SVO_OK; });
    ctx->slot_built.assign(cfg->num_slots, 0);
    ctx->stage_pitch = align_up(w, 256);
    DA(ctx->stage_img, (size_t)ctx->stage_pitch * h * 2);
    ctx->spitch = align_up(w, 64);
    ctx->score_stride = (int64_t)ctx->spitch * h;
    DA(ctx->score, (size_t)ctx->score_stride * n_fast);
    ctx->rowcount_stride = align_up(h, 64);
    DA(ctx->rowcount, sizeof(int) * (size_t)ctx->rowcount_stride * n_fast);
    DA(ctx->kp_xy, sizeof(float2) * (size_t)cap * n_fast);
    DA(ctx->kp_resp, sizeof(float) * (size_t)cap * n_fast);
    DA(ctx->kp_n, sizeof(int) * (size_t)n_fast);
    DA(ctx->pts_in, sizeof(float2) * (size_t)cap * B_lk);
    for (int i = 0; i < 4; i++) {
        DA(ctx->pts_out[i], sizeof(float2) * (size_t)cap * B_lk);
        DA(ctx->status[i], (size_t)cap * B_lk);
        DA(ctx->cmp[i], sizeof(float2) * (size_t)cap * B);
    }
    DA(ctx->keep, (size_t)cap * B_lk);
    DA(ctx->m_out, sizeof(int) * (size_t)B);
    DA(ctx->X3, sizeof(float) * 3 * (size_t)cap * B);
    if (geom_workspace_bytes(*cfg, B, &ctx->pnp_ws_bytes) != SVO_OK) return fail(SVO_ERR_ARG);
    PHASE("plan + kernel attributes (code object load)");
    DA(ctx->pnp_ws, ctx->pnp_ws_bytes);
    dev_defer(ctx, [ctx]() { return geom_workspace_init(ctx); });
    DA(ctx->d_results, sizeof(svo_step_result) * (size_t)B);
    DA(ctx->bslots, orb ? 256 : (size_t)2 * n_img * ctx->geom.slot_bytes);
    DA(ctx->kp_n_snap, sizeof(int) * (size_t)(3 * n_img));
    if (B > 1) {
        // a batch context is what the runner and the stream create: their frame buffers and record rings come with it
        const size_t per_cam = (size_t)ctx->stage_pitch * h * (size_t)(B + 1);
        for (int k = 0; k < 2; k++) {
            DA(ctx->fb[k], 2 * per_cam);
            DA(ctx->d_async[k], sizeof(svo_step_result) * (size_t)B);
        }
    }
    if (orb) {
        int rc = orb_alloc(ctx);
        if (rc != SVO_OK) { fprintf(stderr, "svo_create: %s\n", ctx->err.c_str()); return fail(rc); }
    }
#undef DA
    if (dev_commit(ctx) != SVO_OK) { fprintf(stderr, "svo_create: device memory (%zu MB): %s\n", ctx->arena.planned >> 20, ctx->err.c_str()); return fail(SVO_ERR_HIP); }
    PHASE("one device allocation + deferred initialisation");
    CK(hipHostMalloc(&ctx->h_stage, (size_t)ctx->stage_pitch * h * 2, hipHostMallocDefault));
    for (int k = 0; k < 2; k++) CK(hipEventCreateWithFlags(&ctx->ev_stage[k], hipEventDisableTiming));
    // (a high-priority side stream was measured in round 6: no difference, 5.80 against 5.80 ms per ORB step and 14.32 against
    // 14.34 per LK step -- what holds a pose-stage kernel back beside the next front end is free LDS and wave slots on a CU, not
    // the queue's place at the dispatcher)
    CK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ctx->ev_back, hipEventDisableTiming));
    if (B > 1) {
        for (int k = 0; k < 2; k++) {
            CK(hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ctx->ev_fb_free[k], hipEventDisableTiming));
            CK(hipEventCreateWithFlags(&ctx->ev_async[k], hipEventDisableTiming));
        }
        CK(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        CK(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
        ctx->async_ready = true;
    }
    ctx->h_pinned_bytes = sizeof(svo_step_result) * (size_t)B + 4096 +
                          (size_t)cap * (sizeof(svo_keypoint) + 32) + sizeof(int) * 64;
    CK(hipHostMalloc(&ctx->h_pinned, ctx->h_pinned_bytes, hipHostMallocDefault));
    PHASE("events, streams, page-locked scratch");
    CK(hipStreamSynchronize(ctx->stream));
    PHASE("stream sync");
#undef CK
    *out = ctx;
    return SVO_OK;
}

extern "C" void svo_destroy(svo_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    free_all(ctx);
    delete ctx;
}

extern "C" const char *svo_last_error(const svo_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }

extern "C" int svo_set_stream(svo_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SVO_ERR_ARG;
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return SVO_OK;
}

extern "C" int svo_sync(svo_ctx *ctx)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->back_pending) { SVO_HIP(hipStreamSynchronize(ctx->side_stream)); ctx->back_pending = false; }
    return SVO_OK;
}

extern "C" int svo_set_overlap(svo_ctx *ctx, int on)
{
    if (!ctx) return SVO_ERR_ARG;
    int rc = svo_wait_results(ctx);
    if (rc) return rc;
    ctx->overlap = on != 0;
    return SVO_OK;
}

// Makes the context's stream wait (on the device, no host sync) for the pose stage of the last
// svo_track_batch when overlap mode runs it on the side stream.
extern "C" int svo_wait_results(svo_ctx *ctx)
{
    if (!ctx) return SVO_ERR_ARG;
    if (ctx->back_pending) {
        SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_back, 0));
        ctx->back_pending = false;
    }
    return SVO_OK;
}

// ABI v7: ordering against work on OTHER streams on the device, without a host synchronisation.
extern "C" int svo_wait_stream(svo_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SVO_ERR_ARG;
    hipStream_t other = (hipStream_t)hip_stream;
    if (other == ctx->stream) return SVO_OK;                     // same queue: already ordered
    SVO_HIP(hipSetDevice(ctx->device));
    if (!ctx->ev_order) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    SVO_HIP(hipEventRecord(ctx->ev_order, other));
    SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_order, 0));
    return SVO_OK;
}

extern "C" int svo_signal_stream(svo_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SVO_ERR_ARG;
    hipStream_t other = (hipStream_t)hip_stream;
    SVO_HIP(hipSetDevice(ctx->device));
    // the pose stage of an overlap-mode batch ends on the side stream: its event first (kept pending for the context's own waits)
    if (ctx->back_pending && other != ctx->side_stream) SVO_HIP(hipStreamWaitEvent(other, ctx->ev_back, 0));
    if (other == ctx->stream) return SVO_OK;
    if (!ctx->ev_order) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    SVO_HIP(hipEventRecord(ctx->ev_order, ctx->stream));
    SVO_HIP(hipStreamWaitEvent(other, ctx->ev_order, 0));
    return SVO_OK;
}

// ABI v9: the front end only (everything that reads the caller's frames is on ctx->stream; the pose stage of an overlap-mode
// batch, which reads context memory only, is on the side stream and is not waited for).
extern "C" int svo_signal_stream_inputs(svo_ctx *ctx, void *hip_stream)
{
    if (!ctx) return SVO_ERR_ARG;
    hipStream_t other = (hipStream_t)hip_stream;
    if (other == ctx->stream) return SVO_OK;
    SVO_HIP(hipSetDevice(ctx->device));
    if (!ctx->ev_order) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_order, hipEventDisableTiming));
    SVO_HIP(hipEventRecord(ctx->ev_order, ctx->stream));
    SVO_HIP(hipStreamWaitEvent(other, ctx->ev_order, 0));
    return SVO_OK;
}

extern "C" int svo_num_levels(const svo_ctx *ctx) { return ctx ? ctx->geom.nlevels : 0; }

extern "C" int svo_enable_timing(svo_ctx *ctx, int on)
{
    if (!ctx) return SVO_ERR_ARG;
    ctx->timing = on != 0;
    return SVO_OK;
}

// Resolves the stage marks logged since the last query: synchronises the stream, averages the
// elapsed time of each stage over the logged steps ("start" opens a step), clears the log.
extern "C" int svo_get_timing(svo_ctx *ctx, const char **names, float *ms, int cap)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<std::pair<const char *, double>> acc;
    std::vector<int> cnt;
    for (size_t i = 1; i < ctx->marks.size(); i++) {
        const char *nm = ctx->marks[i].first;
        if (strcmp(nm, "start") == 0) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, ctx->marks[i - 1].second, ctx->marks[i].second) != hipSuccess) continue;
        size_t k = 0;
        for (; k < acc.size(); k++) if (acc[k].first == nm) break;
        if (k == acc.size()) { acc.emplace_back(nm, 0.0); cnt.push_back(0); }
        acc[k].second += t; cnt[k]++;
    }
    ctx->marks.clear();
    ctx->ev_used = 0;
    ctx->last_times.clear();
    for (size_t k = 0; k < acc.size(); k++) ctx->last_times.emplace_back(acc[k].first, (float)(acc[k].second / cnt[k]));
    int n = 0;
    for (auto &t : ctx->last_times) {
        if (n >= cap) break;
        if (names) names[n] = t.first;
        if (ms) ms[n] = t.second;
        n++;
    }
    return n;
}

// Host images are copied into the context's staging buffer (aligned pitch); device images are
// used in place.
static int resolve_image(svo_ctx *ctx, const uint8_t *img, int pitch, int mem, int stage_idx,
                         const uint8_t **dptr, int *dpitch)
{
    SVO_ARG(img != nullptr, "null image");
    SVO_ARG(pitch >= ctx->cfg.width, "pitch < width");
    if (mem == SVO_MEM_DEVICE) { *dptr = img; *dpitch = pitch; return SVO_OK; }
    SVO_ARG(mem == SVO_MEM_HOST, "mem must be SVO_MEM_HOST or SVO_MEM_DEVICE");
    return svo::stage_host_image(ctx, img, pitch, stage_idx, dptr, dpitch);
}

extern "C" int svo_fast_detect(svo_ctx *ctx, const uint8_t *img, int pitch, int mem, int threshold,
                               int nonmax, svo_keypoint *out, int cap, int *n_out)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(out && n_out && cap >= 0, "null output");
    SVO_HIP(hipSetDevice(ctx->device));
    const uint8_t *d; int dp;
    int rc = resolve_image(ctx, img, pitch, mem, 0, &d, &dp);
    if (rc) return rc;
    FastArgs a{};
    a.img = d; a.pitch = dp; a.img_stride = 0;
    a.w = ctx->cfg.width; a.h = ctx->cfg.height;
    a.thr = threshold < 0 ? 0 : (threshold > 255 ? 255 : threshold);
    a.nms = nonmax != 0;
    a.score = ctx->score; a.spitch = ctx->spitch; a.score_stride = ctx->score_stride;
    a.rowcount = ctx->rowcount; a.rowcount_stride = ctx->rowcount_stride;
    a.kp_xy = ctx->kp_xy; a.kp_resp = ctx->kp_resp; a.kp_stride = ctx->cfg.max_keypoints;
    a.n_out = ctx->kp_n; a.cap = ctx->cfg.max_keypoints;
    launch_fast(a, 1, ctx->stream);
    SVO_HIP(hipGetLastError());
    // D2H through pinned scratch: count, then xy + response
    int *h_n = (int *)ctx->h_pinned;
    SVO_HIP(hipMemcpyAsync(h_n, ctx->kp_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    int n = *h_n;
    *n_out = n;
    if (n > ctx->cfg.max_keypoints) { ctx->err = "FAST: keypoints exceed max_keypoints"; return SVO_ERR_ARG; }
    if (n > cap) { ctx->err = "FAST: keypoints exceed caller capacity"; return SVO_ERR_ARG; }
    if (n == 0) return SVO_OK;
    float2 *h_xy = (float2 *)((char *)ctx->h_pinned + 256);
    float *h_r = (float *)(h_xy + ctx->cfg.max_keypoints);
    SVO_HIP(hipMemcpyAsync(h_xy, ctx->kp_xy, sizeof(float2) * n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(h_r, ctx->kp_resp, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    // cv::KeyPoint(pt, 7.f, -1, response, 0, -1); always a HOST array (it is an array of structs
    // for the caller's std::vector<cv::KeyPoint>)
    for (int i = 0; i < n; i++) {
        out[i].x = h_xy[i].x; out[i].y = h_xy[i].y; out[i].size = 7.f; out[i].angle = -1.f;
        out[i].response = h_r[i]; out[i].octave = 0; out[i].class_id = -1;
    }
    return SVO_OK;
}

extern "C" int svo_build_pyramid(svo_ctx *ctx, int slot, const uint8_t *img, int pitch, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(slot >= 0 && slot < ctx->cfg.num_slots, "slot out of range");
    SVO_HIP(hipSetDevice(ctx->device));
    const uint8_t *d; int dp;
    int rc = resolve_image(ctx, img, pitch, mem, 0, &d, &dp);
    if (rc) return rc;
    PyrArgs a{};
    a.g = ctx->geom; a.img = d; a.pitch = dp; a.img_stride = 0;
    a.slots = ctx->slots + (size_t)slot * ctx->geom.slot_bytes; a.slot_stride = 0;
    launch_pyramid(a, 1, ctx->stream);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST) SVO_HIP(hipStreamSynchronize(ctx->stream));   // staging buffer is reused
    ctx->slot_built[slot] = 1;
    return SVO_OK;
}

extern "C" int svo_read_pyramid_level(svo_ctx *ctx, int slot, int level, uint8_t *out, int out_pitch,
                                      int mem, int *w, int *h)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(slot >= 0 && slot < ctx->cfg.num_slots, "slot out of range");
    SVO_ARG(level >= 0 && level < ctx->geom.nlevels, "level out of range");
    if (!ctx->slot_built[slot]) { ctx->err = "pyramid slot not built"; return SVO_ERR_STATE; }
    const int lw = ctx->geom.w[level], lh = ctx->geom.h[level];
    if (w) *w = lw;
    if (h) *h = lh;
    if (!out) return SVO_OK;
    SVO_ARG(out_pitch >= lw, "out_pitch < level width");
    SVO_HIP(hipSetDevice(ctx->device));
    const uint8_t *s = ctx->slots + (size_t)slot * ctx->geom.slot_bytes;
    if (mem == SVO_MEM_DEVICE) {
        launch_pyr_read_level(ctx->geom, s, level, out, out_pitch, ctx->stream);
        SVO_HIP(hipGetLastError());
        return SVO_OK;
    }
    SVO_HIP(hipMemcpy2DAsync(out, out_pitch, s + ctx->geom.origin[level], ctx->geom.pitch[level], lw, lh,
                             hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

// copies n items in (host) or aliases (device)
template <typename T>
static int to_device(svo_ctx *ctx, const T *src, T *scratch, int n, int mem, const T **d)
{
    if (mem == SVO_MEM_DEVICE) { *d = src; return SVO_OK; }
    SVO_HIP(hipMemcpyAsync(scratch, src, sizeof(T) * (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    *d = scratch;
    return SVO_OK;
}

static void fill_lk_common(svo_ctx *ctx, LkArgs &a, int n)
{
    a.g = ctx->geom;
    a.slot_stride = 0;
    a.pts_stride = 0;
    a.n_pts = nullptr; a.n_fixed = n; a.cap = n;
    a.match_err = ctx->cfg.feature_match_error;
    a.match_err_f = (float)ctx->cfg.feature_match_error;
    a.accum = ctx->cfg.lk_accum;
}

extern "C" int svo_lk_track(svo_ctx *ctx, int slot_prev, int slot_next, const svo_pt2f *prev_pts, int n,
                            svo_pt2f *next_pts, uint8_t *status, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(slot_prev >= 0 && slot_prev < ctx->cfg.num_slots && slot_next >= 0 &&
            slot_next < ctx->cfg.num_slots, "slot out of range");
    SVO_ARG(n >= 0 && n <= ctx->cfg.max_keypoints, "n exceeds max_keypoints");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    if (!ctx->slot_built[slot_prev] || !ctx->slot_built[slot_next]) { ctx->err = "pyramid slot not built"; return SVO_ERR_STATE; }
    if (n == 0) return SVO_OK;
    SVO_ARG(prev_pts && next_pts && status, "null pointer");
    SVO_HIP(hipSetDevice(ctx->device));
    const float2 *d_in;
    int rc = to_device<float2>(ctx, (const float2 *)prev_pts, ctx->pts_in, n, mem, &d_in);
    if (rc) return rc;
    LkArgs a{};
    fill_lk_common(ctx, a, n);
    a.ncalls = 1;
    a.prev[0] = ctx->slots + (size_t)slot_prev * ctx->geom.slot_bytes;
    a.next[0] = ctx->slots + (size_t)slot_next * ctx->geom.slot_bytes;
    a.pts_in = d_in;
    a.pts_out[0] = mem == SVO_MEM_DEVICE ? (float2 *)next_pts : ctx->pts_out[0];
    a.status[0] = mem == SVO_MEM_DEVICE ? status : ctx->status[0];
    launch_lk(a, 1, n, ctx->stream);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(next_pts, ctx->pts_out[0], sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipMemcpyAsync(status, ctx->status[0], (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

extern "C" int svo_circular_match(svo_ctx *ctx, int slot_prevL, int slot_prevR, int slot_curL, int slot_curR,
                                  const svo_pt2f *t1_left, int n, svo_pt2f *out_t1_left,
                                  svo_pt2f *out_t1_right, svo_pt2f *out_t2_right, svo_pt2f *out_t2_left,
                                  int *m_out, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    const int sl[4] = {slot_prevL, slot_prevR, slot_curL, slot_curR};
    for (int i = 0; i < 4; i++) {
        SVO_ARG(sl[i] >= 0 && sl[i] < ctx->cfg.num_slots, "slot out of range");
        if (!ctx->slot_built[sl[i]]) { ctx->err = "pyramid slot not built"; return SVO_ERR_STATE; }
    }
    SVO_ARG(n >= 0 && n <= ctx->cfg.max_keypoints, "n exceeds max_keypoints");
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    SVO_ARG(m_out != nullptr, "null m_out");
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    *m_out = 0;
    if (n == 0) return SVO_OK;
    SVO_ARG(t1_left && out_t1_left && out_t1_right && out_t2_right && out_t2_left, "null pointer");
    SVO_HIP(hipSetDevice(ctx->device));
    const float2 *d_in;
    int rc = to_device<float2>(ctx, (const float2 *)t1_left, ctx->pts_in, n, mem, &d_in);
    if (rc) return rc;
    auto S = [&](int s) { return ctx->slots + (size_t)s * ctx->geom.slot_bytes; };
    LkArgs a{};
    fill_lk_common(ctx, a, n);
    a.ncalls = 4;
    // L1 -> R1 -> R2 -> L2 -> L1'   (src/tracking.cpp:593-618)
    a.prev[0] = S(slot_prevL); a.next[0] = S(slot_prevR);
    a.prev[1] = S(slot_prevR); a.next[1] = S(slot_curR);
    a.prev[2] = S(slot_curR);  a.next[2] = S(slot_curL);
    a.prev[3] = S(slot_curL);  a.next[3] = S(slot_prevL);
    a.pts_in = d_in;
    for (int i = 0; i < 4; i++) { a.pts_out[i] = ctx->pts_out[i]; a.status[i] = ctx->status[i]; }
    a.keep = ctx->keep;
    launch_lk(a, 1, n, ctx->stream);
    CompactArgs c{};
    c.keep = ctx->keep; c.n_pts = nullptr; c.n_fixed = n; c.pts_stride = 0; c.cap = n;
    c.in[0] = d_in; c.in[1] = ctx->pts_out[0]; c.in[2] = ctx->pts_out[1]; c.in[3] = ctx->pts_out[2];
    svo_pt2f *outs[4] = {out_t1_left, out_t1_right, out_t2_right, out_t2_left};
    for (int i = 0; i < 4; i++) c.out[i] = mem == SVO_MEM_DEVICE ? (float2 *)outs[i] : ctx->cmp[i];
    c.m_out = ctx->m_out;
    launch_compact(c, 1, ctx->stream);
    SVO_HIP(hipGetLastError());
    int *h_m = (int *)ctx->h_pinned;
    SVO_HIP(hipMemcpyAsync(h_m, ctx->m_out, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    *m_out = *h_m;
    if (mem == SVO_MEM_HOST && *h_m > 0) {
        for (int i = 0; i < 4; i++)
            SVO_HIP(hipMemcpyAsync(outs[i], ctx->cmp[i], sizeof(float2) * (size_t)*h_m, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    }
    return SVO_OK;
}

extern "C" int svo_orb_extract(svo_ctx *ctx, const uint8_t *img, int pitch, int mem, svo_keypoint *kps, uint8_t *desc,
                               int cap, int *n_out, int *per_level)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(kps && desc && n_out && cap >= 0, "null output");
    SVO_HIP(hipSetDevice(ctx->device));
    int rc = orb_alloc(ctx);
    if (rc) return rc;
    const uint8_t *d; int dp;
    rc = resolve_image(ctx, img, pitch, mem, 0, &d, &dp);
    if (rc) return rc;
    rc = orb_extract_batch(ctx, d, nullptr, dp, 0, 0, 1, ctx->stream);
    if (rc) return rc;
    SVO_HIP(hipGetLastError());
    int *h = (int *)ctx->h_pinned;        // [0] n, [1] overflow, [8..16) per-level counts, [32..) cell counts
    SVO_HIP(hipMemcpyAsync(h, ctx->orb_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(h + 1, ctx->orb_overflow, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(h + 8, ctx->orb_sel_cnt, sizeof(int) * ctx->orb_geom.nlevels, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const int n = h[0];
    *n_out = n;
    if (h[1]) {
        ctx->err = (h[1] & 2) ? "ORB: FAST candidates exceed the per-cell (256) or per-level (4 * max_keypoints) capacity"
                   : (h[1] & 1) ? "ORB quadtree node pool overflow" : "ORB: keypoints exceed max_keypoints";
        return SVO_ERR_ARG;
    }
    if (per_level) for (int l = 0; l < 8; l++) per_level[l] = l < ctx->orb_geom.nlevels ? h[8 + l] : 0;
    if (n > cap) { ctx->err = "ORB: keypoints exceed caller capacity"; return SVO_ERR_ARG; }
    if (n == 0) return SVO_OK;
    svo_keypoint *hk = (svo_keypoint *)((char *)ctx->h_pinned + 256);
    uint8_t *hd = (uint8_t *)(hk + ctx->orb_kp_cap);
    SVO_HIP(hipMemcpyAsync(hk, ctx->orb_kps, sizeof(svo_keypoint) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipMemcpyAsync(hd, ctx->orb_desc, (size_t)32 * n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    memcpy(kps, hk, sizeof(svo_keypoint) * (size_t)n);
    memcpy(desc, hd, (size_t)32 * n);
    return SVO_OK;
}

extern "C" int svo_orb_read_level(svo_ctx *ctx, int level, uint8_t *out, int *w, int *h)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_HIP(hipSetDevice(ctx->device));
    int rc = orb_alloc(ctx);
    if (rc) return rc;
    SVO_ARG(level >= 0 && level < ctx->orb_geom.nlevels, "level out of range");
    const OrbGeom &g = ctx->orb_geom;
    if (w) *w = g.w[level];
    if (h) *h = g.h[level];
    if (!out) return SVO_OK;
    // (svo_add_frame / the batch calls read level 0 in place from the caller's frames: image slot 0 then holds levels >= 1 only)
    SVO_ARG(level > 0 || ctx->orb_level0_in_slot, "level 0 was read in place from the input frame by the last extraction (svo_add_frame / batch): "
                                                  "it is the input image; svo_orb_extract copies it");
    SVO_HIP(hipMemcpy2DAsync(out, g.w[level], ctx->orb_slots + g.origin[level], g.pitch[level], g.w[level], g.h[level],
                             hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

extern "C" int svo_orb_read_candidates(svo_ctx *ctx, int level, float *out4, int cap, int *n_out)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_HIP(hipSetDevice(ctx->device));
    int rc = orb_alloc(ctx);
    if (rc) return rc;
    SVO_ARG(level >= 0 && level < ctx->orb_geom.nlevels && out4 && n_out, "bad argument");
    int n = 0;
    SVO_HIP(hipMemcpy(&n, ctx->orb_lvl_cnt + level, sizeof(int), hipMemcpyDeviceToHost));
    *n_out = n;
    if (n > cap) n = cap;
    if (n > 0) SVO_HIP(hipMemcpy(out4, ctx->orb_lvl_cand + (size_t)level * ctx->orb_cand_cap, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost));
    return SVO_OK;
}

extern "C" int svo_match_hamming(svo_ctx *ctx, const uint8_t *query, int nq, const uint8_t *train, int nt,
                                 int32_t *train_idx, float *distance, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(mem == SVO_MEM_HOST || mem == SVO_MEM_DEVICE, "bad mem");
    SVO_HIP(hipSetDevice(ctx->device));
    int rc = orb_alloc(ctx);
    if (rc) return rc;
    SVO_ARG(nq >= 0 && nt >= 0 && nq <= ctx->orb_kp_cap && nt <= ctx->orb_kp_cap, "descriptor count exceeds max_keypoints");
    if (nq == 0) return SVO_OK;
    SVO_ARG(query && train_idx && distance && (train || nt == 0), "null pointer");
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    const uint8_t *dq = query, *dt = train;
    if (mem == SVO_MEM_HOST) {
        // stage through the descriptor slots of image 0 / 1
        SVO_HIP(hipMemcpyAsync(ctx->orb_desc, query, (size_t)32 * nq, hipMemcpyHostToDevice, ctx->stream));
        if (nt > 0) SVO_HIP(hipMemcpyAsync(ctx->orb_desc + (size_t)32 * ctx->orb_kp_cap, train, (size_t)32 * nt, hipMemcpyHostToDevice, ctx->stream));
        dq = ctx->orb_desc; dt = ctx->orb_desc + (size_t)32 * ctx->orb_kp_cap;
    }
    orb_launch_match_fixed(ctx, dq, nq, dt, nt, ctx->stream);
    SVO_HIP(hipGetLastError());
    if (mem == SVO_MEM_HOST) {
        SVO_HIP(hipMemcpyAsync(train_idx, ctx->orb_midx[0], sizeof(int) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipMemcpyAsync(distance, ctx->orb_mdist[0], sizeof(float) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
    } else {
        SVO_HIP(hipMemcpyAsync(train_idx, ctx->orb_midx[0], sizeof(int) * (size_t)nq, hipMemcpyDeviceToDevice, ctx->stream));
        SVO_HIP(hipMemcpyAsync(distance, ctx->orb_mdist[0], sizeof(float) * (size_t)nq, hipMemcpyDeviceToDevice, ctx->stream));
    }
    return SVO_OK;
}

extern "C" int svo_triangulate(svo_ctx *ctx, const double P1[12], const double P2[12], const svo_pt2f *x1,
                               const svo_pt2f *x2, int n, svo_pt3f *out, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;     // shares X3 / cmp with a pending pose stage
    return stage_triangulate(ctx, P1, P2, x1, x2, n, out, mem);
}

extern "C" int svo_pnp_ransac(svo_ctx *ctx, const svo_pt3f *obj, const svo_pt2f *img, int n, const double K[9],
                              int iterations, float reproj_err, double confidence, svo_pnp_result *res,
                              uint8_t *inlier_mask, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    return stage_pnp_ransac(ctx, obj, img, n, K, iterations, reproj_err, confidence, res, inlier_mask, mem);
}

extern "C" int svo_add_frame(svo_ctx *ctx, const uint8_t *left, const uint8_t *right, int pitch, int mem,
                             svo_step_result *res)
{
    if (!ctx) return SVO_ERR_ARG;
    return pipeline_add_frame(ctx, left, right, pitch, mem, res);
}

extern "C" int svo_reset(svo_ctx *ctx)
{
    if (!ctx) return SVO_ERR_ARG;
    ctx->online_frames = 0; ctx->online_cur = 0;
    for (int i = 0; i < 16; i++) ctx->pose[i] = (i % 5 == 0) ? 1.0 : 0.0;
    return SVO_OK;
}

extern "C" int svo_get_pose(svo_ctx *ctx, double pose[16])
{
    if (!ctx || !pose) return SVO_ERR_ARG;
    memcpy(pose, ctx->pose, sizeof(double) * 16);
    return SVO_OK;
}

extern "C" int svo_set_pose(svo_ctx *ctx, const double pose[16])
{
    if (!ctx || !pose) return SVO_ERR_ARG;
    memcpy(ctx->pose, pose, sizeof(double) * 16);
    return SVO_OK;
}

extern "C" int svo_track_batch(svo_ctx *ctx, const uint8_t *left_frames, const uint8_t *right_frames,
                               int pitch, int64_t frame_stride, int n_frames, const double *pose0,
                               svo_step_result *results, int results_mem)
{
    if (!ctx) return SVO_ERR_ARG;
    return pipeline_track_batch(ctx, left_frames, right_frames, pitch, frame_stride, n_frames, pose0,
                                results, results_mem);
}

// ---- host-resident frame batches ---------------------------------------------------------------
// ctx may be NULL (ABI v6): page-locked memory does not belong to a context or a device (hipHostMallocPortable), so a
// runner can pin its frame buffers on one thread WHILE svo_create builds the context on another -- on a short run
// page-locking half a gigabyte (0.25 ms per MB) and context creation are each a quarter of the wall time.
extern "C" int svo_host_alloc(svo_ctx *ctx, size_t bytes, void **out)
{
    if (!out || bytes == 0) { if (ctx) ctx->err = "bad argument: null output / zero size"; return SVO_ERR_ARG; }
    *out = nullptr;
    if (ctx && hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return SVO_ERR_HIP; }
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e != hipSuccess) {
        if (ctx) ctx->err = std::string("hipHostMalloc: ") + hipGetErrorString(e);
        return SVO_ERR_HIP;
    }
    return SVO_OK;
}

extern "C" int svo_host_free(svo_ctx *ctx, void *p)
{
    if (p && hipHostFree(p) != hipSuccess) { if (ctx) ctx->err = "hipHostFree failed"; return SVO_ERR_HIP; }
    return SVO_OK;
}

static int frame_buffers(svo_ctx *ctx)
{
    if (ctx->copy_stream) return SVO_OK;
    const size_t per_cam = (size_t)ctx->stage_pitch * ctx->cfg.height * (size_t)(ctx->cfg.max_batch + 1);
    // every resource behind its own guard: a call that failed half way leaves what it made for the next attempt
    // (copy_stream, made last, is what marks the set complete)
    for (int k = 0; k < 2; k++) {
        if (!ctx->fb[k] && dev_alloc(ctx, &ctx->fb[k], 2 * per_cam) != SVO_OK) return SVO_ERR_HIP;
        if (!ctx->ev_up[k]) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_up[k], hipEventDisableTiming));
        if (!ctx->ev_fb_free[k]) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_fb_free[k], hipEventDisableTiming));
    }
    SVO_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    return SVO_OK;
}

extern "C" int svo_upload_frames_at(svo_ctx *ctx, int buf, int first_slot, const uint8_t *left_frames, const uint8_t *right_frames,
                                    int pitch, int64_t frame_stride, int n_frames);
extern "C" int svo_upload_frames(svo_ctx *ctx, int buf, const uint8_t *left_frames, const uint8_t *right_frames,
                                 int pitch, int64_t frame_stride, int n_frames)
{
    return svo_upload_frames_at(ctx, buf, 0, left_frames, right_frames, pitch, frame_stride, n_frames);
}

extern "C" int svo_upload_frames_at(svo_ctx *ctx, int buf, int first_slot, const uint8_t *left_frames, const uint8_t *right_frames,
                                    int pitch, int64_t frame_stride, int n_frames)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(buf == 0 || buf == 1, "buf must be 0 or 1");
    SVO_ARG(left_frames && right_frames, "null frames");
    SVO_ARG(first_slot >= 0 && n_frames >= 1 && first_slot + n_frames <= ctx->cfg.max_batch + 1, "first_slot + n_frames must be in [1, max_batch + 1]");
    // slot 0 is either uploaded or carried on the device (SVO_CONTINUE_CARRY_FRAME): nothing else may be left out
    SVO_ARG(first_slot <= 1, "first_slot must be 0 (a whole batch) or 1 (frame 0 is carried on the device)");
    SVO_ARG(pitch >= ctx->cfg.width && frame_stride >= (int64_t)pitch * ctx->cfg.height, "bad pitch / frame_stride");
    SVO_HIP(hipSetDevice(ctx->device));
    int rc = frame_buffers(ctx);
    if (rc) return rc;
    const int w = ctx->cfg.width, h = ctx->cfg.height, sp = ctx->stage_pitch;
    const size_t fbytes = (size_t)sp * h, per_cam = fbytes * (size_t)(ctx->cfg.max_batch + 1);
    // the batch that last read this buffer must have been ingested
    if (ctx->fb_used[buf]) SVO_HIP(hipStreamWaitEvent(ctx->copy_stream, ctx->ev_fb_free[buf], 0));
    const uint8_t *src[2] = {left_frames, right_frames};
    for (int cam = 0; cam < 2; cam++) {
        uint8_t *dst = ctx->fb[buf] + cam * per_cam + (size_t)first_slot * fbytes;
        if (pitch == sp && frame_stride == (int64_t)fbytes) {
            SVO_HIP(hipMemcpyAsync(dst, src[cam], fbytes * (size_t)n_frames, hipMemcpyHostToDevice, ctx->copy_stream));
        } else {
            for (int f = 0; f < n_frames; f++)
                SVO_HIP(hipMemcpy2DAsync(dst + f * fbytes, sp, src[cam] + f * frame_stride, pitch, w, h,
                                         hipMemcpyHostToDevice, ctx->copy_stream));
        }
    }
    SVO_HIP(hipEventRecord(ctx->ev_up[buf], ctx->copy_stream));
    ctx->fb_frames[buf] = first_slot + n_frames;
    ctx->fb_first[buf] = first_slot;
    return SVO_OK;
}

extern "C" int svo_wait_upload(svo_ctx *ctx, int buf)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(buf == 0 || buf == 1, "buf must be 0 or 1");
    SVO_ARG(ctx->fb_frames[buf] > 0, "nothing uploaded into this buffer");
    SVO_HIP(hipEventSynchronize(ctx->ev_up[buf]));
    return SVO_OK;
}

extern "C" int svo_track_uploaded(svo_ctx *ctx, int buf, int n_frames, const double *pose0,
                                  svo_step_result *results, int results_mem)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(buf == 0 || buf == 1, "buf must be 0 or 1");
    SVO_ARG(n_frames >= 2 && n_frames <= ctx->fb_frames[buf], "n_frames exceeds what was uploaded");
    // an upload that started at slot 1 left slot 0 to a carried frame: only an async launch with SVO_CONTINUE_CARRY_FRAME reads it
    SVO_ARG(ctx->fb_first[buf] == 0, "frame slot 0 of this buffer was not uploaded (svo_upload_frames_at first_slot = 1): "
                                     "launch it with svo_track_uploaded_async and SVO_CONTINUE_CARRY_FRAME");
    SVO_HIP(hipSetDevice(ctx->device));
    const size_t fbytes = (size_t)ctx->stage_pitch * ctx->cfg.height, per_cam = fbytes * (size_t)(ctx->cfg.max_batch + 1);
    SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_up[buf], 0));
    int rc = pipeline_track_batch(ctx, ctx->fb[buf], ctx->fb[buf] + per_cam, ctx->stage_pitch, (int64_t)fbytes, n_frames,
                                  pose0, results, results_mem);
    if (rc < 0) return rc;
    SVO_HIP(hipEventRecord(ctx->ev_fb_free[buf], ctx->stream));
    ctx->fb_used[buf] = true;
    return rc;
}

extern "C" int svo_track_uploaded_async(svo_ctx *ctx, int buf, int n_frames, const double *pose0, int continue_chain)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(buf == 0 || buf == 1, "buf must be 0 or 1");
    SVO_ARG(n_frames >= 2 && n_frames <= ctx->fb_frames[buf], "n_frames exceeds what was uploaded");
    SVO_ARG(ctx->async_tail - ctx->async_head < 2, "two batches are already outstanding: collect one first");
    SVO_HIP(hipSetDevice(ctx->device));
    const int r = (int)(ctx->async_tail & 1), n_pairs = n_frames - 1;
    if (!ctx->async_ready) {                       // an online-sized context (max_batch 1) sets the ring up on first use
        for (int k = 0; k < 2; k++) {
            if (!ctx->d_async[k] && dev_alloc(ctx, &ctx->d_async[k], sizeof(svo_step_result) * (size_t)ctx->cfg.max_batch) != SVO_OK) return SVO_ERR_HIP;
            if (!ctx->ev_async[k]) SVO_HIP(hipEventCreateWithFlags(&ctx->ev_async[k], hipEventDisableTiming));
        }
        if (!ctx->fetch_stream) SVO_HIP(hipStreamCreateWithFlags(&ctx->fetch_stream, hipStreamNonBlocking));
        ctx->async_ready = true;
    }
    SVO_ARG((continue_chain & ~(SVO_CONTINUE_CHAIN | SVO_CONTINUE_CARRY_FRAME)) == 0, "unknown continue_chain bits");
    SVO_ARG(!(continue_chain & SVO_CONTINUE_CARRY_FRAME) || (continue_chain & SVO_CONTINUE_CHAIN),
            "SVO_CONTINUE_CARRY_FRAME without SVO_CONTINUE_CHAIN: the carried frame belongs to the chain being continued");
    SVO_ARG(ctx->fb_first[buf] == 0 || (continue_chain & SVO_CONTINUE_CARRY_FRAME),
            "frame slot 0 of this buffer was not uploaded (svo_upload_frames_at first_slot = 1): it holds stale pixels unless the "
            "launch carries the previous batch's last frame (SVO_CONTINUE_CHAIN | SVO_CONTINUE_CARRY_FRAME)");
    if (continue_chain) {
        SVO_ARG(ctx->async_tail > 0, "continue_chain needs a previous async batch");
        const int pr = r ^ 1;
        SVO_ARG(ctx->async_last_pairs > 0, "continue_chain needs a previous async batch");
        ctx->seed_dev = ctx->d_async[pr][ctx->async_last_pairs - 1].pose;
    }
    const size_t fbytes = (size_t)ctx->stage_pitch * ctx->cfg.height, per_cam = fbytes * (size_t)(ctx->cfg.max_batch + 1);
    SVO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_up[buf], 0));
    int rc = pipeline_track_batch(ctx, ctx->fb[buf], ctx->fb[buf] + per_cam, ctx->stage_pitch, (int64_t)fbytes, n_frames,
                                  continue_chain ? nullptr : pose0, ctx->d_async[r], SVO_MEM_DEVICE,
                                  (continue_chain & SVO_CONTINUE_CARRY_FRAME) != 0);
    ctx->seed_dev = nullptr;
    if (rc < 0) return rc;
    SVO_HIP(hipEventRecord(ctx->ev_fb_free[buf], ctx->stream));
    ctx->fb_used[buf] = true;
    // the records land in d_async[r] at the end of the pose stage, wherever that ran
    SVO_HIP(hipEventRecord(ctx->ev_async[r], ctx->back_pending ? ctx->side_stream : ctx->stream));
    ctx->async_n[r] = n_pairs;
    ctx->async_last_pairs = n_pairs;
    ctx->async_tail++;
    ctx->carry_slot = n_pairs;                   // frame slot n_pairs now holds this batch's last frame
    return rc;
}

// Non-blocking companion of svo_collect_results (ABI v6): a streaming caller polls between frames.
extern "C" int svo_results_ready(svo_ctx *ctx, int *n_pairs)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(n_pairs, "null output");
    *n_pairs = 0;
    if (ctx->async_tail == ctx->async_head) return SVO_OK;
    const int r = (int)(ctx->async_head & 1);
    SVO_HIP(hipSetDevice(ctx->device));
    const hipError_t e = hipEventQuery(ctx->ev_async[r]);
    if (e == hipSuccess) *n_pairs = ctx->async_n[r];
    else if (e != hipErrorNotReady) SVO_HIP(e);
    return SVO_OK;
}

extern "C" int svo_collect_results(svo_ctx *ctx, svo_step_result *results, int n_pairs)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(ctx->async_tail != ctx->async_head, "no outstanding batch");
    const int r = (int)(ctx->async_head & 1);
    SVO_ARG(results && n_pairs >= 1 && n_pairs <= ctx->async_n[r], "the oldest outstanding batch has fewer pairs");
    SVO_HIP(hipSetDevice(ctx->device));
    // wait for THAT batch only (its successor may be running), then fetch on a stream of its own: the
    // copy stream may be busy with the next chunk's upload
    SVO_HIP(hipEventSynchronize(ctx->ev_async[r]));
    svo_step_result *h = (svo_step_result *)((char *)ctx->h_pinned + 4096);
    SVO_HIP(hipMemcpyAsync(h, ctx->d_async[r], sizeof(svo_step_result) * (size_t)n_pairs, hipMemcpyDeviceToHost, ctx->fetch_stream));
    SVO_HIP(hipStreamSynchronize(ctx->fetch_stream));
    memcpy(results, h, sizeof(svo_step_result) * (size_t)n_pairs);
    ctx->async_n[r] = 0;
    ctx->async_head++;
    return SVO_OK;
}

extern "C" int svo_chain_relative(svo_ctx *ctx, const double *T_rel_inv, const int32_t *ok, int n, const double *pose0,
                                  double *poses_out, int mem)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_HIP(hipSetDevice(ctx->device));
    return stage_chain_relative(ctx, T_rel_inv, ok, n, pose0, poses_out, mem);
}

// ---- read-back of the online state: what the reference keeps in Frame::features_* / shows in
// displayTracking ------------------------------------------------------------------------------
extern "C" int svo_get_frame_keypoints(svo_ctx *ctx, int side, svo_keypoint *kps, uint8_t *descriptors, int cap, int *n_out)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(n_out && cap >= 0 && (kps || cap == 0), "null output");
    SVO_ARG(side == 0 || side == 1, "side must be 0 (left) or 1 (right)");
    SVO_ARG(ctx->online_frames > 0, "no frame has been added");
    SVO_HIP(hipSetDevice(ctx->device));
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    const int cur = ctx->online_cur;
    *n_out = 0;
    if (ctx->cfg.track_mode == SVO_MODE_ORB) {
        const int slot = 2 * cur + side, kcap = ctx->orb_kp_cap;
        int n = 0;
        SVO_HIP(hipMemcpyAsync(&n, ctx->orb_n + slot, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
        n = n < kcap ? n : kcap;
        SVO_ARG(n <= cap, "keypoint capacity too small");
        if (n > 0) {
            SVO_HIP(hipMemcpyAsync(kps, (const svo_keypoint *)ctx->orb_kps + (size_t)slot * kcap, sizeof(svo_keypoint) * n,
                                   hipMemcpyDeviceToHost, ctx->stream));
            if (descriptors)
                SVO_HIP(hipMemcpyAsync(descriptors, ctx->orb_desc + (size_t)slot * kcap * 32, (size_t)32 * n, hipMemcpyDeviceToHost,
                                       ctx->stream));
            SVO_HIP(hipStreamSynchronize(ctx->stream));
        }
        *n_out = n;
        return SVO_OK;
    }
    SVO_ARG(side == 0, "LK mode detects on the left image only (src/tracking.cpp:94-113)");
    const int kcap = ctx->cfg.max_keypoints;
    int n = 0;
    SVO_HIP(hipMemcpyAsync(&n, ctx->kp_n + cur, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    n = n < kcap ? n : kcap;
    SVO_ARG(n <= cap, "keypoint capacity too small");
    if (n > 0) {
        std::vector<float2> xy((size_t)n);
        std::vector<float> resp((size_t)n);
        SVO_HIP(hipMemcpyAsync(xy.data(), ctx->kp_xy + (size_t)cur * kcap, sizeof(float2) * n, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipMemcpyAsync(resp.data(), ctx->kp_resp + (size_t)cur * kcap, sizeof(float) * n, hipMemcpyDeviceToHost, ctx->stream));
        SVO_HIP(hipStreamSynchronize(ctx->stream));
        for (int i = 0; i < n; i++) {                   // the cv::KeyPoint cv::FAST produces
            kps[i].x = xy[i].x; kps[i].y = xy[i].y; kps[i].size = 7.f; kps[i].angle = -1.f;
            kps[i].response = resp[i]; kps[i].octave = 0; kps[i].class_id = -1;
        }
    }
    *n_out = n;
    return SVO_OK;
}

// The same read-back for pair `pair` of the most recent svo_track_batch / svo_track_uploaded launch (ABI v6): what a
// caller needs to audit a batch against another implementation -- bench.py's self-check does.
extern "C" int svo_get_batch_tracks(svo_ctx *ctx, int pair, svo_pt2f *t1_left, svo_pt2f *t1_right, svo_pt2f *t2_right,
                                    svo_pt2f *t2_left, uint8_t *inlier, int cap, int *n_out)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(n_out && cap >= 0, "null output");
    SVO_ARG(pair >= 0 && pair < ctx->last_batch_pairs, "pair is not part of the last batch launch");
    SVO_HIP(hipSetDevice(ctx->device));
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    int *h_n = (int *)ctx->h_pinned;
    SVO_HIP(hipMemcpyAsync(h_n, ctx->m_out + pair, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    const int n = *h_n;
    SVO_ARG(n >= 0 && n <= ctx->cfg.max_keypoints, "corrupt track count");
    SVO_ARG(n <= cap, "track capacity too small");
    *n_out = n;
    if (n == 0) return SVO_OK;
    const size_t o = (size_t)pair * ctx->cfg.max_keypoints;
    svo_pt2f *dst[4] = {t1_left, t1_right, t2_right, t2_left};
    for (int k = 0; k < 4; k++) {
        if (!dst[k]) continue;
        if (k == 2 && ctx->cfg.track_mode == SVO_MODE_ORB) { memset(dst[k], 0, sizeof(svo_pt2f) * (size_t)n); continue; }
        SVO_HIP(hipMemcpyAsync(dst[k], ctx->cmp[k] + o, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (inlier) SVO_HIP(hipMemcpyAsync(inlier, pnp_inlier_mask(ctx) + o, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

extern "C" int svo_get_last_tracks(svo_ctx *ctx, svo_pt2f *t1_left, svo_pt2f *t1_right, svo_pt2f *t2_right,
                                   svo_pt2f *t2_left, uint8_t *inlier, int cap, int *n_out)
{
    if (!ctx) return SVO_ERR_ARG;
    SVO_ARG(n_out && cap >= 0, "null output");
    SVO_HIP(hipSetDevice(ctx->device));
    if (svo_wait_results(ctx) != SVO_OK) return SVO_ERR_HIP;
    const int n = ctx->online_frames >= 2 ? ctx->online_tracked : 0;
    SVO_ARG(n <= cap, "track capacity too small");
    *n_out = n;
    if (n == 0) return SVO_OK;
    svo_pt2f *dst[4] = {t1_left, t1_right, t2_right, t2_left};
    for (int k = 0; k < 4; k++) {
        if (!dst[k]) continue;
        if (k == 2 && ctx->cfg.track_mode == SVO_MODE_ORB) { memset(dst[k], 0, sizeof(svo_pt2f) * (size_t)n); continue; }
        SVO_HIP(hipMemcpyAsync(dst[k], ctx->cmp[k], sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (inlier) SVO_HIP(hipMemcpyAsync(inlier, pnp_inlier_mask(ctx), (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    SVO_HIP(hipStreamSynchronize(ctx->stream));
    return SVO_OK;
}

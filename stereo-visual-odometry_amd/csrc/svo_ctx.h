// svo_ctx.h -- the context object behind the C-ABI (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <functional>
#include <string>
#include <utility>
#include <vector>
#include "../../include/svo_abi.h"
#include "svo_kernels.h"

// Every device buffer of a context comes out of ONE hipMalloc (svo_create: ~30 separate allocations were a fifth of a
// short run's start-up): while `planning`, dev_alloc only records (where the pointer goes, size); dev_commit allocates
// the sum, hands the pointers out and runs the initialisations that were waiting for them (dev_defer).  Allocations made
// after the commit (lazy paths: ORB buffers of an LK context, frame buffers of an online context) are separate and
// remembered in `extra`.
struct DevArena {
    char *base = nullptr;
    size_t planned = 0;
    bool planning = false;
    std::vector<std::pair<void **, size_t>> plan;
    std::vector<std::function<int()>> after;
    std::vector<void *> extra;
};

struct svo_ctx {
    svo_config cfg;
    DevArena arena;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::string err;
    svo::PyrGeom geom;

    int n_img = 0;                    // images the batch buffers are sized for (max_batch + 1)
    // ---- stage-API resources
    uint8_t *slots = nullptr;         // num_slots pyramid slots
    std::vector<char> slot_built;
    uint8_t *stage_img = nullptr;     // device staging for host images (2 images, aligned pitch)
    int stage_pitch = 0;
    uint8_t *h_stage = nullptr;       // pinned mirror of stage_img: rows gathered on the host, one H2D copy per image
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool h_stage_busy[2] = {false, false};
    // ---- FAST scratch + outputs, n_img images
    uint8_t *score = nullptr; int spitch = 0; int64_t score_stride = 0;
    int *rowcount = nullptr; int64_t rowcount_stride = 0;
    float2 *kp_xy = nullptr; float *kp_resp = nullptr; int *kp_n = nullptr;
    // ---- LK buffers, max_batch items x max_keypoints
    float2 *pts_in = nullptr;
    float2 *pts_out[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t *status[4] = {nullptr, nullptr, nullptr, nullptr};
    uint8_t *keep = nullptr;
    float2 *cmp[4] = {nullptr, nullptr, nullptr, nullptr};    // compacted t1l, t1r, t2r, t2l
    int *m_out = nullptr;
    // ---- geometry buffers (geometry.hip)
    float *X3 = nullptr;              // triangulated points, 3 floats per point
    void *pnp_ws = nullptr; size_t pnp_ws_bytes = 0;
    svo_step_result *d_results = nullptr;
    // ---- batch-mode pyramid slots: 2 * n_img
    uint8_t *bslots = nullptr;
    // ---- online state (svo_add_frame)
    int online_frames = 0;            // frames fed since reset
    int online_cur = 0;               // which half of the 2-frame ring holds the latest frame
    int last_batch_pairs = 0;         // pairs of the most recent batch launch (svo_get_batch_tracks)
    int carry_slot = -1;              // frame slot that still holds the LAST frame of the previous async batch (its pyramids /
                                      // keypoints / descriptors): set only by a successful svo_track_uploaded_async, dropped by
                                      // every other entry point that writes frame slots (SVO_CONTINUE_CARRY_FRAME needs it)
    int online_tracked = 0;           // tracks of the last svo_add_frame pair (0 when it stopped before matching)
    double pose[16];
    // ---- ORB path (allocated on first use: orb_alloc)
    bool orb_ready = false;
    bool orb_qt_parallel = false;            // the node-parallel quadtree kernel is usable for this configuration
    bool orb_level0_in_slot = false;         // the last extraction copied level 0 into the image slots (false: read in place, OrbL0)
    bool orb_resize_staged = false;          // SVO_ORB_RESIZE_STAGED: the LDS-staged resize kernel on every level (A/B measurements, its test)
    bool orb_copy_level0 = false;            // SVO_ORB_COPY_LEVEL0: level 0 copied into the slot even where it could be read in place (A/B, its test)
    svo::OrbGeom orb_geom;
    uint8_t *orb_slots = nullptr, *orb_blur = nullptr;
    void *orb_xtab = nullptr, *orb_ytab = nullptr;       // cv::resize coordinate / weight tables
    float4 *orb_cell_cand = nullptr; int *orb_cell_cnt = nullptr;
    float4 *orb_lvl_cand = nullptr; int *orb_lvl_cnt = nullptr;
    int orb_node_cap = 0;                                // quadtree node capacity (largest level quota + slack)
    void *orb_qkeys = nullptr, *orb_qtmp = nullptr;      // quadtree key scratch (inputs larger than its LDS)
    int *orb_sel = nullptr, *orb_sel_cnt = nullptr, *orb_overflow = nullptr;
    void *orb_kps = nullptr; uint8_t *orb_desc = nullptr; int *orb_n = nullptr; int orb_kp_cap = 0, orb_cand_cap = 0;
    int *orb_midx[2] = {nullptr, nullptr}; float *orb_mdist[2] = {nullptr, nullptr};
    // ---- pinned host scratch
    void *h_pinned = nullptr; size_t h_pinned_bytes = 0;
    // ---- overlap mode (svo_set_overlap): the pose stage of batch k runs on side_stream while
    //      the caller's stream already ingests / tracks batch k+1
    bool overlap = false;
    hipStream_t side_stream = nullptr;
    // ---- host-frame batches (svo_upload_frames): two device frame buffers filled on copy_stream
    uint8_t *fb[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_fb_free[2] = {nullptr, nullptr};
    bool fb_used[2] = {false, false};
    int fb_frames[2] = {0, 0};
    int fb_first[2] = {0, 0};          // first frame slot of the last upload into the buffer (svo_upload_frames_at: 0 or 1)
    // svo_track_uploaded_async: two result buffers, collected in launch order
    svo_step_result *d_async[2] = {nullptr, nullptr};
    hipEvent_t ev_async[2] = {nullptr, nullptr};
    int async_n[2] = {0, 0};          // pairs of the outstanding batch in each buffer (0 = free)
    unsigned async_head = 0, async_tail = 0;   // next to collect / next to launch
    int async_last_pairs = 0;         // pairs of the most recently launched async batch
    const double *seed_dev = nullptr; // device-side seed pose of the next chain launch (continue_chain)
    hipStream_t fetch_stream = nullptr;
    bool async_ready = false;                 // d_async / ev_async / fetch_stream all exist (set after the last of them succeeded)
    hipEvent_t ev_front = nullptr, ev_back = nullptr;
    hipEvent_t ev_order = nullptr;    // svo_wait_stream / svo_signal_stream (made on first use)
    bool back_pending = false;
    int *kp_n_snap = nullptr;         // n_prev / n_cur (/ ORB capacity flags) of the batch the pose stage works on: 3 x max_batch
    // ---- timing
    // stage marks are HIP events recorded on the context's stream; they are resolved (elapsed
    // times averaged per stage over all steps since the last query) in svo_get_timing
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;                               // every event ever created
    size_t ev_used = 0;
    std::vector<std::pair<const char *, hipEvent_t>> marks;        // log since the last query
    std::vector<std::pair<const char *, float>> last_times;
};

#define SVO_HIP(call)                                                                         \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e__);                    \
            return SVO_ERR_HIP;                                                               \
        }                                                                                     \
    } while (0)

#define SVO_ARG(cond, msg)                                                                    \
    do {                                                                                      \
        if (!(cond)) { ctx->err = std::string("bad argument: ") + msg; return SVO_ERR_ARG; }   \
    } while (0)

namespace svo {
// svo_abi.hip: the context's device memory
int dev_alloc_raw(svo_ctx *ctx, void **out, size_t bytes);
template <class T> inline int dev_alloc(svo_ctx *ctx, T **out, size_t bytes) { return dev_alloc_raw(ctx, (void **)out, bytes); }
int dev_defer(svo_ctx *ctx, std::function<int()> fn);      // runs fn now, or after dev_commit while the arena is being planned
// geometry.hip
int geom_workspace_bytes(const svo_config &cfg, int n_items, size_t *bytes);
int geom_workspace_init(svo_ctx *ctx);
// snap: the frames' keypoint counts / capacity flags to freeze into ctx->kp_n_snap beside the triangulation (null: none)
struct SnapSpec { const int *n, *ovf; int fp0, fc0, fstep, per; };
void launch_triangulate_batch(svo_ctx *ctx, int n_items, int max_pts, const float2 *x1, const float2 *x2,
                              const int *n_pts, int n_fixed, const SnapSpec *snap = nullptr, hipStream_t st = nullptr);   // st null: the context's stream
void launch_snap_counts(svo_ctx *ctx, int n_items, const SnapSpec &snap, hipStream_t st);
void launch_pnp_batch(svo_ctx *ctx, int n_items, const float2 *img, const int *n_pts, int n_fixed, hipStream_t st);
// orb.hip
int orb_alloc(svo_ctx *ctx);
void orb_free(svo_ctx *ctx);
void timing_mark(svo_ctx *ctx, const char *name);     // HIP event on the context's stream when timing is enabled
const uint8_t *pnp_inlier_mask(const svo_ctx *ctx);
int stage_chain_relative(svo_ctx *ctx, const double *T, const int32_t *ok, int n, const double *pose0_host, double *out, int mem);
int stage_host_image(svo_ctx *ctx, const uint8_t *img, int pitch, int stage_idx, const uint8_t **dptr, int *dpitch);
int orb_extract_batch(svo_ctx *ctx, const uint8_t *img, const uint8_t *img2, int pitch, int64_t img_stride, int slot0,
                      int n_img, hipStream_t st, bool in_place = false);
int orb_match_pairs(svo_ctx *ctx, int n_pairs, int fp0, int fc0, int fstep, hipStream_t st);
void orb_launch_match_fixed(svo_ctx *ctx, const uint8_t *q, int nq, const uint8_t *t, int nt, hipStream_t st);
void launch_finalize_chain(svo_ctx *ctx, int n_pairs, const int *n_prev, const int *n_cur, const int *ovf,
                           const double *pose0_host, hipStream_t st);      // ctx->seed_dev != null: seed read on the device
}  // namespace svo

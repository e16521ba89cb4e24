// svo_kernels.h -- argument blocks and launch entry points of the HIP kernels (internal).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "svo_device.h"

namespace svo {

// ---- FAST (fast.hip) ------------------------------------------------------------------------
struct FastArgs {
    const uint8_t *img; int pitch; int64_t img_stride;     // batch b reads img + b*img_stride
    int w, h, thr, nms;
    uint8_t *score; int spitch; int64_t score_stride;      // suppressed score map (scratch)
    int *rowcount; int64_t rowcount_stride;                // per-row keypoint counts (scratch)
    float2 *kp_xy; float *kp_resp; int64_t kp_stride;      // outputs, `cap` entries per image
    int *n_out;                                            // keypoints found per image (may exceed cap)
    int cap;
};
void launch_fast(const FastArgs &a, int batch, hipStream_t st);
// svo_config.fast_keep_strongest: in-place selection of the `keep` highest-response corners per image, raster order kept
void launch_fast_keep_strongest(const FastArgs &a, int batch, int keep, hipStream_t st);

// ---- LK pyramid (pyramid.hip) ---------------------------------------------------------------
struct PyrArgs {
    PyrGeom g;
    const uint8_t *img; int pitch; int64_t img_stride;     // source images
    const uint8_t *img2;                                   // non-null: image b comes from (b odd ? img2 : img) + (b/2)*img_stride
    uint8_t *slots; int64_t slot_stride;                   // destination slots (b-th image -> slots + b*slot_stride)
};
void launch_pyramid(const PyrArgs &a, int batch, hipStream_t st);
void launch_pyr_read_level(const PyrGeom &g, const uint8_t *slot, int l, uint8_t *out, int out_pitch,
                           hipStream_t st);

// ---- pyramidal LK (lk.hip) ------------------------------------------------------------------
constexpr int kMaxChain = 4;
struct LkArgs {
    PyrGeom g;
    int ncalls;                              // 1 (cv::calcOpticalFlowPyrLK) or 4 (circular match)
    // call c tracks from image prev[c] to image next[c]; pointers are slot bases for batch item 0,
    // item b adds b*slot_stride
    const uint8_t *prev[kMaxChain], *next[kMaxChain];
    int64_t slot_stride;
    const float2 *pts_in; int64_t pts_stride;     // input points, item b at pts_in + b*pts_stride
    const int *n_pts;                             // per-item point count (device) or NULL -> n_fixed
    int n_fixed;
    int cap;                                      // points per item processed at most
    float2 *pts_out[kMaxChain];                   // call c output (failed points included)
    uint8_t *status[kMaxChain];
    uint8_t *keep;                                // ncalls == 4: deleteBadmatchFeatures predicate
    float match_err_f; double match_err;          // feature_match_error
    int accum;                                    // svo_config.lk_accum: SVO_LK_ACCUM_EXACT (lk.hip), _SSE2 or _SIMD128 (lk_sse2.hip)
    int gx, batch, spread;                                // filled by launch_lk: workgroups (sse2: waves) per item, items
};
void launch_lk(const LkArgs &a, int batch, int max_pts, hipStream_t st);
void launch_lk_sse2(const LkArgs &a, int batch, int max_pts, hipStream_t st);   // lk_sse2.hip; launch_lk dispatches on a.accum

// stable compaction of the four point lists by `keep` (one workgroup per item)
struct CompactArgs {
    const uint8_t *keep; const int *n_pts; int n_fixed; int64_t pts_stride; int cap;
    const float2 *in[4]; float2 *out[4];
    int *m_out;
};
void launch_compact(const CompactArgs &a, int batch, hipStream_t st);

}  // namespace svo

/*
 * svo_abi.h -- C-ABI of the MI355X-native stereo-VO hot path (libsvo_hip.so).
 *
 * The reference (liuzhenboo/Stereo-Visual-Odometry) has no plugin/FFI layer: its hot path is
 * the private part of lzb_vio::Tracking, which calls OpenCV 3.  Each entry point below replaces
 * one of those call sites (cited as reference file:line); the host-side C++ mirror of
 * lzb_vio::{System,Tracking,Frame,...} in stereo-visual-odometry_amd/host/ and the Python test
 * binding both sit on top of exactly this surface.  INTEGRATION.md shows the reference-side
 * binding a maintainer would add.
 *
 * Conventions
 *  - plain C types only; every function returns an int status (SVO_OK == 0, < 0 hard error,
 *    > 0 soft "tracking failed" reason mirroring the reference's failure exits); nothing throws
 *    or aborts across the boundary; svo_last_error() gives a message for the last hard error.
 *  - `mem` says where the caller's buffers live: SVO_MEM_HOST (copied H2D/D2H by the call) or
 *    SVO_MEM_DEVICE (HBM pointers, e.g. a torch tensor's data_ptr(); no copies).  STREAM ORDER of device buffers: every
 *    kernel of a call runs on the context's stream (svo_set_stream) -- the pose stage of an overlap-mode batch on the
 *    context's side stream --, and the library orders nothing against other streams by itself.  A caller that produces
 *    inputs or (zero-)fills outputs on ANOTHER stream calls svo_wait_stream(ctx, that_stream) before the entry point, and
 *    svo_signal_stream(ctx, consumer_stream) -- or svo_sync() -- before it reads device-resident results from another
 *    stream (ABI v7; both are event waits on the device, no host synchronisation).
 *  - one context per (thread, GPU); calls on a context are serialised by the caller.
 *  - images are 8-bit grayscale, row-major, `pitch` bytes per row.  A call's kernels read the caller's DEVICE frames in the order
 *    of the context's stream until the call's last front-end kernel (ORB mode reads pyramid level 0 in place; LK mode runs FAST
 *    on the frame itself): frames handed over with SVO_MEM_DEVICE stay unchanged until then (svo_sync, or stream order).
 */
#ifndef SVO_ABI_H
#define SVO_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_ABI_VERSION 9

/* status codes */
#define SVO_OK                 0
#define SVO_ERR_ARG           -1   /* bad argument / capacity exceeded */
#define SVO_ERR_HIP           -2   /* a HIP runtime call failed */
#define SVO_ERR_NOMEM         -3
#define SVO_ERR_STATE         -4   /* e.g. pyramid slot not built */
/* soft failures of one frame step == the reference's `return false` exits */
#define SVO_FAIL_FEW_KEYPOINTS 1   /* src/tracking.cpp:261  (< 30 FAST corners)           */
#define SVO_FAIL_FEW_TRACKS    2   /* src/tracking.cpp:274  (< num_features_tracking)      */
#define SVO_FAIL_INLIER_RATIO  3   /* src/tracking.cpp:491  (inliers / tracked < rate)     */
#define SVO_FAIL_ROTATION_GATE 4   /* src/tracking.cpp:308  (|euler| >= 0.1 rad)           */
#define SVO_FAIL_TRANSL_GATE   5   /* src/tracking.cpp:311  (|t|^2 outside the window)     */
#define SVO_FAIL_CAPACITY      6   /* not a reference exit: a frame of the pair had more keypoints
                                      than svo_config.max_keypoints (cv::FAST is uncapped); the
                                      pair is reported failed instead of tracking a truncated set */

#define SVO_MEM_HOST   0
#define SVO_MEM_DEVICE 1

typedef struct svo_ctx svo_ctx;

typedef struct { float x, y; } svo_pt2f;                 /* cv::Point2f */
typedef struct { float x, y, z; } svo_pt3f;              /* cv::Point3f */
typedef struct {                                         /* cv::KeyPoint */
    float x, y, size, angle, response;
    int32_t octave, class_id;
} svo_keypoint;

/* Mirrors the live keys of config/default.yaml + capacities. */
typedef struct {
    int32_t width, height;          /* image size, fixed per context                          */
    int32_t max_keypoints;          /* capacity per image; exceeding it is SVO_ERR_ARG         */
    int32_t max_batch;              /* frame pairs per svo_track_batch launch (>= 1)           */
    int32_t num_slots;              /* pyramid slots for the stage API (>= 4)                  */
    int32_t fast_threshold;         /* 20, hard-coded at src/tracking.cpp:99                   */
    int32_t num_features_tracking;  /* config/default.yaml:69                                  */
    int32_t iterations;             /* iterationsCount, :80                                    */
    float   reproj_err;             /* reprojectionError, :81                                  */
    float   confidence;             /* confidence, :82 (a float at src/tracking.cpp:481)       */
    double  feature_match_error;    /* :66                                                     */
    double  inlier_rate;            /* :77                                                     */
    double  min_move2, max_move2;   /* squared translation gate; LK mode: 0.0005^2, 100 (:311) */
    double  P1[12], P2[12];         /* projMatr1_/projMatr2_, src/parameter.cpp:44-45          */
    /* track_mode (config/default.yaml:75) and the ORBextractor constructor arguments (:89-93)   */
    int32_t track_mode;             /* SVO_MODE_LK ("LK_stereof2f_pnp") or SVO_MODE_ORB ("ORB_stereof2f_pnp");
                                       in ORB mode min_move2 / max_move2 = minmove^2 / maxmove^2 (:87-88) */
    int32_t orb_nfeatures;          /* nFeatures 2000 */
    float   orb_scale_factor;       /* fScaleFactor 1.2 */
    int32_t orb_nlevels;            /* nLevels 8 */
    int32_t orb_ini_th, orb_min_th; /* fIniThFAST 20, fMinThFAST 7 */
    /* ABI v6.  Order of the five float sums A11, A12, A22, b1, b2 inside cv::calcOpticalFlowPyrLK
       (src/tracking.cpp:593-618), the one place where upstream's result depends on the build's SIMD width:
       SVO_LK_ACCUM_EXACT (default): exact integer sums converted to float once -- upstream's acctype = int64
           variant, independent of any order (canonical choice C0);
       SVO_LK_ACCUM_SSE2: float accumulation in a lane order of upstream's x86 SIMD code AS RESTATED in oracle/lk.c
           mode 2 (four lanes over x = 0..19 + scalar tail for A, madd pairs (k, k + 4) over x = 0..15 + scalar tail
           for b) -- recalled from lkpyramid.cpp, NOT validated against an OpenCV binary (none exists in the build
           environment; tests/test_cv_crosscheck.py is the check for a box that has one).  Bit-identical to that
           oracle mode; about 1.9x the LK kernel time;
       SVO_LK_ACCUM_SIMD128 (ABI v7): the universal-intrinsic (CV_SIMD128) block restated whole = oracle mode 4: b
           as above, A in groups of eight pixels (four lanes over x = 0..15 + scalar tail x = 16..20).  Same cost.
       SVO_LK_ACCUM_SSE2_LEGACY (ABI v8): the older hand-written CV_SSE2 block restated whole = oracle mode 3: A as in
           SVO_LK_ACCUM_SSE2, b with one float add per pixel product (_mm_mullo / _mm_mulhi_epi16 of (It_k It_k) x (Ix_k Iy_k):
           pixels 0, 1, then 4, 5 of a group of eight into qb0, 2, 3, then 6, 7 into qb1).  About 2.2x the LK kernel time.
       Which of the three an x86 OpenCV 3 build runs depends on its version; none is validated
       against a binary.  YAML key `lk_accum: exact | sse2 | simd128 | sse2_legacy`. */
    int32_t lk_accum;
    /* ABI v6.  LK mode, fused entry points only (svo_add_frame / svo_track_*): 0 = track every cv::FAST corner, as
       the reference does (src/tracking.cpp:94-113); N > 0 = keep the N highest-response corners of every left image
       (ties: raster order first; the kept corners stay in raster order) -- BASELINE config #4's "2000 features per
       frame".  n_prev_kps / n_cur_kps then report the kept counts.  YAML key `fast_keep_strongest`. */
    int32_t fast_keep_strongest;
} svo_config;

#define SVO_MODE_LK  0
#define SVO_MODE_ORB 1
#define SVO_LK_ACCUM_EXACT 0
#define SVO_LK_ACCUM_SSE2  1
#define SVO_LK_ACCUM_SIMD128 2
#define SVO_LK_ACCUM_SSE2_LEGACY 3

typedef struct {                    /* solvePnPRansac + Rodrigues outcome */
    double rvec[3], tvec[3], R[9];
    int32_t n_inliers, ransac_iters, best_iter, lm_iters, ok, _pad;
} svo_pnp_result;

typedef struct {                    /* one Tracking::AddFrame step (LK mode) */
    int32_t ok;                     /* Track() result                                           */
    int32_t fail_stage;             /* 0 or one of SVO_FAIL_*                                   */
    int32_t n_prev_kps, n_cur_kps, n_tracked, n_inliers;
    int32_t ransac_iters, lm_iters;
    double  rvec[3], tvec[3], R[9];
    double  T_rel_inv[16];          /* inv([R t; 0 1]) -- what frame_pose_ is multiplied by     */
    double  pose[16];               /* frame_pose_ after this step (chained from the batch's
                                       initial pose; failed steps leave it unchanged)           */
} svo_step_result;

/* ---- lifecycle ------------------------------------------------------------------------- */
int         svo_abi_version(void);
int         svo_config_bytes(void);      /* sizeof(svo_config) of this build (ABI v6): a binding checks its own struct against it */
void        svo_default_config(svo_config *cfg, int width, int height);   /* default.yaml + KITTI rig */
int         svo_device_count(int *n);   /* HIP devices visible to this process (ABI v5): the multi-sequence runner deals
                                           sequences to them; SVO_ERR_HIP with *n = 0 when the runtime finds none */
int         svo_create(const svo_config *cfg, int device, svo_ctx **out);
void        svo_destroy(svo_ctx *ctx);
const char *svo_last_error(const svo_ctx *ctx);
int         svo_set_stream(svo_ctx *ctx, void *hip_stream);   /* NULL -> context's own stream */
int         svo_sync(svo_ctx *ctx);
/* ABI v7.  svo_wait_stream: everything queued on `hip_stream` so far happens-before whatever this context launches next
 * (an event recorded on hip_stream, waited for by the context's stream).  svo_signal_stream: everything this context has
 * launched so far -- the side-stream pose stage of an overlap-mode svo_track_batch included -- happens-before whatever is
 * queued on `hip_stream` next.  NULL = the legacy default stream.  Device-side waits only. */
int         svo_wait_stream(svo_ctx *ctx, void *hip_stream);
int         svo_signal_stream(svo_ctx *ctx, void *hip_stream);
/* ABI v9.  svo_signal_stream_inputs: every kernel that READS the caller's input buffers of the calls made so far (the frames
 * of svo_track_batch / svo_add_frame are read in place until the end of the front end) happens-before whatever is queued on
 * `hip_stream` next -- the stream may then overwrite or free them.  Unlike svo_signal_stream it does NOT wait for the
 * side-stream pose stage of an overlap-mode batch: a producer that calls it after every batch keeps the overlap of batch k's
 * pose stage with batch k + 1's front end (svo_signal_stream after every batch would serialise them through the producer's
 * stream).  Results are complete only after svo_signal_stream / svo_sync. */
int         svo_signal_stream_inputs(svo_ctx *ctx, void *hip_stream);
int         svo_num_levels(const svo_ctx *ctx);               /* LK pyramid levels actually built */

/* ---- stage API: one call per OpenCV call site of the reference ---------------------------- */

/* cv::FAST(img, kps, threshold, nonmax)  -- src/tracking.cpp:101 (Detect_OpenCVFASTFeatures),
 * src/ORBextractor.cpp:763,768.  Row-major ordered output. */
int svo_fast_detect(svo_ctx *ctx, const uint8_t *img, int pitch, int mem, int threshold,
                    int nonmax, svo_keypoint *out, int cap, int *n_out);

/* buildOpticalFlowPyramid half of cv::calcOpticalFlowPyrLK (src/tracking.cpp:593-618): builds the
 * padded 4-level pyramid of one image into slot `slot`; consecutive LK calls reuse it. */
int svo_build_pyramid(svo_ctx *ctx, int slot, const uint8_t *img, int pitch, int mem);
/* test/debug read-back of one level (without border) */
int svo_read_pyramid_level(svo_ctx *ctx, int slot, int level, uint8_t *out, int out_pitch, int mem,
                           int *w, int *h);

/* cv::calcOpticalFlowPyrLK(prev, next, prev_pts, next_pts, status, err, Size(21,21), 3,
 *   TermCriteria(COUNT+EPS, 30, 0.01), 0, 0.001)  -- src/tracking.cpp:593,600,607,613 */
int svo_lk_track(svo_ctx *ctx, int slot_prev, int slot_next, const svo_pt2f *prev_pts, int n,
                 svo_pt2f *next_pts, uint8_t *status, int mem);

/* Tracking::LK_Robust_Find_MuliImage_MatchedFeatures incl. deleteBadmatchFeatures
 * (src/tracking.cpp:583-660): the 4-call loop L1->R1->R2->L2->L1' fused per point, then the
 * stable filter.  out_* receive the M survivors in input order; *m_out = M. */
int svo_circular_match(svo_ctx *ctx, int slot_prevL, int slot_prevR, int slot_curL, int slot_curR,
                       const svo_pt2f *t1_left, int n, svo_pt2f *out_t1_left,
                       svo_pt2f *out_t1_right, svo_pt2f *out_t2_right, svo_pt2f *out_t2_left,
                       int *m_out, int mem);

/* cv::triangulatePoints + cv::convertPointsFromHomogeneous -- src/tracking.cpp:292-294, :190-192 */
int svo_triangulate(svo_ctx *ctx, const double P1[12], const double P2[12], const svo_pt2f *x1,
                    const svo_pt2f *x2, int n, svo_pt3f *out, int mem);

/* cv::solvePnPRansac(obj, img, K, 0, rvec, t, true, iterations, reproj_err, confidence, inliers,
 *   SOLVEPNP_ITERATIVE) + cv::Rodrigues -- src/tracking.cpp:485-488.  res is always a HOST
 * struct; inlier_mask (n bytes, may be NULL) follows `mem`. */
int svo_pnp_ransac(svo_ctx *ctx, const svo_pt3f *obj, const svo_pt2f *img, int n, const double K[9],
                   int iterations, float reproj_err, double confidence, svo_pnp_result *res,
                   uint8_t *inlier_mask, int mem);

/* ORBextractor::operator()(image, mask, keypoints, descriptors) -- src/ORBextractor.cpp:990-1055,
 * called twice per frame by Tracking::Detect_MyORBFeatures (src/tracking.cpp:502-532).  kps is a HOST
 * array of cv::KeyPoint records, desc n x 32 bytes (HOST).  per_level (8 ints, may be NULL) receives
 * the keypoints kept per pyramid level. */
int svo_orb_extract(svo_ctx *ctx, const uint8_t *img, int pitch, int mem, svo_keypoint *kps, uint8_t *desc,
                    int cap, int *n_out, int *per_level);
/* test/debug read-back of one ORB pyramid level (tight rows) of the last svo_orb_extract.  (After svo_add_frame / a batch call in
 * ORB mode level 0 may not be there -- those read it in place from the input frame -- and asking for it is SVO_ERR_ARG.) */
int svo_orb_read_level(svo_ctx *ctx, int level, uint8_t *out, int *w, int *h);
/* test/debug: FAST candidates (x, y, response, 0) of one level of the last svo_orb_extract, before the quadtree */
int svo_orb_read_candidates(svo_ctx *ctx, int level, float *out4, int cap, int *n_out);

/* DescriptorMatcher::create("BruteForce-Hamming")->match(query, train, matches) --
 * src/tracking.cpp:539-544: for every query row the first train row of minimum Hamming distance.
 * Descriptors are 32-byte rows (16-byte aligned when device-resident). */
int svo_match_hamming(svo_ctx *ctx, const uint8_t *query, int nq, const uint8_t *train, int nt, int32_t *train_idx,
                      float *distance, int mem);

/* ---- fused API: Tracking::AddFrame in LK mode (src/tracking.cpp:49-77, 258-344) ------------ */

/* Online step: feeds one stereo frame.  The first call only detects features (StereoInit_f2f,
 * :78-92) and returns SVO_OK with res->ok = 1, n_prev_kps = 0.  Later calls track against the
 * previous frame.  Returns SVO_OK (res->ok = 1), a SVO_FAIL_* code (res->ok = 0; the pose chain
 * skips this step, as the reference does), or a hard error < 0.  res is a HOST struct. */
int svo_add_frame(svo_ctx *ctx, const uint8_t *left, const uint8_t *right, int pitch, int mem,
                  svo_step_result *res);
int svo_reset(svo_ctx *ctx);                         /* back to INITING, pose = identity */
int svo_get_pose(svo_ctx *ctx, double pose[16]);     /* frame_pose_ (src/tracking.h:117)  */
int svo_set_pose(svo_ctx *ctx, const double pose[16]);   /* seeds frame_pose_ of the online path (e.g. a context
                                                            rebuilt for another frame size continues the chain) */

/* Batched step: n_frames consecutive stereo frames resident in HBM (frame f at base +
 * f*frame_stride), n_frames - 1 <= max_batch pairs processed as one set of launches
 * (every consecutive pair is independent: SURVEY.md section 0 fact 3), poses chained on device.
 * results: n_frames - 1 records, location per `results_mem`.  pose0 (host, may be NULL = identity)
 * seeds the chain.  Frame 0's features are detected as part of the batch. */
int svo_track_batch(svo_ctx *ctx, const uint8_t *left_frames, const uint8_t *right_frames,
                    int pitch, int64_t frame_stride, int n_frames, const double *pose0,
                    svo_step_result *results, int results_mem);

/* Overlap mode for svo_track_batch with DEVICE-resident results (off by default).  The pose stage
 * (RANSAC-EPnP + LM, gates, chain: one latency-bound wave per pair) of batch k then runs on a side
 * stream while the context's stream already builds pyramids / detects / tracks batch k+1.  The
 * results of a batch are complete after svo_sync(), or -- in stream order, without a host sync --
 * after svo_wait_results(); the next svo_track_batch call waits for them by itself before it
 * reuses the shared buffers. */
int svo_set_overlap(svo_ctx *ctx, int on);
int svo_wait_results(svo_ctx *ctx);

/* Host-resident frame batches: image ingest off the critical path (replaces
 * the per-frame cv::imread -> Tracking::AddFrame hand-over of reference src/System.cpp:46-58,75-104
 * for the batched runner).
 *   svo_host_alloc / svo_host_free : page-locked host memory, so decoder threads write straight
 *       into DMA-able buffers;
 *   svo_upload_frames : ASYNCHRONOUS host-to-device copy of n_frames stereo frames (frame f at
 *       base + f*frame_stride, rows `pitch` apart) into the context's device frame buffer `buf`
 *       (0 or 1) on the context's copy stream -- it runs beside the kernels of the batch that
 *       lives in the other buffer.  The host memory must stay untouched until svo_wait_upload;
 *   svo_track_uploaded : svo_track_batch on the frames of buffer `buf` (ordered after their upload
 *       on the device, no host wait). */
int svo_host_alloc(svo_ctx *ctx, size_t bytes, void **out);   /* ctx may be NULL (ABI v6): the memory is portable across
                                                                  devices, so it can be pinned while svo_create still runs */
int svo_host_free(svo_ctx *ctx, void *p);
int svo_upload_frames(svo_ctx *ctx, int buf, const uint8_t *left_frames, const uint8_t *right_frames,
                      int pitch, int64_t frame_stride, int n_frames);
/* ABI v7: the same into frame slots first_slot .. first_slot + n_frames - 1 of the buffer (svo_upload_frames = first_slot 0).
 * A stream's micro-batch whose frame 0 is CARRIED on the device (SVO_CONTINUE_CARRY_FRAME) uploads its new frames only:
 * first_slot = 1; slot 0 is then never read. */
int svo_upload_frames_at(svo_ctx *ctx, int buf, int first_slot, const uint8_t *left_frames, const uint8_t *right_frames,
                         int pitch, int64_t frame_stride, int n_frames);
int svo_wait_upload(svo_ctx *ctx, int buf);
int svo_track_uploaded(svo_ctx *ctx, int buf, int n_frames, const double *pose0,
                       svo_step_result *results, int results_mem);
/* The same without waiting for the GPU (ABI v4): the n_frames - 1 step records stay in the context
 * until svo_collect_results copies them to a HOST array (it waits for that batch only).  Up to TWO
 * batches may be outstanding, collected in launch order, so a caller keeps the GPU busy like this:
 *     upload(0); track_async(0);  upload(1); track_async(1); collect(0);  upload(2); track_async(2); collect(1); ...
 * -- the upload of chunk k+1 and, in overlap mode, the pose stage of chunk k run beside chunk k+1's
 * front end, and the host only ever waits for a batch that has a successor queued behind it.
 * continue_chain != 0 seeds the pose chain with the LAST pose of the previous async batch on the device
 * (no host round trip; pose0 is ignored); 0 seeds it with pose0 (NULL = identity).
 * ABI v6: continue_chain may also carry SVO_CONTINUE_CARRY_FRAME (continue_chain = SVO_CONTINUE_CHAIN |
 * SVO_CONTINUE_CARRY_FRAME): the caller states that frame 0 of this batch IS the last frame of the previous async batch
 * (the halo frame of a stream's micro-batches); its pyramids / keypoints / descriptors are then carried over on the device
 * instead of being computed again from the uploaded copy. */
#define SVO_CONTINUE_CHAIN       1
#define SVO_CONTINUE_CARRY_FRAME 2
int svo_track_uploaded_async(svo_ctx *ctx, int buf, int n_frames, const double *pose0, int continue_chain);
int svo_collect_results(svo_ctx *ctx, svo_step_result *results, int n_pairs);
/* ABI v6, non-blocking: *n_pairs = the pairs of the OLDEST outstanding async batch when its records are complete (a
 * following svo_collect_results does not wait), 0 when it is still running or nothing is outstanding.  What a
 * streaming caller polls between frames (host: System::StreamPoll behind Step_ros, reference src/System.cpp:60-74). */
int svo_results_ready(svo_ctx *ctx, int *n_pairs);

/* Read-back of the online state (after svo_add_frame), for callers that keep the reference's
 * per-frame carriers or draw what Tracking::displayTracking drew (src/tracking.cpp:345-382):
 *   svo_get_frame_keypoints : the keypoints detected on the frame just added.  LK mode: the
 *       cv::FAST corners of the LEFT image, in cv::FAST order, as the cv::KeyPoint records
 *       Detect_OpenCVFASTFeatures pushes into Frame::features_left_ (src/tracking.cpp:94-113);
 *       side must be 0.  ORB mode: side 0 / 1 = left / right ORB keypoints and, if `descriptors`
 *       is not NULL, their n x 32 descriptor bytes (Frame::left_/right_Descriptors_, :511-526).
 *   svo_get_last_tracks : the matched tracks that fed the pose solver for the pair just tracked
 *       (t1_left, t1_right, t2_right -- zeros in ORB mode --, t2_left) and the RANSAC inlier flags;
 *       n = 0 when the step stopped before matching.  Any output pointer may be NULL.
 * Host pointers; capacities in elements; SVO_ERR_ARG when a capacity is too small. */
int svo_get_frame_keypoints(svo_ctx *ctx, int side, svo_keypoint *kps, uint8_t *descriptors, int cap, int *n_out);
int svo_get_last_tracks(svo_ctx *ctx, svo_pt2f *t1_left, svo_pt2f *t1_right, svo_pt2f *t2_right,
                        svo_pt2f *t2_left, uint8_t *inlier, int cap, int *n_out);
/* ABI v6: the same for pair `pair` (0-based) of the most recent svo_track_batch / svo_track_uploaded(_async) launch --
 * valid until the next launch on this context; waits for that batch's pose stage. */
int svo_get_batch_tracks(svo_ctx *ctx, int pair, svo_pt2f *t1_left, svo_pt2f *t1_right, svo_pt2f *t2_right,
                         svo_pt2f *t2_left, uint8_t *inlier, int cap, int *n_out);

/* Serial prefix product of n inverse relative motions (svo_step_result.T_rel_inv, row-major 4x4),
 * skipping pairs with ok == 0:  poses_out[p] = pose0 * prod_{q <= p, ok[q]} T[q]  -- the
 * `frame_pose_ = frame_pose_ * T.inv()` recurrence of reference src/tracking.cpp:318 for frame
 * pairs that were tracked as independent chunks (other launches, contexts or GPUs: SURVEY.md 8e
 * granularity 2).  pose0 is a HOST pointer (NULL = identity); T_rel_inv, ok and poses_out live where
 * `mem` says.  SVO_MEM_HOST: the call returns when poses_out is complete; SVO_MEM_DEVICE: one
 * launch in stream order on the context's stream, no host synchronisation. */
int svo_chain_relative(svo_ctx *ctx, const double *T_rel_inv, const int32_t *ok, int n,
                       const double *pose0, double *poses_out, int mem);

/* Kernel-level timing of the last svo_track_batch / svo_add_frame, measured with HIP events on
 * the context's stream: fills up to `cap` (name, milliseconds) pairs, returns the count.
 * Enabled by svo_enable_timing(ctx, 1); adds event records between stages. */
int svo_enable_timing(svo_ctx *ctx, int on);
int svo_get_timing(svo_ctx *ctx, const char **names, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* SVO_ABI_H */

// tracking.cpp -- lzb_vio::Tracking on top of the HIP library (reference src/tracking.cpp).
// The state machine is the reference's (AddFrame :49-77, StereoInit_f2f :78-92, Track :115-128);
// everything below it is ONE svo_add_frame call: FAST -> 4x LK -> filter -> triangulate ->
// solvePnPRansac -> gates -> frame_pose_ update all happen on the GPU.
#include "lzb_vio/tracking.h"

namespace lzb_vio {

Tracking::Tracking(System *system, Parameter::Ptr parameter, Sensors::Ptr sensors)
{
    sensors_ = sensors;
    system_ = system;
    parameter_ = parameter;
    memset(&last_, 0, sizeof(last_));
    Readparameter();
    // Config is a process-wide singleton (as in the reference, src/config.cpp:26): everything this object
    // needs from it is read NOW, so several System objects -- one per sequence -- can be built one after
    // the other and then run side by side
    max_keypoints_key_ = Config::Has("max_keypoints") ? Config::Get<int>("max_keypoints") : 0;
    // order of the float sums inside cv::calcOpticalFlowPyrLK (include/svo_abi.h): `exact` is independent of the
    // build; `sse2` / `simd128` / `sse2_legacy` add them in float in the lane orders of upstream's x86 SIMD code as restated in
    // oracle/lk.c (modes 2 / 4 / 3; recalled, not validated against an OpenCV binary) at about 1.9-2.2 x the LK kernel time
    fast_keep_strongest_ = Config::Has("fast_keep_strongest") ? Config::Get<int>("fast_keep_strongest") : 0;
    if (Config::Has("lk_accum")) {
        const std::string v = Config::Get<std::string>("lk_accum");
        if (v == "sse2") lk_accum_ = SVO_LK_ACCUM_SSE2;
        else if (v == "simd128") lk_accum_ = SVO_LK_ACCUM_SIMD128;
        else if (v == "sse2_legacy") lk_accum_ = SVO_LK_ACCUM_SSE2_LEGACY;
        else if (v != "exact") LZB_LOG("WARNING", "lk_accum: '%s' is none of 'exact', 'sse2', 'simd128', 'sse2_legacy'; using 'exact'", v.c_str());
    }
}

Tracking::~Tracking()
{
    if (ctx_) svo_destroy(ctx_);
}

void Tracking::Readparameter()
{
    num_features_init_ = parameter_->num_features_init_;
    num_features_ = parameter_->num_features_;
    num_features_tracking_bad_ = parameter_->num_features_tracking_bad_;
    num_features_needed_for_keyframe_ = parameter_->num_features_needed_for_keyframe_;
    init_landmarks_ = parameter_->init_landmarks_;
    feature_match_error_ = parameter_->feature_match_error_;
    track_mode_ = parameter_->track_mode_;
    num_features_tracking_ = parameter_->num_features_tracking_;
    inlier_rate_ = parameter_->inlier_rate_;
    iterationsCount_ = parameter_->iterationsCount_;
    reprojectionError_ = parameter_->reprojectionError_;
    confidence_ = parameter_->confidence_;
    maxmove_ = parameter_->maxmove_;
    minmove_ = parameter_->minmove_;
    nFeatures_ = parameter_->nFeatures_;
    fScaleFactor_ = parameter_->fScaleFactor_;
    nLevels_ = parameter_->nLevels_;
    fIniThFAST_ = parameter_->fIniThFAST_;
    fMinThFAST_ = parameter_->fMinThFAST_;
}

void Tracking::Set_vo(System *slam) { system_ = slam; }

bool Tracking::Reset()
{
    status_ = TrackingStatus::INITING;
    current_frame_ = nullptr; last_frame_ = nullptr;
    frame_pose_ = Pose4x4();
    Px_ = Py_ = Pz_ = 0;
    memset(&last_, 0, sizeof(last_));
    if (ctx_) svo_reset(ctx_);
    return true;
}

// The HIP context is sized by the first frame (the reference learns the size from cv::imread too).
bool Tracking::EnsureContext(int width, int height, int max_batch)
{
    if (ctx_ && width == ctx_w_ && height == ctx_h_ && max_batch <= ctx_batch_) return true;
    const bool resized = ctx_ && (width != ctx_w_ || height != ctx_h_);
    if (resized)
        LZB_LOG("WARNING", "frame size changed from %dx%d to %dx%d: the HIP context is rebuilt; this frame only "
                "re-initialises the tracker (no motion is estimated for it), the pose chain continues from frame_pose_",
                ctx_w_, ctx_h_, width, height);
    if (ctx_) { svo_destroy(ctx_); ctx_ = nullptr; }
    svo_config cfg;
    svo_default_config(&cfg, width, height);
    cfg.max_batch = max_batch;
    // cv::FAST is uncapped in the reference; the device buffers are not.  Capacity per image: the additive
    // YAML key max_keypoints, else one keypoint per 24 pixels (NMS keeps at most one corner per 3x3
    // block; textured KITTI frames hold 2-5 k, i.e. one per ~100-200 pixels), never less than 8192 or
    // than what the ORB extractor is asked for.  A frame that still exceeds it fails its pairs with
    // SVO_FAIL_CAPACITY, which is logged as an error below.
    {
        long cap = max_keypoints_key_ > 0 ? max_keypoints_key_ : (long)width * height / 24;
        if (cap < 8192) cap = 8192;
        if (cap < 2L * nFeatures_) cap = 2L * nFeatures_;
        if (track_mode_ == "ORB_stereof2f_pnp" && cap > 16384) cap = 16384;     // 16-bit indices in the ORB kernels
        if (cap > (1 << 20)) cap = 1 << 20;
        cfg.max_keypoints = (int)cap;
    }
    cfg.fast_threshold = 20;                                     // hard-coded, src/tracking.cpp:99
    cfg.num_features_tracking = num_features_tracking_;
    cfg.iterations = iterationsCount_;
    cfg.reproj_err = reprojectionError_;
    cfg.confidence = confidence_;
    cfg.feature_match_error = feature_match_error_;
    cfg.inlier_rate = inlier_rate_;
    cfg.lk_accum = lk_accum_;
    cfg.fast_keep_strongest = fast_keep_strongest_ > 0 ? fast_keep_strongest_ : 0;
    if (track_mode_ == "ORB_stereof2f_pnp") {
        // the shipped default (config/default.yaml:75): ORBextractor(nFeatures, fScaleFactor, nLevels,
        // fIniThFAST, fMinThFAST) (src/tracking.cpp:20) and the configured minmove / maxmove gate (:215)
        cfg.track_mode = SVO_MODE_ORB;
        cfg.orb_nfeatures = nFeatures_; cfg.orb_scale_factor = fScaleFactor_; cfg.orb_nlevels = nLevels_;
        cfg.orb_ini_th = fIniThFAST_; cfg.orb_min_th = fMinThFAST_;
        cfg.min_move2 = minmove_ * minmove_;
        cfg.max_move2 = maxmove_ * maxmove_;
    } else {
        cfg.track_mode = SVO_MODE_LK;
        cfg.min_move2 = 0.0005 * 0.0005;                         // LK mode, src/tracking.cpp:311
        cfg.max_move2 = 100.0;
    }
    memcpy(cfg.P1, sensors_->projMatr1_, sizeof(cfg.P1));
    memcpy(cfg.P2, sensors_->projMatr2_, sizeof(cfg.P2));
    int rc = svo_create(&cfg, device_, &ctx_);
    if (rc != SVO_OK) {
        LZB_LOG("ERROR", "svo_create failed (%d): a HIP device is required, there is no CPU path", rc);
        ctx_ = nullptr;
        return false;
    }
    ctx_w_ = width; ctx_h_ = height; ctx_batch_ = max_batch;
    if (resized) svo_set_pose(ctx_, frame_pose_.m);          // the new context's first frame only initialises; the chain goes on
    return true;
}

bool Tracking::AddFrame(Frame::Ptr frame)
{
    current_frame_ = frame;
    bool ok = true;
    switch (status_) {
    case TrackingStatus::INITING:
        StereoInit_f2f();
        break;
    case TrackingStatus::TRACKING_GOOD:
        ok = Track();
        last_frame_ = current_frame_;          // on success AND failure (src/tracking.cpp:59-68)
        return ok;
    case TrackingStatus::LOST:
        break;
    }
    return true;
}

static bool feed(svo_ctx *ctx, Frame::Ptr f, svo_step_result *res, int *rc_out)
{
    const cv::Mat &L = f->left_img_, &R = f->right_img_;
    if (L.empty() || R.empty() || L.rows != R.rows || L.cols != R.cols || L.step != R.step) {
        LZB_LOG("ERROR", "stereo frame %lu has missing or mismatched images", f->id_);
        *rc_out = SVO_ERR_ARG;
        return false;
    }
    *rc_out = svo_add_frame(ctx, L.data, R.data, (int)L.step, SVO_MEM_HOST, res);
    if (*rc_out < 0) LZB_LOG("ERROR", "svo_add_frame: %s", svo_last_error(ctx));
    if (*rc_out == SVO_FAIL_CAPACITY)
        LZB_LOG("ERROR", "frame %lu: more keypoints than the context's capacity (YAML key max_keypoints); "
                "the pair was NOT tracked and the pose keeps its previous value", f->id_);
    return *rc_out == SVO_OK;
}

// Frame::features_left_ / features_right_ / *_Descriptors_ as the reference's detectors leave them
// (Detect_OpenCVFASTFeatures src/tracking.cpp:94-113, Detect_MyORBFeatures :502-532); optional
// because it costs a device-to-host copy and N heap allocations per frame.
void Tracking::FillFeatures()
{
    if (!fill_features_ || !ctx_ || !current_frame_) return;
    const bool orb = track_mode_ == "ORB_stereof2f_pnp";
    std::vector<svo_keypoint> kps(65536);
    std::vector<uint8_t> desc;
    if (orb) desc.resize(kps.size() * 32);
    for (int side = 0; side < (orb ? 2 : 1); side++) {
        int n = 0;
        if (svo_get_frame_keypoints(ctx_, side, kps.data(), orb ? desc.data() : nullptr, (int)kps.size(), &n) != SVO_OK) {
            LZB_LOG("ERROR", "svo_get_frame_keypoints: %s", svo_last_error(ctx_));
            return;
        }
        auto &dst = side == 0 ? current_frame_->features_left_ : current_frame_->features_right_;
        dst.clear();
        dst.reserve((size_t)n);
        for (int i = 0; i < n; i++) {
            cv::KeyPoint kp;
            kp.pt.x = kps[i].x; kp.pt.y = kps[i].y; kp.size = kps[i].size; kp.angle = kps[i].angle;
            kp.response = kps[i].response; kp.octave = kps[i].octave; kp.class_id = kps[i].class_id;
            Feature::Ptr f(new Feature(current_frame_, kp));
            f->is_on_left_image_ = side == 0;
            if (orb) { f->Descriptor_.create(1, 32); memcpy(f->Descriptor_.ptr(0), desc.data() + (size_t)i * 32, 32); }
            dst.push_back(f);
        }
        if (orb) {
            cv::Mat &D = side == 0 ? current_frame_->left_Descriptors_ : current_frame_->right_Descriptors_;
            D.create(n > 0 ? n : 1, 32);
            D.rows = n;
            for (int i = 0; i < n; i++) memcpy(D.ptr(i), desc.data() + (size_t)i * 32, 32);
        }
    }
}

bool Tracking::GetLastTracks(std::vector<cv::Point2f> &t1_left, std::vector<cv::Point2f> &t1_right,
                             std::vector<cv::Point2f> &t2_left, std::vector<unsigned char> &inlier)
{
    t1_left.clear(); t1_right.clear(); t2_left.clear(); inlier.clear();
    if (!ctx_) return false;
    const int cap = last_.n_tracked > 0 ? last_.n_tracked : 0;
    std::vector<svo_pt2f> a((size_t)cap + 1), b((size_t)cap + 1), c((size_t)cap + 1);
    inlier.resize((size_t)cap + 1);
    int n = 0;
    if (svo_get_last_tracks(ctx_, a.data(), b.data(), nullptr, c.data(), inlier.data(), cap, &n) != SVO_OK) return false;
    inlier.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        t1_left.push_back(cv::Point2f{a[i].x, a[i].y});
        t1_right.push_back(cv::Point2f{b[i].x, b[i].y});
        t2_left.push_back(cv::Point2f{c[i].x, c[i].y});
    }
    return true;
}

bool Tracking::StereoInit_f2f()
{
    if (!EnsureContext(current_frame_->left_img_.cols, current_frame_->left_img_.rows)) return false;
    svo_reset(ctx_);
    int rc;
    feed(ctx_, current_frame_, &last_, &rc);
    FillFeatures();
    last_frame_ = current_frame_;
    status_ = TrackingStatus::TRACKING_GOOD;
    return rc >= 0;
}

bool Tracking::Track()
{
    if (track_mode_ == "LK_stereof2f_pnp") return LK_StereoF2F_PnP_Track();
    if (track_mode_ == "ORB_stereof2f_pnp") return ORB_StereoF2F_PnP_Track();
    return false;                               // any other string: Track() is always false (:127)
}

// Both modes are one svo_add_frame call; the context was created for the configured track_mode.
bool Tracking::LK_StereoF2F_PnP_Track() { return TrackOnGpu(); }
bool Tracking::ORB_StereoF2F_PnP_Track() { return TrackOnGpu(); }

bool Tracking::TrackOnGpu()
{
    if (!EnsureContext(current_frame_->left_img_.cols, current_frame_->left_img_.rows)) return false;
    int rc;
    bool ok = feed(ctx_, current_frame_, &last_, &rc);
    if (rc < 0) return false;
    FillFeatures();
    if (ok) {
        memcpy(frame_pose_.m, last_.pose, sizeof(frame_pose_.m));
        Px_ = frame_pose_.m[3]; Py_ = frame_pose_.m[7]; Pz_ = frame_pose_.m[11];
    }
    return ok;
}

// Batched equivalent of n_frames - 1 AddFrame calls on frames already in device buffer `buf`.
bool Tracking::TrackUploaded(int buf, int n_frames, std::vector<svo_step_result> &out)
{
    return TrackUploadedAsync(buf, n_frames) && CollectUploaded(out);
}

bool Tracking::TrackUploadedAsync(int buf, int n_frames, bool continue_chain)
{
    if (!ctx_ || n_frames < 2 || async_tail_ - async_head_ >= 2) return false;
    // a chunk that continues the chain starts with the previous chunk's last frame (the runner's and the stream's one-frame
    // halo): its features are carried over on the device instead of being extracted again
    int rc = svo_track_uploaded_async(ctx_, buf, n_frames, frame_pose_.m, continue_chain ? (SVO_CONTINUE_CHAIN | SVO_CONTINUE_CARRY_FRAME) : 0);
    if (rc < 0) {
        LZB_LOG("ERROR", "svo_track_uploaded_async: %s", svo_last_error(ctx_));
        return false;
    }
    async_pairs_[async_tail_ & 1] = n_frames - 1;
    async_tail_++;
    return true;
}

int Tracking::ResultsReady()
{
    int n = 0;
    if (!ctx_ || async_tail_ == async_head_ || svo_results_ready(ctx_, &n) != SVO_OK) return 0;
    return n;
}

bool Tracking::CollectUploaded(std::vector<svo_step_result> &out)
{
    if (!ctx_ || async_tail_ == async_head_) return false;
    const int n = async_pairs_[async_head_ & 1];
    const size_t first = out.size();
    out.resize(first + (size_t)n);
    int rc = svo_collect_results(ctx_, out.data() + first, n);
    if (rc < 0) {                                            // the context's ring did not advance either: stay in step with it
        LZB_LOG("ERROR", "svo_collect_results: %s", svo_last_error(ctx_));
        out.resize(first);
        return false;
    }
    async_head_++;
    for (size_t i = first; i < out.size(); i++)
        if (out[i].fail_stage == SVO_FAIL_CAPACITY)
            LZB_LOG("ERROR", "pair %zu of the batch: a frame has more keypoints than the context's capacity "
                    "(YAML key max_keypoints); the pair was NOT tracked", i - first);
    last_ = out.back();
    memcpy(frame_pose_.m, last_.pose, sizeof(frame_pose_.m));
    Px_ = frame_pose_.m[3]; Py_ = frame_pose_.m[7]; Pz_ = frame_pose_.m[11];
    status_ = TrackingStatus::TRACKING_GOOD;
    return true;
}

}  // namespace lzb_vio

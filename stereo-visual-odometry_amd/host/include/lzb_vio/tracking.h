// tracking.h -- host-side mirror of lzb_vio::Tracking (reference include/lzb_vio/tracking.h:29-154).
// The public surface is the reference's: Tracking(System*, Parameter::Ptr, Sensors::Ptr),
// AddFrame(Frame::Ptr), GetStatus(), Set_vo(System*).  The private OpenCV stages are replaced by
// ONE call into the HIP library per frame (svo_add_frame, include/svo_abi.h); there is no CPU
// implementation behind this class.  Additive API: GetPose(), LastResult().
#pragma once
#ifndef lzb_vio_TRACKING_H
#define lzb_vio_TRACKING_H

#include "lzb_vio/feature.h"
#include "lzb_vio/frame.h"
#include "lzb_vio/parameter.h"
#include "lzb_vio/sensors.h"
#include "svo_abi.h"

namespace lzb_vio {
class System;

enum class TrackingStatus { INITING, TRACKING_GOOD, LOST };

class Tracking {
public:
    typedef std::shared_ptr<Tracking> Ptr;
    Tracking(System *system, Parameter::Ptr parameter, Sensors::Ptr sensors);
    ~Tracking();
    void Set_vo(System *vo);
    bool AddFrame(Frame::Ptr frame);
    TrackingStatus GetStatus() const { return status_; }

    // private no-op in the reference (include/lzb_vio/tracking.h:46, src/tracking.cpp:662-665; its call
    // sites are commented out); public here and it does what the name says: back to INITING, identity pose
    bool Reset();

    // additive (the reference has no getter for frame_pose_)
    Pose4x4 GetPose() const { return frame_pose_; }
    const svo_step_result &LastResult() const { return last_; }
    void SetDevice(int device) { device_ = device; }         // HIP device of the context (before the first frame); default 0
    int Device() const { return device_; }
    void SetFillFeatures(bool on) { fill_features_ = on; }   // populate Frame::features_* / *_Descriptors_ (costs a D2H)
    // the matched tracks of the pair just tracked + RANSAC inlier flags (what displayTracking drew)
    bool GetLastTracks(std::vector<cv::Point2f> &t1_left, std::vector<cv::Point2f> &t1_right,
                       std::vector<cv::Point2f> &t2_left, std::vector<unsigned char> &inlier);

    // additive: batched tracking of host-resident frames.  The context is
    // (re)created for `max_batch` frame pairs per launch; frames travel with svo_upload_frames and
    // TrackUploaded runs svo_track_uploaded on device buffer `buf`, appends the n_frames - 1 step
    // records and advances frame_pose_ exactly as n_frames - 1 AddFrame calls would.
    bool EnsureBatchContext(int width, int height, int max_batch) { return EnsureContext(width, height, max_batch); }
    svo_ctx *Context() { return ctx_; }
    // pairs of the oldest outstanding async batch when its records are complete (CollectUploaded will not wait), else 0
    int ResultsReady();
    int Outstanding() const { return (int)(async_tail_ - async_head_); }
    bool TrackUploaded(int buf, int n_frames, std::vector<svo_step_result> &out);
    // the same in two halves: launch without waiting for the GPU, then (after the caller has decoded and
    // uploaded the next chunk, whose copy then overlaps this batch's kernels) collect the records
    // (up to two chunks may be outstanding, collected in launch order; continue_chain seeds the pose chain on
    // the device with the previous chunk's last pose, so a chunk can be launched before its predecessor's
    // records have come back)
    bool TrackUploadedAsync(int buf, int n_frames, bool continue_chain = false);
    bool CollectUploaded(std::vector<svo_step_result> &out);

private:
    bool StereoInit_f2f();
    bool Track();
    bool LK_StereoF2F_PnP_Track();
    bool ORB_StereoF2F_PnP_Track();
    bool TrackOnGpu();
    void FillFeatures();
    void Readparameter();
    bool EnsureContext(int width, int height, int max_batch = 1);

    TrackingStatus status_ = TrackingStatus::INITING;
    Frame::Ptr current_frame_ = nullptr, last_frame_ = nullptr;
    Sensors::Ptr sensors_ = nullptr;
    System *system_ = nullptr;
    Parameter::Ptr parameter_ = nullptr;
    Pose4x4 frame_pose_;
    double Px_ = 0, Py_ = 0, Pz_ = 0;

    svo_ctx *ctx_ = nullptr;
    int device_ = 0;
    long max_keypoints_key_ = 0;                             // additive YAML key max_keypoints, read once (0 = absent)
    int fast_keep_strongest_ = 0;                            // additive YAML key fast_keep_strongest (0 = every corner)
    int lk_accum_ = SVO_LK_ACCUM_EXACT;                      // additive YAML key lk_accum: exact (default) | sse2 | simd128
    int ctx_w_ = 0, ctx_h_ = 0, ctx_batch_ = 0;
    int async_pairs_[2] = {0, 0};
    unsigned async_head_ = 0, async_tail_ = 0;
    svo_step_result last_;
    bool fill_features_ = false;

    // Readparameter()
    int num_features_ = 200, num_features_init_ = 100, num_features_tracking_ = 50;
    int num_features_tracking_bad_ = 20, num_features_needed_for_keyframe_ = 80, init_landmarks_ = 5;
    double feature_match_error_ = 10, inlier_rate_ = 0.5;
    int iterationsCount_ = 500;
    float reprojectionError_ = 0.5f, confidence_ = 0.999f;
    double minmove_ = 0.01, maxmove_ = 0.01;
    std::string track_mode_ = "stereoicp_f2f";
    int nFeatures_ = 0, nLevels_ = 0, fIniThFAST_ = 0, fMinThFAST_ = 0;
    float fScaleFactor_ = 0;
};

}  // namespace lzb_vio
#endif

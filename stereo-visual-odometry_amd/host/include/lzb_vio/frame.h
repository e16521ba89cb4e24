// frame.h -- one stereo frame as the caller hands it to Tracking::AddFrame / System::Step_ros.
//
// Public fields follow the reference's lzb_vio::Frame (include/lzb_vio/frame.h:20-65) so that a
// caller filling left_img_ / right_img_ and reading the carriers afterwards needs no change.  The
// images are host buffers (cv::Mat stand-in of common_include.h); everything derived from them
// lives on the GPU and is copied back into features_* / *_Descriptors_ only on request
// (Tracking::SetFillFeatures).  pose_ is a 16-double POD instead of Sophus::SE3d: the reference's
// Tracking never writes it.
#pragma once
#ifndef lzb_vio_FRAME_H
#define lzb_vio_FRAME_H
#include "lzb_vio/common_include.h"

namespace lzb_vio {
struct Feature;

struct Frame {
    typedef std::shared_ptr<Frame> Ptr;

    // ---- input: the rectified 8-bit stereo pair
    cv::Mat left_img_, right_img_;

    // ---- identity / bookkeeping (CreateFrame numbers the frames 0, 1, 2, ...)
    unsigned long id_ = 0;
    unsigned long keyframe_id_ = 0;
    bool is_keyframe_ = false;
    double time_stamp_ = 0;

    // ---- optional read-back of what the detectors found on this frame
    std::vector<std::shared_ptr<Feature>> features_left_, features_right_;
    cv::Mat left_Descriptors_, right_Descriptors_;          // ORB mode: rows x 32 bytes
    std::vector<unsigned char> status_;

    Frame() = default;
    // reference include/lzb_vio/frame.h:42-43 / src/frame.cpp:25-26 (pose: 16 doubles instead of Sophus::SE3d)
    Frame(long id, double time_stamp, const Pose4x4 &pose, const cv::Mat &left, const cv::Mat &right);
    static std::shared_ptr<Frame> CreateFrame();
    void SetKeyFrame();

    Pose4x4 Pose() { std::lock_guard<std::mutex> hold(pose_mutex_); return pose_; }
    void SetPose(const Pose4x4 &p) { std::lock_guard<std::mutex> hold(pose_mutex_); pose_ = p; }

    Pose4x4 pose_;
    std::mutex pose_mutex_;
};

}  // namespace lzb_vio
#endif

// frame.h -- stereo frame carrier (reference include/lzb_vio/frame.h:20-65).  Same public fields;
// pose_ is a 16-double POD instead of Sophus::SE3d (the reference's Tracking never writes it).
#pragma once
#ifndef lzb_vio_FRAME_H
#define lzb_vio_FRAME_H
#include "lzb_vio/common_include.h"

namespace lzb_vio {
struct Feature;

struct Frame {
    typedef std::shared_ptr<Frame> Ptr;
    unsigned long id_ = 0;
    unsigned long keyframe_id_ = 0;
    bool is_keyframe_ = false;
    double time_stamp_ = 0;
    Pose4x4 pose_;
    std::mutex pose_mutex_;
    cv::Mat left_img_, right_img_;
    cv::Mat left_Descriptors_, right_Descriptors_;
    std::vector<std::shared_ptr<Feature>> features_left_;
    std::vector<std::shared_ptr<Feature>> features_right_;
    std::vector<unsigned char> status_;

    Frame() {}
    Pose4x4 Pose() { std::unique_lock<std::mutex> lck(pose_mutex_); return pose_; }
    void SetPose(const Pose4x4 &p) { std::unique_lock<std::mutex> lck(pose_mutex_); pose_ = p; }
    void SetKeyFrame();
    static std::shared_ptr<Frame> CreateFrame();
};
}  // namespace lzb_vio
#endif

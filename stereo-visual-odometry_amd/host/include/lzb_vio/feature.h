// feature.h -- per-keypoint carrier of the host mirror.
//
// The reference hangs one heap-allocated lzb_vio::Feature off Frame::features_left_ for every
// cv::FAST / ORB keypoint (include/lzb_vio/feature.h:16-33, filled at src/tracking.cpp:104-109 and
// :511-526) and never reads it on the hot path.  Here the keypoints live on the GPU; the carriers
// are only materialised when Tracking::SetFillFeatures(true) (or YAML `fill_features: 1`) asks for
// them, through svo_get_frame_keypoints.  Member names are the reference's so that code reading
// them keeps compiling.
#pragma once
#ifndef lzb_vio_FEATURE_H
#define lzb_vio_FEATURE_H
#include "lzb_vio/common_include.h"

namespace lzb_vio {
struct Frame;

struct Feature {
    typedef std::shared_ptr<Feature> Ptr;

    cv::KeyPoint position_;                 // pixel position, size, angle, response, octave
    cv::Mat Descriptor_;                    // reference feature.h:24; never written by the reference's Tracking
                                            // (descriptors live in Frame::*_Descriptors_): ORB mode fills it with the
                                            // keypoint's 1 x 32 descriptor row when the carriers are requested
    std::weak_ptr<Frame> frame_;            // owner (weak: a frame owns its features, not vice versa)
    bool is_on_left_image_ = true;          // false for ORB keypoints of the right image
    bool is_outlier_ = false;

    Feature() = default;
    Feature(std::shared_ptr<Frame> owner, const cv::KeyPoint &kp) : position_(kp), frame_(owner) {}
};

}  // namespace lzb_vio
#endif

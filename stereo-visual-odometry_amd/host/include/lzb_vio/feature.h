// feature.h -- 2D feature carrier (reference include/lzb_vio/feature.h:16-33).
#pragma once
#ifndef lzb_vio_FEATURE_H
#define lzb_vio_FEATURE_H
#include "lzb_vio/common_include.h"

namespace lzb_vio {
struct Frame;

struct Feature {
    typedef std::shared_ptr<Feature> Ptr;
    std::weak_ptr<Frame> frame_;
    cv::KeyPoint position_;
    bool is_outlier_ = false;
    bool is_on_left_image_ = true;
    Feature() {}
    Feature(std::shared_ptr<Frame> frame, const cv::KeyPoint &kp) : frame_(frame), position_(kp) {}
};
}  // namespace lzb_vio
#endif

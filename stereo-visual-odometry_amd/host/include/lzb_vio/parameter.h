// parameter.h -- typed copy of the YAML keys (reference include/lzb_vio/parameter.h,
// src/parameter.cpp:7-72).  Same member names; matrices are plain row-major double arrays.
#pragma once
#ifndef lzb_vio_PARAMETER_H
#define lzb_vio_PARAMETER_H

#include "lzb_vio/common_include.h"
#include "lzb_vio/config.h"

namespace lzb_vio {

class Parameter {
public:
    typedef std::shared_ptr<Parameter> Ptr;
    Parameter();

    // stereo rig
    double fx1_, fy1_, cx1_, cy1_, fx2_, fy2_, cx2_, cy2_;
    double K1_[9], K2_[9];
    double t_rl_[3], R_rl_[9];
    double projMatr1_[12], projMatr2_[12];     // P1 = K1 [I|0], P2 = K2 [R|t]

    // tracking
    int num_features_init_, num_features_, num_features_tracking_bad_, num_features_needed_for_keyframe_;
    int init_landmarks_;
    double feature_match_error_;
    std::string track_mode_;
    int num_features_tracking_;
    double inlier_rate_;
    int iterationsCount_;
    float reprojectionError_, confidence_;
    double display_scale_;
    int display_x_, display_y_;
    double maxmove_, minmove_;
    int GFTTDetector_num_;
    int nFeatures_;
    float fScaleFactor_;
    int nLevels_, fIniThFAST_, fMinThFAST_;
    std::string dataset_path_;
};

}  // namespace lzb_vio
#endif

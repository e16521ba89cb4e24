// parameter.h -- typed snapshot of the YAML configuration taken once at start-up; mirrors the
// public members of the reference's lzb_vio::Parameter (include/lzb_vio/parameter.h,
// src/parameter.cpp:7-72).  The scalar members come from the key table parameter_keys.def; the
// derived matrices are plain row-major double arrays (no cv::Mat / Eigen in this build).
#pragma once
#ifndef lzb_vio_PARAMETER_H
#define lzb_vio_PARAMETER_H

#include "lzb_vio/common_include.h"
#include "lzb_vio/config.h"

namespace lzb_vio {

class Parameter {
public:
    typedef std::shared_ptr<Parameter> Ptr;
    Parameter();                               // reads every key of the table from Config

#define LZB_KEY(member, key, type) type member;
#include "lzb_vio/parameter_keys.def"
#undef LZB_KEY

    // derived from the keys above plus t_lr0..2 / R_lr0..8
    double K1_[9], K2_[9];                     // [fx 0 cx; 0 fy cy; 0 0 1]
    double t_rl_[3], R_rl_[9];                 // right camera w.r.t. left
    double projMatr1_[12], projMatr2_[12];     // P1 = K1 [I|0], P2 = K2 [R_rl|t_rl]
};

}  // namespace lzb_vio
#endif

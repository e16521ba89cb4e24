// sensors.h -- sensor platform (reference include/lzb_vio/sensors.h): intrinsics and the two
// projection matrices.  The reference's six coordinate-transform helpers are never called and
// are not mirrored.
#pragma once
#ifndef lzb_vio_SENSORS_H
#define lzb_vio_SENSORS_H
#include "lzb_vio/parameter.h"

namespace lzb_vio {

class Sensors {
public:
    typedef std::shared_ptr<Sensors> Ptr;
    explicit Sensors(Parameter::Ptr parameter);
    double fx1_ = 0, fy1_ = 0, cx1_ = 0, cy1_ = 0;
    double fx2_ = 0, fy2_ = 0, cx2_ = 0, cy2_ = 0;
    double K1_[9], K2_[9], t_rl_[3], R_rl_[9];
    double projMatr1_[12], projMatr2_[12];
};

}  // namespace lzb_vio
#endif

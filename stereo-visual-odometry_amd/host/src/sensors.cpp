// sensors.cpp -- copies the rig description out of Parameter (reference src/sensors.cpp:8-25).
#include "lzb_vio/sensors.h"

namespace lzb_vio {

Sensors::Sensors(Parameter::Ptr p)
{
    fx1_ = p->fx1_; fy1_ = p->fy1_; cx1_ = p->cx1_; cy1_ = p->cy1_;
    fx2_ = p->fx2_; fy2_ = p->fy2_; cx2_ = p->cx2_; cy2_ = p->cy2_;
    memcpy(K1_, p->K1_, sizeof(K1_)); memcpy(K2_, p->K2_, sizeof(K2_));
    memcpy(t_rl_, p->t_rl_, sizeof(t_rl_)); memcpy(R_rl_, p->R_rl_, sizeof(R_rl_));
    memcpy(projMatr1_, p->projMatr1_, sizeof(projMatr1_));
    memcpy(projMatr2_, p->projMatr2_, sizeof(projMatr2_));
}

}  // namespace lzb_vio

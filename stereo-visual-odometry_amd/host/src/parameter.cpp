// parameter.cpp -- loads the key table and derives K1, K2, P1 = K1 [I|0], P2 = K2 [R_rl|t_rl]
// (what reference src/parameter.cpp:7-72 computes with cv::Mat products).
#include "lzb_vio/parameter.h"

namespace lzb_vio {

namespace {

void intrinsics(double fx, double fy, double cx, double cy, double K[9])
{
    for (int i = 0; i < 9; i++) K[i] = 0.0;
    K[0] = fx; K[2] = cx; K[4] = fy; K[5] = cy; K[8] = 1.0;
}

// P (3x4, row-major) = K (3x3) * [R (3x3) | t (3)]
void projection(const double K[9], const double R[9], const double t[3], double P[12])
{
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 4; c++) {
            double acc = 0;
            for (int k = 0; k < 3; k++) acc += K[r * 3 + k] * (c < 3 ? R[k * 3 + c] : t[k]);
            P[r * 4 + c] = acc;
        }
}

}  // namespace

Parameter::Parameter()
{
#define LZB_KEY(member, key, type) member = Config::Get<type>(key);
#include "lzb_vio/parameter_keys.def"
#undef LZB_KEY

    char name[16];
    for (int i = 0; i < 3; i++) { snprintf(name, sizeof(name), "t_lr%d", i); t_rl_[i] = Config::Get<double>(name); }
    for (int i = 0; i < 9; i++) { snprintf(name, sizeof(name), "R_lr%d", i); R_rl_[i] = Config::Get<double>(name); }
    intrinsics(fx1_, fy1_, cx1_, cy1_, K1_);
    intrinsics(fx2_, fy2_, cx2_, cy2_, K2_);
    const double eye[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, origin[3] = {0, 0, 0};
    projection(K1_, eye, origin, projMatr1_);
    projection(K2_, R_rl_, t_rl_, projMatr2_);
}

}  // namespace lzb_vio

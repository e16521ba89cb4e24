// parameter.cpp -- eager copy of the config keys; P1 = K1 [I|0], P2 = K2 [R_rl|t_rl]
// (reference src/parameter.cpp:7-72; key list in SURVEY.md Appendix B).
#include "lzb_vio/parameter.h"

namespace lzb_vio {

static void make_K(double fx, double fy, double cx, double cy, double K[9])
{
    const double k[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1};
    memcpy(K, k, sizeof(k));
}

// P = K [R | t]
static void make_P(const double K[9], const double R[9], const double t[3], double P[12])
{
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += K[i * 3 + k] * R[k * 3 + j];
            P[i * 4 + j] = s;
        }
        double s = 0;
        for (int k = 0; k < 3; k++) s += K[i * 3 + k] * t[k];
        P[i * 4 + 3] = s;
    }
}

Parameter::Parameter()
{
    fx1_ = Config::Get<double>("camera_l.fx"); fy1_ = Config::Get<double>("camera_l.fy");
    cx1_ = Config::Get<double>("camera_l.cx"); cy1_ = Config::Get<double>("camera_l.cy");
    fx2_ = Config::Get<double>("camera_r.fx"); fy2_ = Config::Get<double>("camera_r.fy");
    cx2_ = Config::Get<double>("camera_r.cx"); cy2_ = Config::Get<double>("camera_r.cy");
    make_K(fx1_, fy1_, cx1_, cy1_, K1_);
    make_K(fx2_, fy2_, cx2_, cy2_, K2_);
    for (int i = 0; i < 3; i++) t_rl_[i] = Config::Get<double>("t_lr" + std::to_string(i));
    for (int i = 0; i < 9; i++) R_rl_[i] = Config::Get<double>("R_lr" + std::to_string(i));
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, z3[3] = {0, 0, 0};
    make_P(K1_, I3, z3, projMatr1_);
    make_P(K2_, R_rl_, t_rl_, projMatr2_);

    num_features_init_ = Config::Get<int>("num_features_init");
    num_features_ = Config::Get<int>("num_features");
    num_features_tracking_bad_ = Config::Get<int>("num_features_tracking_bad");
    num_features_needed_for_keyframe_ = Config::Get<int>("num_features_needed_for_keyframe");
    init_landmarks_ = Config::Get<int>("init_landmarks");
    feature_match_error_ = Config::Get<double>("feature_match_error");
    track_mode_ = Config::Get<std::string>("track_mode");
    num_features_tracking_ = Config::Get<int>("num_features_tracking");
    inlier_rate_ = Config::Get<double>("inlier_rate");
    iterationsCount_ = Config::Get<int>("iterationsCount");
    reprojectionError_ = Config::Get<float>("reprojectionError");
    confidence_ = Config::Get<float>("confidence");
    display_scale_ = Config::Get<double>("display_scale");
    display_x_ = Config::Get<int>("display_x");
    display_y_ = Config::Get<int>("display_y");
    maxmove_ = Config::Get<double>("maxmove");
    minmove_ = Config::Get<double>("minmove");
    GFTTDetector_num_ = Config::Get<int>("num_features");
    nFeatures_ = Config::Get<int>("nFeatures");
    fScaleFactor_ = Config::Get<float>("fScaleFactor");
    nLevels_ = Config::Get<int>("nLevels");
    fIniThFAST_ = Config::Get<int>("fIniThFAST");
    fMinThFAST_ = Config::Get<int>("fMinThFAST");
    dataset_path_ = Config::Get<std::string>("dataset_path");
}

}  // namespace lzb_vio

// frame.cpp -- id factories (reference src/frame.cpp:28-41).
#include "lzb_vio/frame.h"

namespace lzb_vio {

Frame::Frame(long id, double time_stamp, const Pose4x4 &pose, const cv::Mat &left, const cv::Mat &right)
    : left_img_(left), right_img_(right), id_((unsigned long)id), time_stamp_(time_stamp), pose_(pose) {}

Frame::Ptr Frame::CreateFrame()
{
    static long factory_id = 0;
    Frame::Ptr f(new Frame);
    f->id_ = factory_id++;
    return f;
}

void Frame::SetKeyFrame()
{
    static long keyframe_factory_id = 0;
    is_keyframe_ = true;
    keyframe_id_ = keyframe_factory_id++;
}

}  // namespace lzb_vio

// frame.cpp -- id factories (reference src/frame.cpp:28-41).
#include "lzb_vio/frame.h"
#include <atomic>

namespace lzb_vio {

Frame::Frame(long id, double time_stamp, const Pose4x4 &pose, const cv::Mat &left, const cv::Mat &right)
    : left_img_(left), right_img_(right), id_((unsigned long)id), time_stamp_(time_stamp), pose_(pose) {}

Frame::Ptr Frame::CreateFrame()
{
    static std::atomic<long> factory_id(0);                  // several System objects may run on their own threads
    Frame::Ptr f(new Frame);
    f->id_ = (unsigned long)factory_id.fetch_add(1);
    return f;
}

void Frame::SetKeyFrame()
{
    static std::atomic<long> keyframe_factory_id(0);
    is_keyframe_ = true;
    keyframe_id_ = (unsigned long)keyframe_factory_id.fetch_add(1);
}

}  // namespace lzb_vio

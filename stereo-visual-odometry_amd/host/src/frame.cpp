// frame.cpp -- id factories (reference src/frame.cpp:28-41).
#include "lzb_vio/frame.h"

namespace lzb_vio {

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

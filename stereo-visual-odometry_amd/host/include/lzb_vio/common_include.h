// common_include.h -- minimal value types for the host-side mirror of the reference's lzb_vio API.
// The reference's include hub (include/lzb_vio/common_include.h) pulls in OpenCV, Eigen, Sophus and
// glog; none of them exist here, so this header supplies the few carrier types the public surface
// needs: a ref-counted 8-bit image with cv::Mat's data/rows/cols/step fields, Point2f, KeyPoint
// and a 16-double pose.  If real OpenCV is available a maintainer can alias these to the cv:: ones.
#pragma once
#ifndef lzb_vio_COMMON_INCLUDE_H
#define lzb_vio_COMMON_INCLUDE_H

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <list>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace cv {

struct Point2f {
    float x = 0.f, y = 0.f;
    Point2f() {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

struct Point2i {
    int x = 0, y = 0;
    Point2i() {}
    Point2i(int x_, int y_) : x(x_), y(y_) {}
};
typedef Point2i Point;

// field order of cv::KeyPoint == svo_keypoint
struct KeyPoint {
    Point2f pt;
    float size = 0.f, angle = -1.f, response = 0.f;
    int octave = 0, class_id = -1;
};

// 8-bit single-channel image; copies share the pixel buffer like cv::Mat headers do
class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t *data = nullptr;
    Mat() {}
    Mat(int r, int c) { create(r, c); }
    void create(int r, int c)
    {
        rows = r; cols = c; step = (size_t)((c + 63) / 64 * 64);
        buf_.reset(new uint8_t[step * (size_t)r], std::default_delete<uint8_t[]>());
        data = buf_.get();
    }
    bool empty() const { return data == nullptr; }
    uint8_t *ptr(int r) { return data + step * (size_t)r; }
    const uint8_t *ptr(int r) const { return data + step * (size_t)r; }
private:
    std::shared_ptr<uint8_t> buf_;
};

// the reference's ORBextractor::operator() is declared with cv::InputArray / cv::OutputArray
// (include/lzb_vio/ORBextractor.h:40-42); with the Mat stand-in they are plain references
typedef const Mat &InputArray;
typedef Mat &OutputArray;

}  // namespace cv

namespace lzb_vio {
// the reference keeps frame_pose_ as a 4x4 CV_64F cv::Mat (include/lzb_vio/tracking.h:117)
struct Pose4x4 {
    double m[16];
    Pose4x4() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; }
};
}  // namespace lzb_vio

// LZB_VIO_TIMING=1: seconds since the process started, at the phases of a run (where a short run's wall time goes)
namespace lzb_vio { double lzb_seconds_since_start(); }
#define LZB_PHASE(name) do { if (getenv("LZB_VIO_TIMING")) fprintf(stderr, "[TIMING] %8.4f s  %s\n", lzb_vio::lzb_seconds_since_start(), name); } while (0)
#define LZB_LOG(level, ...) do { fprintf(stderr, "[" level "] " __VA_ARGS__); fprintf(stderr, "\n"); } while (0)

#endif

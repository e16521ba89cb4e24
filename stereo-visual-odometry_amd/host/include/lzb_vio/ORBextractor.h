// ORBextractor.h -- host-side mirror of lzb_vio::ORBextractor (reference
// include/lzb_vio/ORBextractor.h:24-107, src/ORBextractor.cpp:359-418, 990-1085).
//
// Same public surface: the five-argument constructor, operator()(image, mask, keypoints, descriptors),
// the six getters and the public mvImagePyramid.  The extraction itself -- 8-level bilinear pyramid,
// per-cell FAST with threshold fallback, quadtree distribution, intensity-centroid orientation, 7x7
// blur, rotated BRIEF -- runs on the GPU behind svo_orb_extract (include/svo_abi.h); there is no CPU
// implementation behind this class.  The protected ORB-SLAM2 stages (ComputePyramid,
// ComputeKeyPointsOctTree, DistributeOctTree, ComputeKeyPointsOld) and the ExtractorNode helper type
// are therefore not part of the mirror; the protected data members a subclass could read are kept.
//
// Additive: Ok() / LastError() (the reference aborts inside OpenCV on a bad image; this returns an
// empty keypoint set and keeps the message), SetKeepPyramid(false) to skip the device-to-host copy
// that refreshes mvImagePyramid on every call.
#pragma once
#ifndef lzb_vio_ORBEXTRACTOR_H
#define lzb_vio_ORBEXTRACTOR_H

#include "lzb_vio/common_include.h"
#include "svo_abi.h"

namespace lzb_vio {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST);
    ~ORBextractor();
    ORBextractor(const ORBextractor &) = delete;
    ORBextractor &operator=(const ORBextractor &) = delete;

    // Compute the ORB features and descriptors on an image (mask is ignored, as in the reference).
    // keypoints: level-0 coordinates, size 31 * scale, angle in degrees, response = FAST score, octave;
    // descriptors: keypoints.size() rows of 32 bytes.
    void operator()(cv::InputArray image, cv::InputArray mask, std::vector<cv::KeyPoint> &keypoints,
                    cv::OutputArray descriptors);

    int GetLevels() { return nlevels; }
    float GetScaleFactor() { return (float)scaleFactor; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // the pyramid of the LAST image given to operator(): Tracking calls one extractor for the left and
    // then the right image (src/tracking.cpp:508-509), so afterwards this is the right image's pyramid
    std::vector<cv::Mat> mvImagePyramid;

    // additive
    bool Ok() const { return err_.empty(); }
    const std::string &LastError() const { return err_; }
    void SetKeepPyramid(bool on) { keep_pyramid_ = on; }
    const std::vector<int> &FeaturesPerLevel() const { return mnFeaturesPerLevel; }

protected:
    int nfeatures;
    double scaleFactor;
    int nlevels;
    int iniThFAST;
    int minThFAST;
    std::vector<int> mnFeaturesPerLevel;
    std::vector<int> umax;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;

private:
    bool EnsureContext(int width, int height);
    svo_ctx *ctx_ = nullptr;
    int ctx_w_ = 0, ctx_h_ = 0;
    bool keep_pyramid_ = true;
    std::string err_;
    std::vector<svo_keypoint> kp_buf_;
    std::vector<uint8_t> desc_buf_;
};

}  // namespace lzb_vio
#endif

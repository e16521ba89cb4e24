// ORBextractor.h -- host-side mirror of lzb_vio::ORBextractor (reference
// include/lzb_vio/ORBextractor.h:24-107, src/ORBextractor.cpp:359-418, 990-1085).
//
// Same public surface: the five-argument constructor, operator()(image, mask, keypoints, descriptors),
// the six getters and the public mvImagePyramid.  The extraction itself -- 8-level bilinear pyramid,
// per-cell FAST with threshold fallback, quadtree distribution, intensity-centroid orientation, 7x7
// blur, rotated BRIEF -- runs on the GPU behind svo_orb_extract (include/svo_abi.h); there is no CPU
// implementation of the extraction behind this class.
//
// The protected ORB-SLAM2 stages a subclass of the reference class may call (include/lzb_vio/ORBextractor.h:77-85)
// are kept so that such a subclass compiles and behaves (round 4):
//   ComputePyramid(image)            fills mvImagePyramid (it runs the extraction on the GPU and keeps its result)
//   ComputeKeyPointsOctTree(all)     the keypoints of the image last given to ComputePyramid, per level, in LEVEL
//                                    coordinates, oriented -- what src/ORBextractor.cpp:717-807 leaves in allKeypoints
//   DistributeOctTree(...)           ORB-SLAM2's quadtree on the HOST for the caller's own keypoints (the extraction
//                                    itself distributes on the GPU); ties of the "largest node first" order are broken
//                                    by creation order, where the reference sorts by heap address
//   ComputeKeyPointsOld(all)         NOT built (the reference never calls it, src/ORBextractor.cpp:1006): forwards to
//                                    ComputeKeyPointsOctTree and says so once
// and so are the ExtractorNode helper type and the protected data members (`pattern` = the 512 rBRIEF test points).
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

// node of DistributeOctTree's quadtree (reference include/lzb_vio/ORBextractor.h:11-23)
class ExtractorNode {
public:
    ExtractorNode() : bNoMore(false) {}
    void DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4);

    std::vector<cv::KeyPoint> vKeys;
    cv::Point2i UL, UR, BL, BR;
    std::list<ExtractorNode>::iterator lit;
    bool bNoMore;
};

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
    void ComputePyramid(cv::Mat image);
    void ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint>> &allKeypoints);
    std::vector<cv::KeyPoint> DistributeOctTree(const std::vector<cv::KeyPoint> &vToDistributeKeys, const int &minX, const int &maxX,
                                                const int &minY, const int &maxY, const int &nFeatures, const int &level);
    void ComputeKeyPointsOld(std::vector<std::vector<cv::KeyPoint>> &allKeypoints);
    std::vector<cv::Point> pattern;

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
    std::vector<cv::KeyPoint> last_keys_;      // of the image last given to ComputePyramid (ComputeKeyPointsOctTree splits them)
    bool have_last_keys_ = false;
};

}  // namespace lzb_vio
#endif

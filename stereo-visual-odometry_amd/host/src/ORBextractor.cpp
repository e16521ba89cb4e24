// ORBextractor.cpp -- lzb_vio::ORBextractor on top of the HIP library (reference
// src/ORBextractor.cpp:359-418 constructor tables, :990-1055 operator()).
#include "lzb_vio/ORBextractor.h"

#include <cmath>

namespace lzb_vio {

// The scale / sigma tables and the per-level feature quotas of the constructor.  They are plain
// float recurrences; the device side builds the same ones for its own use (csrc/orb.hip
// orb_make_geom), these copies only serve the getters.
ORBextractor::ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
    : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_)
{
    const int L = nlevels > 0 ? nlevels : 0;
    mvScaleFactor.assign(L, 1.0f);
    mvLevelSigma2.assign(L, 1.0f);
    mvInvScaleFactor.assign(L, 1.0f);
    mvInvLevelSigma2.assign(L, 1.0f);
    for (int l = 1; l < L; l++) {
        mvScaleFactor[l] = (float)(mvScaleFactor[l - 1] * scaleFactor);
        mvLevelSigma2[l] = mvScaleFactor[l] * mvScaleFactor[l];
    }
    for (int l = 0; l < L; l++) {
        mvInvScaleFactor[l] = 1.0f / mvScaleFactor[l];
        mvInvLevelSigma2[l] = 1.0f / mvLevelSigma2[l];
    }
    mvImagePyramid.resize(L);
    // geometric split of nfeatures over the levels, the remainder goes to the last one
    mnFeaturesPerLevel.assign(L, 0);
    const float shrink = (float)(1.0f / scaleFactor);
    float want = nfeatures * (1 - shrink) / (1 - (float)std::pow((double)shrink, (double)nlevels));
    int given = 0;
    for (int l = 0; l + 1 < L; l++) {
        mnFeaturesPerLevel[l] = (int)std::lrintf(want);
        given += mnFeaturesPerLevel[l];
        want *= shrink;
    }
    if (L > 0) mnFeaturesPerLevel[L - 1] = nfeatures - given > 0 ? nfeatures - given : 0;
    // row ends of the radius-15 circular patch, made symmetric about the diagonal
    const int half = 15;
    umax.assign(half + 1, 0);
    const int vmax = (int)std::floor(half * std::sqrt(2.f) / 2 + 1), vmin = (int)std::ceil(half * std::sqrt(2.f) / 2);
    for (int v = 0; v <= vmax; ++v) umax[v] = (int)std::lrint(std::sqrt((double)half * half - v * v));
    for (int v = half, v0 = 0; v >= vmin; --v) {
        while (umax[v0] == umax[v0 + 1]) ++v0;
        umax[v] = v0++;
    }
}

ORBextractor::~ORBextractor()
{
    if (ctx_) svo_destroy(ctx_);
}

// One HIP context per image size, created on first use (the reference's extractor learns the size
// from the cv::Mat it is given, too).
bool ORBextractor::EnsureContext(int width, int height)
{
    if (ctx_ && width == ctx_w_ && height == ctx_h_) return true;
    if (ctx_) { svo_destroy(ctx_); ctx_ = nullptr; }
    svo_config cfg;
    svo_default_config(&cfg, width, height);
    cfg.track_mode = SVO_MODE_ORB;
    cfg.orb_nfeatures = nfeatures; cfg.orb_scale_factor = (float)scaleFactor; cfg.orb_nlevels = nlevels;
    cfg.orb_ini_th = iniThFAST; cfg.orb_min_th = minThFAST;
    // keypoints per image <= nfeatures + a few per level; candidates per level <= 4 * max_keypoints
    int cap = 2 * nfeatures + 64;
    cap = cap < 8192 ? 8192 : cap;
    cfg.max_keypoints = cap > 16384 ? 16384 : cap;
    const int rc = svo_create(&cfg, 0, &ctx_);
    if (rc != SVO_OK) {
        ctx_ = nullptr;
        err_ = "svo_create failed: a HIP device is required (there is no CPU path), or the ORB configuration is unsupported";
        LZB_LOG("ERROR", "ORBextractor: %s (%d)", err_.c_str(), rc);
        return false;
    }
    ctx_w_ = width; ctx_h_ = height;
    kp_buf_.resize((size_t)cfg.max_keypoints);
    desc_buf_.resize((size_t)cfg.max_keypoints * 32);
    return true;
}

void ORBextractor::operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints,
                              cv::OutputArray descriptors)
{
    keypoints.clear();
    descriptors = cv::Mat();
    err_.clear();
    if (image.empty()) return;                                  // "if (_image.empty()) return;" (:994-995)
    if (!EnsureContext(image.cols, image.rows)) return;
    int n = 0;
    const int rc = svo_orb_extract(ctx_, image.data, (int)image.step, SVO_MEM_HOST, kp_buf_.data(), desc_buf_.data(),
                                   (int)kp_buf_.size(), &n, nullptr);
    if (rc != SVO_OK) {
        err_ = svo_last_error(ctx_);
        LZB_LOG("ERROR", "ORBextractor: svo_orb_extract: %s", err_.c_str());
        return;
    }
    keypoints.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const svo_keypoint &k = kp_buf_[(size_t)i];
        cv::KeyPoint &o = keypoints[(size_t)i];
        o.pt.x = k.x; o.pt.y = k.y; o.size = k.size; o.angle = k.angle; o.response = k.response;
        o.octave = k.octave; o.class_id = k.class_id;
    }
    if (n > 0) {
        descriptors.create(n, 32);
        for (int i = 0; i < n; i++) memcpy(descriptors.ptr(i), desc_buf_.data() + (size_t)i * 32, 32);
    }
    if (!keep_pyramid_) return;
    for (int l = 0; l < nlevels; l++) {
        int w = 0, h = 0;
        std::vector<uint8_t> tight;
        bool ok = svo_orb_read_level(ctx_, l, nullptr, &w, &h) == SVO_OK;
        if (ok) {
            tight.resize((size_t)w * h);
            ok = svo_orb_read_level(ctx_, l, tight.data(), &w, &h) == SVO_OK;
        }
        if (!ok) {
            // never leave the PREVIOUS image's levels behind a failed read-back: the public pyramid is either
            // this image's or empty, and Ok() reports the failure
            err_ = svo_last_error(ctx_);
            LZB_LOG("ERROR", "ORBextractor: svo_orb_read_level(%d): %s", l, err_.c_str());
            for (int q = l; q < nlevels; q++) mvImagePyramid[(size_t)q] = cv::Mat();
            return;
        }
        cv::Mat lvl(h, w);
        for (int y = 0; y < h; y++) memcpy(lvl.ptr(y), tight.data() + (size_t)y * w, (size_t)w);
        mvImagePyramid[(size_t)l] = lvl;
    }
}

}  // namespace lzb_vio

// ORBextractor.cpp -- lzb_vio::ORBextractor on top of the HIP library (reference
// src/ORBextractor.cpp:359-418 constructor tables, :990-1055 operator()).
#include "lzb_vio/ORBextractor.h"

#include <algorithm>
#include <cmath>

#include "../../csrc/orb_pattern.h"        // the 256 x 4 rBRIEF table (constant data shared with the device code)

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
    // "std::copy(pattern0, pattern0 + npoints, std::back_inserter(pattern))" (reference src/ORBextractor.cpp:394-396)
    pattern.reserve(512);
    for (int i = 0; i < 512; i++) pattern.push_back(cv::Point(svo_bit_pattern_31[2 * i], svo_bit_pattern_31[2 * i + 1]));
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

// ---- the protected ORB-SLAM2 stages -------------------------------------------------------------------------------

// reference src/ORBextractor.cpp:1061-1085 builds the pyramid on the CPU; here the whole extraction runs on the GPU
// and its keypoints are kept for ComputeKeyPointsOctTree
void ORBextractor::ComputePyramid(cv::Mat image)
{
    const bool keep = keep_pyramid_;
    keep_pyramid_ = true;
    cv::Mat desc;
    (*this)(image, cv::Mat(), last_keys_, desc);
    keep_pyramid_ = keep;
    have_last_keys_ = Ok() && !image.empty();
}

// reference src/ORBextractor.cpp:717-807: per level FAST on cells, DistributeOctTree, then
// "keypoints[i].pt.x += minBorderX; .octave = level; .size = scaledPatchSize" and computeOrientation.  operator()
// returns exactly those keypoints with "keypoint->pt *= scale" applied (:1043-1049): the level coordinates are integers,
// so dividing by the scale and rounding recovers them exactly.
void ORBextractor::ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint>> &allKeypoints)
{
    allKeypoints.assign((size_t)(nlevels > 0 ? nlevels : 0), std::vector<cv::KeyPoint>());
    if (!have_last_keys_) {
        LZB_LOG("ERROR", "ORBextractor::ComputeKeyPointsOctTree: call ComputePyramid(image) first");
        return;
    }
    for (const cv::KeyPoint &k : last_keys_) {
        if (k.octave < 0 || k.octave >= nlevels) continue;
        cv::KeyPoint q = k;
        if (k.octave > 0) {
            q.pt.x = std::rint(k.pt.x / mvScaleFactor[(size_t)k.octave]);
            q.pt.y = std::rint(k.pt.y / mvScaleFactor[(size_t)k.octave]);
        }
        allKeypoints[(size_t)k.octave].push_back(q);
    }
}

void ORBextractor::ComputeKeyPointsOld(std::vector<std::vector<cv::KeyPoint>> &allKeypoints)
{
    static bool said = false;
    if (!said) {
        LZB_LOG("WARNING", "ORBextractor::ComputeKeyPointsOld is not built (the reference never calls it): ComputeKeyPointsOctTree runs instead");
        said = true;
    }
    ComputeKeyPointsOctTree(allKeypoints);
}

// reference src/ORBextractor.cpp:430-485: the four children of a node and the keys that fall into each
void ExtractorNode::DivideNode(ExtractorNode &n1, ExtractorNode &n2, ExtractorNode &n3, ExtractorNode &n4)
{
    const int halfX = (int)std::ceil((float)(UR.x - UL.x) / 2), halfY = (int)std::ceil((float)(BR.y - UL.y) / 2);
    const int midX = UL.x + halfX, midY = UL.y + halfY;
    n1.UL = UL;                       n1.UR = cv::Point2i(midX, UL.y);  n1.BL = cv::Point2i(UL.x, midY);  n1.BR = cv::Point2i(midX, midY);
    n2.UL = n1.UR;                    n2.UR = UR;                       n2.BL = n1.BR;                    n2.BR = cv::Point2i(UR.x, midY);
    n3.UL = n1.BL;                    n3.UR = n1.BR;                    n3.BL = BL;                       n3.BR = cv::Point2i(midX, BL.y);
    n4.UL = n3.UR;                    n4.UR = n2.BR;                    n4.BL = n3.BR;                    n4.BR = BR;
    ExtractorNode *child[4] = {&n1, &n2, &n3, &n4};
    for (ExtractorNode *c : child) c->vKeys.reserve(vKeys.size());
    for (const cv::KeyPoint &kp : vKeys) {
        const int right = kp.pt.x < (float)midX ? 0 : 1, below = kp.pt.y < (float)midY ? 0 : 2;
        child[right + below]->vKeys.push_back(kp);
    }
    for (ExtractorNode *c : child) c->bNoMore = c->vKeys.size() == 1;
}

// reference src/ORBextractor.cpp:487-715.  The list walk of a pass: every node with more than one key is divided, its
// non-empty children go to the FRONT of the list; once the next pass could overshoot N the nodes to expand are taken
// largest first until N nodes exist; every node then keeps its best-response key.
std::vector<cv::KeyPoint> ORBextractor::DistributeOctTree(const std::vector<cv::KeyPoint> &vToDistributeKeys, const int &minX,
                                                          const int &maxX, const int &minY, const int &maxY, const int &N,
                                                          const int & /*level*/)
{
    std::vector<cv::KeyPoint> result;
    if (maxY <= minY || maxX <= minX || vToDistributeKeys.empty()) return result;
    const int nIni = (int)std::round((float)(maxX - minX) / (float)(maxY - minY));
    if (nIni <= 0) return result;
    const float hX = (float)(maxX - minX) / (float)nIni;
    std::list<ExtractorNode> nodes;
    std::vector<ExtractorNode *> roots((size_t)nIni);
    for (int i = 0; i < nIni; i++) {
        ExtractorNode ni;
        ni.UL = cv::Point2i((int)(hX * (float)i), 0);
        ni.UR = cv::Point2i((int)(hX * (float)(i + 1)), 0);
        ni.BL = cv::Point2i(ni.UL.x, maxY - minY);
        ni.BR = cv::Point2i(ni.UR.x, maxY - minY);
        ni.vKeys.reserve(vToDistributeKeys.size());
        nodes.push_back(ni);
        roots[(size_t)i] = &nodes.back();
    }
    for (const cv::KeyPoint &kp : vToDistributeKeys) {
        const int strip = (int)(kp.pt.x / hX);
        if (strip >= 0 && strip < nIni) roots[(size_t)strip]->vKeys.push_back(kp);
    }
    for (auto it = nodes.begin(); it != nodes.end();) {
        if (it->vKeys.size() == 1) { it->bNoMore = true; ++it; }
        else if (it->vKeys.empty()) it = nodes.erase(it);
        else ++it;
    }
    // (keys, creation number, node): the creation number breaks ties where the reference compares heap addresses
    struct ToExpand { int size; long seq; ExtractorNode *node; };
    long created = 0;
    std::vector<ToExpand> expand;
    auto add_children = [&](ExtractorNode (&c)[4], int *n_big) {
        for (ExtractorNode &ch : c) {
            if (ch.vKeys.empty()) continue;
            nodes.push_front(ch);
            nodes.front().lit = nodes.begin();
            if (ch.vKeys.size() > 1) {
                if (n_big) ++*n_big;
                expand.push_back(ToExpand{(int)ch.vKeys.size(), created, &nodes.front()});
            }
            ++created;
        }
    };
    bool finish = false;
    while (!finish) {
        const size_t prev_size = nodes.size();
        int n_to_expand = 0;
        expand.clear();
        for (auto it = nodes.begin(); it != nodes.end();) {
            if (it->bNoMore) { ++it; continue; }
            ExtractorNode c[4];
            it->DivideNode(c[0], c[1], c[2], c[3]);
            add_children(c, &n_to_expand);
            it = nodes.erase(it);
        }
        if ((int)nodes.size() >= N || nodes.size() == prev_size) finish = true;
        else if ((int)nodes.size() + n_to_expand * 3 > N) {
            while (!finish) {
                const size_t before = nodes.size();
                std::vector<ToExpand> todo;
                todo.swap(expand);
                std::sort(todo.begin(), todo.end(), [](const ToExpand &a, const ToExpand &b) {
                    return a.size != b.size ? a.size < b.size : a.seq < b.seq;
                });
                for (size_t j = todo.size(); j-- > 0;) {
                    ExtractorNode c[4];
                    todo[j].node->DivideNode(c[0], c[1], c[2], c[3]);
                    add_children(c, nullptr);
                    nodes.erase(todo[j].node->lit);
                    if ((int)nodes.size() >= N) break;
                }
                if ((int)nodes.size() >= N || nodes.size() == before) finish = true;
            }
        }
    }
    result.reserve(nodes.size());
    for (const ExtractorNode &nd : nodes) {
        const cv::KeyPoint *best = &nd.vKeys[0];
        for (const cv::KeyPoint &kp : nd.vKeys)
            if (kp.response > best->response) best = &kp;
        result.push_back(*best);
    }
    return result;
}

}  // namespace lzb_vio

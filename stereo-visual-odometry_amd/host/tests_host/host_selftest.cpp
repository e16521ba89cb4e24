// host_selftest -- checks of the host mirror for pytest.
//   host_selftest config.yaml image1 [image2 ...]      CPU only: YAML surface and image readers
//   host_selftest --track config.yaml n_frames         GPU: Step_ros over <dataset_path>/image_{0,1}/%06d.{png,pgm}
//                                                       with the per-frame carriers filled (fill_features)
//   host_selftest --orb config.yaml left right          GPU: one lzb_vio::ORBextractor called on the left and then the
//                                                       right image, as Tracking::Detect_MyORBFeatures does
//                                                       (reference src/tracking.cpp:508-509); prints byte hashes of
//                                                       the keypoints, descriptors and of mvImagePyramid
#include "lzb_vio/ORBextractor.h"
#include "lzb_vio/System.h"
#include <cmath>

static unsigned long long hash_bytes(const void *p, size_t n, unsigned long long h = 0)
{
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) h = h * 31 + b[i];
    return h;
}

// a subclass of the extractor, as a user of the reference class could write one: it reaches the protected stages
struct OpenExtractor : lzb_vio::ORBextractor {
    using lzb_vio::ORBextractor::ORBextractor;
    using lzb_vio::ORBextractor::ComputePyramid;
    using lzb_vio::ORBextractor::ComputeKeyPointsOctTree;
    using lzb_vio::ORBextractor::ComputeKeyPointsOld;
    using lzb_vio::ORBextractor::DistributeOctTree;
    using lzb_vio::ORBextractor::pattern;
    using lzb_vio::ORBextractor::umax;
};

// host_selftest --quadtree keys.txt minX maxX minY maxY N : DistributeOctTree (host code, no GPU) on "x y response" lines
static int quadtree_mode(int argc, char **argv)
{
    if (argc < 8) return 2;
    FILE *f = fopen(argv[2], "r");
    if (!f) return 3;
    std::vector<cv::KeyPoint> keys;
    float x, y, r;
    while (fscanf(f, "%f %f %f", &x, &y, &r) == 3) {
        cv::KeyPoint k;
        k.pt.x = x; k.pt.y = y; k.response = r; k.class_id = (int)keys.size();      // class_id carries the input index
        keys.push_back(k);
    }
    fclose(f);
    OpenExtractor ex(2000, 1.2f, 8, 20, 7);
    printf("pattern=%zu first=%d,%d last=%d,%d umax0=%d umax15=%d\n", ex.pattern.size(), ex.pattern[0].x, ex.pattern[0].y,
           ex.pattern[511].x, ex.pattern[511].y, ex.umax[0], ex.umax[15]);
    const std::vector<cv::KeyPoint> out = ex.DistributeOctTree(keys, atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), atoi(argv[6]), atoi(argv[7]), 0);
    printf("selected %zu:", out.size());
    for (const auto &k : out) printf(" %d", k.class_id);
    printf("\n");
    return 0;
}

static int orb_mode(int argc, char **argv)
{
    if (argc < 5) return 2;
    if (!lzb_vio::Config::SetParameterFile(argv[2])) return 3;
    lzb_vio::Parameter p;
    lzb_vio::ORBextractor ex(p.nFeatures_, p.fScaleFactor_, p.nLevels_, p.fIniThFAST_, p.fMinThFAST_);
    printf("levels=%d scale=%.6f", ex.GetLevels(), ex.GetScaleFactor());
    for (float v : ex.GetScaleFactors()) printf(" %.9g", v);
    for (float v : ex.GetInverseScaleSigmaSquares()) printf(" %.9g", v);
    printf(" quota");
    for (int q : ex.FeaturesPerLevel()) printf(" %d", q);
    printf("\n");
    static_assert(sizeof(cv::KeyPoint) == 28, "cv::KeyPoint stand-in must be the 28-byte record");
    for (int i = 3; i <= 4; i++) {
        cv::Mat img;
        if (!lzb_vio::ReadImageGray(argv[i], img)) return 4;
        std::vector<cv::KeyPoint> kps;
        cv::Mat desc;
        ex(img, cv::Mat(), kps, desc);
        if (!ex.Ok()) { fprintf(stderr, "ORBextractor: %s\n", ex.LastError().c_str()); return 5; }
        unsigned long long hd = 0;
        for (int r = 0; r < desc.rows; r++) hd = hash_bytes(desc.ptr(r), 32, hd);
        printf("image%d n=%zu rows=%d cols=%d kp_hash=%llu desc_hash=%llu\n", i - 2, kps.size(), desc.rows, desc.cols,
               hash_bytes(kps.data(), kps.size() * sizeof(cv::KeyPoint)), hd);
    }
    for (size_t l = 0; l < ex.mvImagePyramid.size(); l++) {
        const cv::Mat &m = ex.mvImagePyramid[l];
        unsigned long long h = 0;
        for (int y = 0; y < m.rows; y++) h = hash_bytes(m.ptr(y), (size_t)m.cols, h);
        printf("pyramid%zu rows=%d cols=%d hash=%llu\n", l, m.rows, m.cols, h);
    }
    // the protected stages through a subclass: ComputePyramid + ComputeKeyPointsOctTree on the LEFT image give its keypoints
    // per level in level coordinates; scaled back they are operator()'s keypoints bit for bit
    {
        OpenExtractor ox(p.nFeatures_, p.fScaleFactor_, p.nLevels_, p.fIniThFAST_, p.fMinThFAST_);
        cv::Mat img;
        if (!lzb_vio::ReadImageGray(argv[3], img)) return 4;
        std::vector<std::vector<cv::KeyPoint>> all, old;
        ox.ComputePyramid(img);
        ox.ComputeKeyPointsOctTree(all);
        ox.ComputeKeyPointsOld(old);
        std::vector<cv::KeyPoint> ref;
        cv::Mat d;
        ox(img, cv::Mat(), ref, d);
        size_t at = 0;
        bool same = old.size() == all.size();
        printf("octree levels=%zu counts", all.size());
        for (size_t l = 0; l < all.size(); l++) {
            printf(" %zu", all[l].size());
            const float sc = ox.GetScaleFactors()[l];
            for (const cv::KeyPoint &k : all[l]) {
                const bool integral = k.pt.x == std::floor(k.pt.x) && k.pt.y == std::floor(k.pt.y);
                const float bx = l ? k.pt.x * sc : k.pt.x, by = l ? k.pt.y * sc : k.pt.y;
                same = same && at < ref.size() && integral && k.octave == (int)l && bx == ref[at].pt.x && by == ref[at].pt.y &&
                       k.angle == ref[at].angle && k.response == ref[at].response && k.size == ref[at].size;
                at++;
            }
        }
        printf(" total=%zu of %zu same=%d pyramid0=%dx%d\n", at, ref.size(), (int)(same && at == ref.size()),
               ox.mvImagePyramid[0].cols, ox.mvImagePyramid[0].rows);
    }
    // an empty image returns an empty set (src/ORBextractor.cpp:994-995)
    std::vector<cv::KeyPoint> none;
    cv::Mat nodesc;
    ex(cv::Mat(), cv::Mat(), none, nodesc);
    printf("empty n=%zu rows=%d\n", none.size(), nodesc.rows);
    return 0;
}

static int track_mode(int argc, char **argv)
{
    if (argc < 4) return 2;
    std::string cfg = argv[2];
    lzb_vio::System vo(cfg);
    vo.GetTracking()->SetFillFeatures(true);
    lzb_vio::Parameter p;
    const int n = atoi(argv[3]);
    for (int i = 0; i < n; i++) {
        auto f = lzb_vio::Frame::CreateFrame();
        char name[64];
        bool ok = true;
        for (int cam = 0; cam < 2 && ok; cam++) {
            snprintf(name, sizeof(name), "/image_%d/%06d.png", cam, i);
            ok = lzb_vio::ReadImageGray(p.dataset_path_ + name, cam == 0 ? f->left_img_ : f->right_img_);
            if (!ok) {
                snprintf(name, sizeof(name), "/image_%d/%06d.pgm", cam, i);
                ok = lzb_vio::ReadImageGray(p.dataset_path_ + name, cam == 0 ? f->left_img_ : f->right_img_);
            }
        }
        if (!ok) return 4;
        bool good = vo.Step_ros(f);
        std::vector<cv::Point2f> a, b, c;
        std::vector<unsigned char> in;
        vo.GetTracking()->GetLastTracks(a, b, c, in);
        int n_in = 0;
        for (unsigned char v : in) n_in += v != 0;
        double sx = 0;
        for (auto &ft : f->features_left_) sx += ft->position_.pt.x + 2.0 * ft->position_.pt.y + ft->position_.response;
        printf("frame=%d good=%d featuresL=%zu featuresR=%zu descL=%d descR=%d tracks=%zu inliers=%d n_tracked=%d n_inliers=%d sumL=%.3f\n",
               i, good ? 1 : 0, f->features_left_.size(), f->features_right_.size(), f->left_Descriptors_.rows,
               f->right_Descriptors_.rows, a.size(), n_in, vo.GetTracking()->LastResult().n_tracked,
               vo.GetTracking()->LastResult().n_inliers, sx);
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    if (std::string(argv[1]) == "--track") return track_mode(argc, argv);
    if (std::string(argv[1]) == "--orb") return orb_mode(argc, argv);
    if (std::string(argv[1]) == "--quadtree") return quadtree_mode(argc, argv);
    if (!lzb_vio::Config::SetParameterFile(argv[1])) return 3;
    lzb_vio::Parameter p;
    printf("track_mode=%s\n", p.track_mode_.c_str());
    printf("dataset_path=%s\n", p.dataset_path_.c_str());
    printf("fx=%.6f cx=%.6f cy=%.6f\n", p.fx1_, p.cx1_, p.cy1_);
    printf("P2_03=%.9f\n", p.projMatr2_[3]);
    printf("feature_match_error=%.3f num_features_tracking=%d inlier_rate=%.4f\n", p.feature_match_error_,
           p.num_features_tracking_, p.inlier_rate_);
    printf("iterationsCount=%d reprojectionError=%.3f confidence=%.3f\n", p.iterationsCount_,
           p.reprojectionError_, p.confidence_);
    printf("nFeatures=%d fScaleFactor=%.2f nLevels=%d fIniThFAST=%d fMinThFAST=%d\n", p.nFeatures_,
           p.fScaleFactor_, p.nLevels_, p.fIniThFAST_, p.fMinThFAST_);
    printf("missing=%d\n", lzb_vio::Config::Get<int>("no_such_key"));
    for (int i = 2; i < argc; i++) {
        cv::Mat m;
        bool ok = lzb_vio::ReadImageGray(argv[i], m);
        unsigned long long sum = 0;
        if (ok) for (int y = 0; y < m.rows; y++) for (int x = 0; x < m.cols; x++) sum = sum * 31 + m.ptr(y)[x];
        printf("image%d ok=%d rows=%d cols=%d hash=%llu\n", i - 1, ok ? 1 : 0, m.rows, m.cols, sum);
    }
    return 0;
}

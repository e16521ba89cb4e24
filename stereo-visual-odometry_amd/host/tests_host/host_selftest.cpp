// host_selftest -- CPU-only checks of the host mirror: YAML surface and image readers.
// usage: host_selftest config.yaml image1 [image2 ...]; prints key=value lines for pytest.
#include "lzb_vio/System.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
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

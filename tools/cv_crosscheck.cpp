// tools/cv_crosscheck.cpp -- the C++ side of the OpenCV cross-check (tests/test_cv_crosscheck_cpp.py): runs the REAL
// OpenCV callees of the reference's hot path on files written by the test and dumps their raw outputs, so that a box
// with OpenCV headers but no cv2 module can still pin the oracle.  Built only where `pkg-config --exists opencv4` (or
// `opencv`) succeeds -- tools/Makefile; there is no OpenCV in the build image, so this file has never been compiled here.
//
// usage: cv_crosscheck <dir>
//   in : <dir>/L0.pgm R0.pgm L1.pgm R1.pgm (binary PGM), pts.f32 (n x 2 float32), P.f64 (P1 | P2, 24 doubles)
//   out: <dir>/fast.f32 (n x 3: x y response), lk_<k>.f32 (n x 2) + lk_<k>.u8 (status) for the chain L0>R0, R0>R1, R1>L1, L1>L0
//        (each call's output is the next call's input, src/tracking.cpp:593-618), X4.f32 (4 x m), pnp.f64 (rvec tvec),
//        pnp_inliers.i32, version.txt
// Call-site arguments: /root/reference/src/tracking.cpp:101, 593-618, 292-294, 485.
#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/features2d.hpp>
#include <opencv2/imgcodecs.hpp>
#include <opencv2/video/tracking.hpp>

#include <cstdio>
#include <string>
#include <vector>

template <typename T>
static std::vector<T> slurp(const std::string &path)
{
    std::vector<T> v;
    if (FILE *f = std::fopen(path.c_str(), "rb")) {
        std::fseek(f, 0, SEEK_END);
        v.resize((size_t)std::ftell(f) / sizeof(T));
        std::fseek(f, 0, SEEK_SET);
        if (std::fread(v.data(), sizeof(T), v.size(), f) != v.size()) v.clear();
        std::fclose(f);
    }
    return v;
}
template <typename T>
static void dump(const std::string &path, const T *p, size_t n)
{
    if (FILE *f = std::fopen(path.c_str(), "wb")) { std::fwrite(p, sizeof(T), n, f); std::fclose(f); }
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: cv_crosscheck <dir>\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    cv::Mat img[4];
    const char *names[4] = {"L0", "R0", "L1", "R1"};
    for (int i = 0; i < 4; i++) {
        img[i] = cv::imread(d + names[i] + ".pgm", cv::IMREAD_GRAYSCALE);
        if (img[i].empty()) { std::fprintf(stderr, "cannot read %s\n", names[i]); return 1; }
    }
    // cv::FAST(img, kps, 20, true)
    std::vector<cv::KeyPoint> kps;
    cv::FAST(img[0], kps, 20, true);
    std::vector<float> fk;
    for (const auto &k : kps) { fk.push_back(k.pt.x); fk.push_back(k.pt.y); fk.push_back(k.response); }
    dump(d + "fast.f32", fk.data(), fk.size());
    // the circular LK chain, every output (failed points included) feeding the next call
    std::vector<float> pf = slurp<float>(d + "pts.f32");
    std::vector<cv::Point2f> cur(pf.size() / 2), nxt;
    for (size_t i = 0; i < cur.size(); i++) cur[i] = cv::Point2f(pf[2 * i], pf[2 * i + 1]);
    const int chain[4][2] = {{0, 1}, {1, 3}, {3, 2}, {2, 0}};       // L0>R0, R0>R1, R1>L1, L1>L0
    std::vector<std::vector<cv::Point2f>> outs;
    for (int c = 0; c < 4; c++) {
        std::vector<uchar> st;
        std::vector<float> err;
        cv::calcOpticalFlowPyrLK(img[chain[c][0]], img[chain[c][1]], cur, nxt, st, err, cv::Size(21, 21), 3,
                                 cv::TermCriteria(cv::TermCriteria::COUNT + cv::TermCriteria::EPS, 30, 0.01), 0, 0.001);
        dump(d + "lk_" + std::to_string(c) + ".f32", reinterpret_cast<const float *>(nxt.data()), 2 * nxt.size());
        dump(d + "lk_" + std::to_string(c) + ".u8", st.data(), st.size());
        outs.push_back(nxt);
        cur = nxt;
    }
    // triangulatePoints on the first two point sets the test hands over (x1.f32, x2.f32), then solvePnPRansac against x3.f32
    std::vector<double> P = slurp<double>(d + "P.f64");
    std::vector<float> a = slurp<float>(d + "x1.f32"), b = slurp<float>(d + "x2.f32"), c3 = slurp<float>(d + "x3.f32");
    if (P.size() == 24 && !a.empty() && a.size() == b.size()) {
        cv::Mat P1(3, 4, CV_64F, P.data()), P2(3, 4, CV_64F, P.data() + 12);
        const int m = (int)a.size() / 2;
        cv::Mat x1(m, 1, CV_32FC2, a.data()), x2(m, 1, CV_32FC2, b.data()), X4;
        cv::triangulatePoints(P1, P2, x1, x2, X4);
        cv::Mat X4c = X4.isContinuous() ? X4 : X4.clone();
        dump(d + "X4.f32", X4c.ptr<float>(), (size_t)4 * m);
        if (c3.size() == a.size()) {
            cv::Mat X3;
            cv::convertPointsFromHomogeneous(X4.t(), X3);
            cv::Mat K = P1(cv::Rect(0, 0, 3, 3)).clone(), rvec = cv::Mat::zeros(3, 1, CV_64F), t = cv::Mat::zeros(3, 1, CV_64F);
            cv::Mat x3(m, 1, CV_32FC2, c3.data());
            std::vector<int> inliers;
            cv::solvePnPRansac(X3, x3, K, cv::noArray(), rvec, t, true, 500, 0.5f, 0.99, inliers, cv::SOLVEPNP_ITERATIVE);
            double pose[6] = {rvec.at<double>(0), rvec.at<double>(1), rvec.at<double>(2), t.at<double>(0), t.at<double>(1), t.at<double>(2)};
            dump(d + "pnp.f64", pose, 6);
            dump(d + "pnp_inliers.i32", inliers.data(), inliers.size());
        }
    }
    if (FILE *f = std::fopen((d + "version.txt").c_str(), "w")) { std::fputs(CV_VERSION, f); std::fclose(f); }
    return 0;
}

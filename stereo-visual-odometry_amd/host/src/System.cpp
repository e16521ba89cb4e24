// System.cpp -- facade + KITTI dataset reader (reference src/System.cpp).  The reference's dead
// 0.5x resize (:93-97) is dropped; PNG decode replaces cv::imread.
#include "lzb_vio/System.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <zlib.h>

namespace lzb_vio {

// ---- image readers ---------------------------------------------------------------------------
static bool read_file(const std::string &path, std::vector<uint8_t> &buf)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    bool ok = n > 0 && fread(buf.data(), 1, (size_t)n, f) == (size_t)n;
    fclose(f);
    return ok;
}

// largest width / height accepted from a file header (svo_create's own limit): a corrupted or
// hostile header must be refused before anything is allocated from it
static const int kMaxImageDim = 16384;

static bool read_pgm(const std::vector<uint8_t> &b, cv::Mat &out)
{
    size_t p = 2;
    int vals[3], nv = 0;
    while (nv < 3 && p < b.size()) {
        while (p < b.size() && (b[p] == ' ' || b[p] == '\n' || b[p] == '\r' || b[p] == '\t')) p++;
        if (p < b.size() && b[p] == '#') { while (p < b.size() && b[p] != '\n') p++; continue; }
        int v = 0; bool any = false;
        while (p < b.size() && b[p] >= '0' && b[p] <= '9') {
            v = v * 10 + (b[p] - '0'); p++; any = true;
            if (v > kMaxImageDim) return false;             // also keeps the accumulator far from overflow
        }
        if (!any) return false;
        vals[nv++] = v;
    }
    p++;                                                    // single whitespace after maxval
    if (nv != 3 || vals[0] < 1 || vals[1] < 1 || vals[2] != 255 || b.size() < p + (size_t)vals[0] * vals[1]) return false;
    out.create(vals[1], vals[0]);
    for (int y = 0; y < vals[1]; y++) memcpy(out.ptr(y), &b[p + (size_t)y * vals[0]], vals[0]);
    return true;
}

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | (p[1] << 16) | (p[2] << 8) | p[3]; }

// 8-bit, non-interlaced PNG: gray (type 0), gray+alpha (4), RGB (2), RGBA (6)
static bool read_png(const std::vector<uint8_t> &b, cv::Mat &out)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 33 || memcmp(b.data(), sig, 8) != 0) return false;
    size_t p = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat;
    while (p + 12 <= b.size()) {
        uint32_t len = be32(&b[p]);
        const uint8_t *type = &b[p + 4];
        if (p + 12 + len > b.size()) return false;
        if (!memcmp(type, "IHDR", 4)) {
            w = (int)be32(&b[p + 8]); h = (int)be32(&b[p + 12]);
            depth = b[p + 16]; ctype = b[p + 17]; interlace = b[p + 20];
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), &b[p + 8], &b[p + 8 + len]);
        } else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + len;
    }
    if (w <= 0 || h <= 0 || w > kMaxImageDim || h > kMaxImageDim || depth != 8 || interlace != 0) return false;
    int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    // deflate expands by at most ~1032x: a header promising more pixels than the IDAT bytes can hold is corrupt
    if ((stride + 1) * (size_t)h > idat.size() * 1032 + 64) return false;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return false;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    out.create(h, w);
    for (int y = 0; y < h; y++) {
        const uint8_t *src = &raw[(stride + 1) * (size_t)y];
        const int ft = src[0];
        for (size_t i = 0; i < stride; i++) {
            int a = i >= (size_t)ch ? cur[i - ch] : 0, bb = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0, x = src[1 + i];
            int pr;
            switch (ft) {
            case 0: pr = 0; break;
            case 1: pr = a; break;
            case 2: pr = bb; break;
            case 3: pr = (a + bb) >> 1; break;
            case 4: { int pp = a + bb - c, pa = abs(pp - a), pb = abs(pp - bb), pc = abs(pp - c);
                      pr = (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
            default: return false;
            }
            cur[i] = (uint8_t)(x + pr);
        }
        uint8_t *d = out.ptr(y);
        if (ch <= 2) for (int x = 0; x < w; x++) d[x] = cur[(size_t)x * ch];
        else for (int x = 0; x < w; x++) {
            // cv::imread(IMREAD_GRAYSCALE) colour conversion: (R*4899 + G*9617 + B*1868 + 8192) >> 14
            const uint8_t *q = &cur[(size_t)x * ch];
            d[x] = (uint8_t)((q[0] * 4899 + q[1] * 9617 + q[2] * 1868 + 8192) >> 14);
        }
        prev.swap(cur);
    }
    return true;
}

bool ReadImageGray(const std::string &path, cv::Mat &out)
{
    std::vector<uint8_t> b;
    if (!read_file(path, b) || b.size() < 8) return false;
    if (b[0] == 'P' && b[1] == '5') return read_pgm(b, out);
    return read_png(b, out);
}

// ---- System ------------------------------------------------------------------------------------
System::System(std::string &config_path) : config_file_path_(config_path)
{
    if (Config::SetParameterFile(config_file_path_) == false) {
        fprintf(stderr, "unable to open %s\n", config_file_path_.c_str());
        exit(-1);                                          // as the reference does (src/System.cpp:15-19)
    }
    init_parameter_ = Parameter::Ptr(new Parameter);
    sensors_ = Sensors::Ptr(new Sensors(init_parameter_));
    tracking_ = Tracking::Ptr(new Tracking(this, init_parameter_, sensors_));
    dataset_path_ = init_parameter_->dataset_path_;
    if (Config::Has("pose_file")) SetPoseFile(Config::Get<std::string>("pose_file"));   // additive key
    // additive keys: the headless stand-in for Tracking::displayTracking (src/tracking.cpp:345-382) and
    // the reference's per-frame Feature carriers
    if (Config::Has("tracks_file")) SetTracksFile(Config::Get<std::string>("tracks_file"));
    if (Config::Has("fill_features") && Config::Get<int>("fill_features") != 0) tracking_->SetFillFeatures(true);
}

System::~System()
{
    if (pose_file_) fclose(pose_file_);
    if (tracks_file_) fclose(tracks_file_);
}

bool System::SetTracksFile(const std::string &path)
{
    if (tracks_file_) fclose(tracks_file_);
    tracks_file_ = fopen(path.c_str(), "w");
    if (tracks_file_) fprintf(tracks_file_, "# F frame ok fail_stage n_cur_kps n_tracked n_inliers / T x1l y1l x1r y1r x2l y2l inlier\n");
    return tracks_file_ != nullptr;
}

// one record per frame + one row per matched track: the data displayTracking draws as lines/circles
void System::WriteTracks()
{
    if (!tracks_file_) return;
    const svo_step_result &r = tracking_->LastResult();
    fprintf(tracks_file_, "F %d %d %d %d %d %d\n", current_image_index_ - 1, r.ok, r.fail_stage, r.n_cur_kps, r.n_tracked, r.n_inliers);
    std::vector<cv::Point2f> a, b, c;
    std::vector<unsigned char> in;
    if (!tracking_->GetLastTracks(a, b, c, in)) return;
    for (size_t i = 0; i < a.size(); i++)
        fprintf(tracks_file_, "T %.4f %.4f %.4f %.4f %.4f %.4f %d\n", a[i].x, a[i].y, b[i].x, b[i].y, c[i].x, c[i].y, (int)in[i]);
}

bool System::SetPoseFile(const std::string &path)
{
    if (pose_file_) fclose(pose_file_);
    pose_file_ = fopen(path.c_str(), "w");
    return pose_file_ != nullptr;
}

void System::WritePose()
{
    Pose4x4 P = tracking_->GetPose();
    WritePoseRow(P.m);
}

void System::WritePoseRow(const double *pose16)
{
    if (!pose_file_) return;
    for (int i = 0; i < 12; i++) fprintf(pose_file_, "%.9e%c", pose16[i], i == 11 ? '\n' : ' ');
}

void System::Run()
{
    const int batch = Config::Has("batch_size") ? Config::Get<int>("batch_size") : 1;
    if (batch > 1) {
        RunBatched(batch, Config::Has("decode_threads") ? Config::Get<int>("decode_threads") : 8);
        Shutdown();
        return;
    }
    while (1) {
        if (Step() == false) break;
    }
    Shutdown();
}

bool System::Step()
{
    Frame::Ptr new_frame = NextFrame_kitti();
    if (new_frame == nullptr) return false;
    auto t1 = std::chrono::steady_clock::now();
    bool success = tracking_->AddFrame(new_frame);
    auto t2 = std::chrono::steady_clock::now();
    double dt = std::chrono::duration_cast<std::chrono::duration<double>>(t2 - t1).count();
    if (getenv("LZB_VIO_VERBOSE")) LZB_LOG("INFO", "VO cost time: %f seconds (%s)", dt, success ? "ok" : "skipped");
    WritePose();
    WriteTracks();
    return true;                                           // the reference ignores AddFrame's result here (:53,58)
}

bool System::Step_ros(Frame::Ptr new_frame)
{
    if (new_frame == nullptr) return false;
    bool success = tracking_->AddFrame(new_frame);
    WritePose();
    WriteTracks();
    return success;
}

// <dataset_path>/image_0/%06d.png + image_1/%06d.png (src/System.cpp:77-85); .pgm accepted too
bool System::ReadStereo(int index, cv::Mat &left, cv::Mat &right)
{
    char name[32];
    const char *ext[2] = {"png", "pgm"};
    for (int cam = 0; cam < 2; cam++) {
        bool ok = false;
        for (int e = 0; e < 2 && !ok; e++) {
            snprintf(name, sizeof(name), "/image_%d/%06d.%s", cam, index, ext[e]);
            ok = ReadImageGray(dataset_path_ + name, cam == 0 ? left : right);
        }
        if (!ok) return false;
    }
    return true;
}

Frame::Ptr System::NextFrame_kitti()
{
    cv::Mat image_left, image_right;
    if (!ReadStereo(current_image_index_, image_left, image_right)) {
        LZB_LOG("WARNING", "cannot find images at index %d", current_image_index_);
        return nullptr;
    }
    auto new_frame = Frame::CreateFrame();
    new_frame->left_img_ = image_left;
    new_frame->right_img_ = image_right;
    current_image_index_++;
    return new_frame;
}

// ---- batched runner ------------------------------------------------------------------------------
// Chunk c holds frames [c*B, c*B + B]: B pairs plus a one-frame halo (the last frame of a chunk is
// the first of the next, copied rather than decoded again).  While the GPU tracks chunk c the
// decoder threads fill the other page-locked buffer with chunk c+1.
void System::RunBatched(int B, int decode_threads)
{
    cv::Mat l0, r0;
    if (!ReadStereo(0, l0, r0)) { LZB_LOG("WARNING", "cannot find images at index %d", 0); return; }
    const int w = l0.cols, h = l0.rows;
    if (r0.cols != w || r0.rows != h) { LZB_LOG("ERROR", "left/right size mismatch at index %d", 0); return; }
    if (!tracking_->EnsureBatchContext(w, h, B)) return;
    svo_ctx *ctx = tracking_->Context();
    const int pitch = (w + 255) / 256 * 256;                // the library's staging pitch: one copy per camera
    const size_t fbytes = (size_t)pitch * h;
    uint8_t *pin[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    for (int k = 0; k < 2; k++)
        for (int cam = 0; cam < 2; cam++)
            if (svo_host_alloc(ctx, fbytes * (size_t)(B + 1), (void **)&pin[k][cam]) != SVO_OK) {
                LZB_LOG("ERROR", "svo_host_alloc: %s", svo_last_error(ctx));
                return;
            }
    const int T = decode_threads < 1 ? 1 : decode_threads;
    // decodes frames first .. first+count-1 into slots slot0.. of buffer k; returns how many
    // consecutive frames (from `first`) were read
    auto decode = [&](int k, int slot0, int first, int count) -> int {
        std::vector<char> ok((size_t)count, 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < T && t < count; t++)
            pool.emplace_back([&, t]() {
                cv::Mat l, r;
                for (int i = t; i < count; i += T) {
                    if (!ReadStereo(first + i, l, r) || l.cols != w || l.rows != h || r.cols != w || r.rows != h) continue;
                    for (int y = 0; y < h; y++) {
                        memcpy(pin[k][0] + (size_t)(slot0 + i) * fbytes + (size_t)y * pitch, l.ptr(y), (size_t)w);
                        memcpy(pin[k][1] + (size_t)(slot0 + i) * fbytes + (size_t)y * pitch, r.ptr(y), (size_t)w);
                    }
                    ok[(size_t)i] = 1;
                }
            });
        for (auto &th : pool) th.join();
        int n = 0;
        while (n < count && ok[(size_t)n]) n++;
        return n;
    };
    auto upload = [&](int k, int n) {
        int rc = svo_upload_frames(ctx, k, pin[k][0], pin[k][1], pitch, (int64_t)fbytes, n);
        if (rc != SVO_OK) LZB_LOG("ERROR", "svo_upload_frames: %s", svo_last_error(ctx));
        return rc == SVO_OK;
    };

    int k = 0, next = 0;
    int cur_n = decode(0, 0, 0, B + 1);
    next = cur_n;
    current_image_index_ = cur_n > 0 ? 1 : 0;
    if (cur_n > 0) WritePose();                             // frame 0: StereoInit_f2f, pose = identity
    std::vector<svo_step_result> recs;
    auto flush = [&](double seconds, int pairs) {           // the oldest outstanding chunk's records -> pose file
        recs.clear();
        if (!tracking_->CollectUploaded(recs)) return false;
        if (getenv("LZB_VIO_VERBOSE")) LZB_LOG("INFO", "VO cost time: %f seconds for %d pairs", seconds, pairs);
        for (const auto &r : recs) { WritePoseRow(r.pose); current_image_index_++; }
        return true;
    };
    bool ok = cur_n >= 2 && upload(0, cur_n);
    int outstanding = 0, chunk = 0;
    auto t_prev = std::chrono::steady_clock::now();
    while (ok) {
        int nn = 0;
        std::thread bg;
        if (cur_n == B + 1) {                               // a full chunk: there may be more frames
            // the other page-locked buffer is the source of the upload issued one iteration ago
            if (chunk > 0 && svo_wait_upload(ctx, k ^ 1) != SVO_OK) { ok = false; break; }
            bg = std::thread([&]() {
                for (int cam = 0; cam < 2; cam++)
                    memcpy(pin[k ^ 1][cam], pin[k][cam] + (size_t)B * fbytes, fbytes);
                nn = 1 + decode(k ^ 1, 1, next, B);
            });
        }
        // returns at once: the GPU works on chunk c (its pose chain continues from chunk c-1's last pose
        // on the device) while chunk c-1's records are fetched and chunk c+1 is decoded
        ok = tracking_->TrackUploadedAsync(k, cur_n, chunk > 0);
        if (ok) outstanding++;
        auto t_now = std::chrono::steady_clock::now();
        if (ok && outstanding == 2) { ok = flush(std::chrono::duration<double>(t_now - t_prev).count(), B); outstanding--; }
        t_prev = t_now;
        if (bg.joinable()) bg.join();
        // chunk c+1 crosses PCIe on the copy stream beside chunk c's kernels
        const bool more = ok && nn >= 2 && upload(k ^ 1, nn);
        if (!ok || !more) break;
        next += nn - 1;
        k ^= 1;
        cur_n = nn;
        chunk++;
    }
    while (ok && outstanding > 0) { ok = flush(0.0, 0); outstanding--; }
    svo_sync(ctx);
    for (int q = 0; q < 2; q++)
        for (int cam = 0; cam < 2; cam++) svo_host_free(ctx, pin[q][cam]);
}

void System::Shutdown() {}
void System::Reset() {}

}  // namespace lzb_vio

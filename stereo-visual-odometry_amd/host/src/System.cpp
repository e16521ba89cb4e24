// System.cpp -- facade + KITTI dataset reader (reference src/System.cpp).  The reference's dead
// 0.5x resize (:93-97) is dropped; PNG decode replaces cv::imread.
#include "lzb_vio/System.h"
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <algorithm>
#include <sched.h>
#include <sys/stat.h>
#include <thread>
#include <zlib.h>
#include <memory>
#include "fast_inflate.h"

namespace lzb_vio {

// seconds since the process started (/proc/self/stat's start time is in clock ticks: the static initialiser below runs
// when liblzb_vio.so is loaded, a few milliseconds after exec -- close enough for a phase log)
static const std::chrono::steady_clock::time_point g_process_t0 = std::chrono::steady_clock::now();
double lzb_seconds_since_start() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - g_process_t0).count(); }

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

static inline int paeth_pred(int a, int b, int c)
{
    int pa = b - c, pb = a - c, pc = pa + pb;                // p - a, p - b, p - c for p = a + b - c
    pa = abs(pa); pb = abs(pb); pc = abs(pc);
    const int t = pb <= pc ? b : c;
    return (pa <= pb && pa <= pc) ? a : t;
}

// Two consecutive Paeth rows of a one-byte-per-pixel image at once.  A Paeth pixel waits for its left
// neighbour, so a row is one dependent chain of ~8 cycles per pixel (it was half of the decode time of a
// KITTI-size frame: PNG encoders pick Paeth for most rows of camera images); the row below needs only
// pixels of the upper row that are at least one column behind, so the two chains run skewed by one pixel and
// the CPU overlaps them.  A: filtered bytes of the upper row (P = the reconstructed row above it), B: of the
// lower row; both reconstructed in place.
static void png_unfilter_paeth2(uint8_t *A, uint8_t *B, const uint8_t *P, size_t n)
{
    if (n == 0) return;
    A[0] = (uint8_t)(A[0] + P[0]);
    B[0] = (uint8_t)(B[0] + A[0]);
    if (n == 1) return;
    A[1] = (uint8_t)(A[1] + paeth_pred(A[0], P[1], P[0]));
    int aA = A[1], aB = B[0];
    for (size_t i = 1; i + 1 < n; i++) {
        const int xa = A[i + 1] + paeth_pred(aA, P[i + 1], P[i]);            // A[i + 1]
        const int xb = B[i] + paeth_pred(aB, aA, A[i - 1]);                  // B[i]: its upper neighbours A[i], A[i - 1] are done
        aA = xa & 0xFF; aB = xb & 0xFF;
        A[i + 1] = (uint8_t)aA; B[i] = (uint8_t)aB;
    }
    B[n - 1] = (uint8_t)(B[n - 1] + paeth_pred(aB, A[n - 1], A[n - 2]));
}

// FOUR consecutive Paeth rows at once, skewed by one pixel each (step t reconstructs R0[t], R1[t - 1], R2[t - 2], R3[t - 3]): four
// independent dependency chains in flight instead of two -- the one-shot reader below has the whole filtered image in memory,
// so it can look three rows ahead.  P = the reconstructed row above R0.
static void png_unfilter_paeth4(uint8_t *R0, uint8_t *R1, uint8_t *R2, uint8_t *R3, const uint8_t *P, size_t n)
{
    uint8_t *rows[4] = {R0, R1, R2, R3};
    auto px = [&](int r, size_t j) {
        const uint8_t *up = r ? rows[r - 1] : P;
        const int a = j ? rows[r][j - 1] : 0, b = up[j], c = j ? up[j - 1] : 0;
        rows[r][j] = (uint8_t)(rows[r][j] + paeth_pred(a, b, c));
    };
    if (n < 8) {
        for (int r = 0; r < 4; r++) for (size_t j = 0; j < n; j++) px(r, j);
        return;
    }
    for (size_t t = 0; t < 4; t++)
        for (int r = 0; r <= (int)t && r < 4; r++) px(r, t - (size_t)r);
    int l0 = R0[3], p0 = R0[2], l1 = R1[2], p1 = R1[1], l2 = R2[1], p2 = R2[0], l3 = R3[0];
    for (size_t t = 4; t < n; t++) {
        const int n0 = R0[t] + paeth_pred(l0, P[t], P[t - 1]);
        const int n1 = R1[t - 1] + paeth_pred(l1, l0, p0);
        const int n2 = R2[t - 2] + paeth_pred(l2, l1, p1);
        const int n3 = R3[t - 3] + paeth_pred(l3, l2, p2);
        p0 = l0; p1 = l1; p2 = l2;
        l0 = n0 & 0xFF; l1 = n1 & 0xFF; l2 = n2 & 0xFF; l3 = n3 & 0xFF;
        R0[t] = (uint8_t)l0; R1[t - 1] = (uint8_t)l1; R2[t - 2] = (uint8_t)l2; R3[t - 3] = (uint8_t)l3;
    }
    for (size_t t = n; t < n + 3; t++)
        for (int r = (int)(t - n) + 1; r < 4; r++) px(r, t - (size_t)r);
}

// PNG row filters (RFC 2083 section 6) undone in place: row = filtered bytes in, reconstructed bytes out;
// prev = the reconstructed row above (all zero for the first row); bpp = bytes per pixel.  One loop per
// filter type (the per-byte switch of the first version cost more than the inflate it followed).
static bool png_unfilter_row(int ft, uint8_t *row, const uint8_t *prev, size_t n, size_t bpp)
{
    switch (ft) {
    case 0: return true;
    case 1: for (size_t i = bpp; i < n; i++) row[i] = (uint8_t)(row[i] + row[i - bpp]); return true;
    case 2: for (size_t i = 0; i < n; i++) row[i] = (uint8_t)(row[i] + prev[i]); return true;
    case 3:
        for (size_t i = 0; i < bpp && i < n; i++) row[i] = (uint8_t)(row[i] + (prev[i] >> 1));
        for (size_t i = bpp; i < n; i++) row[i] = (uint8_t)(row[i] + ((row[i - bpp] + prev[i]) >> 1));
        return true;
    case 4:
        for (size_t i = 0; i < bpp && i < n; i++) row[i] = (uint8_t)(row[i] + prev[i]);     // a = c = 0: the predictor is b
        for (size_t i = bpp; i < n; i++) row[i] = (uint8_t)(row[i] + paeth_pred(row[i - bpp], prev[i], prev[i - bpp]));
        return true;
    default: return false;
    }
}

// 8-bit, non-interlaced PNG: gray (type 0), gray+alpha (4), RGB (2), RGBA (6).  The IDAT chunks are
// inflated as ONE zlib stream fed chunk by chunk (no concatenated copy), a band of rows at a time (the
// filtered bytes stay in cache for the unfilter pass), and a gray image is reconstructed straight into
// its destination rows -- `dst(w, h, &pitch)` is asked for them once the header is known, so the batched
// runner's decoder threads write into page-locked memory with no cv::Mat in between.
template <typename GetDst>
static bool png_decode(const std::vector<uint8_t> &b, GetDst dst_for)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (b.size() < 33 || memcmp(b.data(), sig, 8) != 0) return false;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<std::pair<size_t, size_t>> idat;            // (offset, length) of every IDAT payload, in file order
    size_t idat_bytes = 0;
    for (size_t p = 8; p + 12 <= b.size();) {
        const uint32_t len = be32(&b[p]);
        const uint8_t *type = &b[p + 4];
        if (p + 12 + (size_t)len > b.size()) return false;
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) return false;
            w = (int)be32(&b[p + 8]); h = (int)be32(&b[p + 12]);
            depth = b[p + 16]; ctype = b[p + 17]; interlace = b[p + 20];
        } else if (!memcmp(type, "IDAT", 4)) {
            if (len) { idat.emplace_back(p + 8, (size_t)len); idat_bytes += len; }
        } else if (!memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    if (w <= 0 || h <= 0 || w > kMaxImageDim || h > kMaxImageDim || depth != 8 || interlace != 0 || idat.empty()) return false;
    const int ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
    if (!ch) return false;
    const size_t stride = (size_t)w * ch;
    // deflate expands by at most ~1032x: a header promising more pixels than the IDAT bytes can hold is corrupt
    // (refused before anything is allocated from it)
    if ((stride + 1) * (size_t)h > idat_bytes * 1032 + 64) return false;
    int pitch = 0;
    uint8_t *out = dst_for(w, h, &pitch);
    if (!out || pitch < w) return false;

    // ---- one shot (fast_inflate.h): the whole stream into a per-thread buffer, then the filters with up to four Paeth rows in
    // flight.  It says yes only to a stream that decoded to exactly the image, ended with the input and matches its Adler-32;
    // whatever it does not accept goes through zlib below, which stays the judge of what a corrupt file is.
    {
        static thread_local std::vector<uint8_t> zbuf, rawbuf;
        static thread_local std::unique_ptr<finf::Tables> tabs;
        const size_t raw_bytes = (stride + 1) * (size_t)h;
        if (!tabs) tabs.reset(new finf::Tables);
        if (zbuf.size() < idat_bytes + 16) zbuf.resize(idat_bytes + 16);
        if (rawbuf.size() < raw_bytes + finf::kSlack) rawbuf.resize(raw_bytes + finf::kSlack);
        size_t zp = 0;
        for (const auto &c : idat) { memcpy(zbuf.data() + zp, &b[c.first], c.second); zp += c.second; }
        memset(zbuf.data() + zp, 0, 16);
        if (!getenv("LZB_VIO_PNG_ZLIB") && finf::inflate_zlib(zbuf.data(), idat_bytes, rawbuf.data(), raw_bytes, *tabs)) {
            std::vector<uint8_t> zero(stride, 0);
            const uint8_t *prev = zero.data();
            uint8_t *raw = rawbuf.data();
            for (int y = 0; y < h; y++) {
                uint8_t *src = raw + (stride + 1) * (size_t)y;
                uint8_t *d = out + (size_t)y * pitch;
                const int ft = src[0];
                if (ch == 1) {
                    memcpy(d, src + 1, stride);
                    int run = 1;                              // consecutive Paeth rows from here (at most four)
                    if (ft == 4) while (run < 4 && y + run < h && src[(stride + 1) * (size_t)run] == 4) run++;
                    if (ft == 4 && run == 4) {
                        for (int r = 1; r < 4; r++) memcpy(d + (size_t)r * pitch, src + (stride + 1) * (size_t)r + 1, stride);
                        png_unfilter_paeth4(d, d + pitch, d + 2 * (size_t)pitch, d + 3 * (size_t)pitch, prev, stride);
                        y += 3;
                        prev = d + 3 * (size_t)pitch;
                    } else if (ft == 4 && run >= 2) {
                        memcpy(d + pitch, src + stride + 2, stride);
                        png_unfilter_paeth2(d, d + pitch, prev, stride);
                        y += 1;
                        prev = d + pitch;
                    } else {
                        if (!png_unfilter_row(ft, d, prev, stride, 1)) return false;
                        prev = d;
                    }
                } else {
                    if (!png_unfilter_row(ft, src + 1, prev, stride, (size_t)ch)) return false;
                    if (ch == 2) for (int x = 0; x < w; x++) d[x] = src[1 + (size_t)x * 2];
                    else for (int x = 0; x < w; x++) {
                        const uint8_t *q = src + 1 + (size_t)x * ch;
                        d[x] = (uint8_t)((q[0] * 4899 + q[1] * 9617 + q[2] * 1868 + 8192) >> 14);
                    }
                    prev = src + 1;                          // the whole filtered image is in memory: the row above stays where it is
                }
            }
            return true;
        }
    }
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit(&zs) != Z_OK) return false;
    size_t chunk = 0;
    zs.next_in = const_cast<Bytef *>(&b[idat[0].first]); zs.avail_in = (uInt)idat[0].second;
    const int band = 32;                                    // rows inflated per call
    std::vector<uint8_t> raw((stride + 1) * (size_t)band), keep(ch > 1 ? stride : 0), zero(stride, 0);
    const uint8_t *prev = zero.data();
    bool ok = true;
    for (int y0 = 0; ok && y0 < h; y0 += band) {
        const int rows = h - y0 < band ? h - y0 : band;
        zs.next_out = raw.data(); zs.avail_out = (uInt)((stride + 1) * (size_t)rows);
        while (ok && zs.avail_out > 0) {
            if (zs.avail_in == 0) {
                if (++chunk >= idat.size()) { ok = false; break; }
                zs.next_in = const_cast<Bytef *>(&b[idat[chunk].first]); zs.avail_in = (uInt)idat[chunk].second;
            }
            const int rc = inflate(&zs, Z_NO_FLUSH);
            if (rc == Z_STREAM_END) { ok = zs.avail_out == 0; break; }
            if (rc != Z_OK && !(rc == Z_BUF_ERROR && zs.avail_in == 0)) ok = false;
        }
        for (int r = 0; ok && r < rows; r++) {
            uint8_t *src = &raw[(stride + 1) * (size_t)r];
            uint8_t *d = out + (size_t)(y0 + r) * pitch;
            if (ch == 1) {                                  // reconstructed in the destination row itself
                memcpy(d, src + 1, stride);
                if (src[0] == 4 && r + 1 < rows && src[stride + 1] == 4) {      // two Paeth rows: one skewed pass
                    uint8_t *d2 = d + pitch;
                    memcpy(d2, src + stride + 2, stride);
                    png_unfilter_paeth2(d, d2, prev, stride);
                    prev = d2;
                    r++;
                } else {
                    ok = png_unfilter_row(src[0], d, prev, stride, 1);
                    prev = d;
                }
            } else {
                ok = png_unfilter_row(src[0], src + 1, prev, stride, (size_t)ch);
                if (ch == 2) for (int x = 0; x < w; x++) d[x] = src[1 + (size_t)x * 2];
                else for (int x = 0; x < w; x++) {
                    // cv::imread(IMREAD_GRAYSCALE) colour conversion: (R*4899 + G*9617 + B*1868 + 8192) >> 14
                    const uint8_t *q = src + 1 + (size_t)x * ch;
                    d[x] = (uint8_t)((q[0] * 4899 + q[1] * 9617 + q[2] * 1868 + 8192) >> 14);
                }
                // the band buffer is reused: the last reconstructed row of a band is kept for the next one
                if (r == rows - 1) { memcpy(keep.data(), src + 1, stride); prev = keep.data(); }
                else prev = src + 1;
            }
        }
    }
    // The rows are complete, the STREAM must be too: one more inflate with a one-byte output buffer has to reach
    // Z_STREAM_END without producing anything -- that is where zlib compares the Adler-32 trailer (a damaged literal
    // inside valid deflate syntax yields Z_DATA_ERROR here) and where extra rows or trailing bytes show.
    for (bool ended = false; ok && !ended;) {
        uint8_t extra = 0;
        zs.next_out = &extra; zs.avail_out = 1;
        if (zs.avail_in == 0 && chunk + 1 < idat.size()) {
            chunk++;
            zs.next_in = const_cast<Bytef *>(&b[idat[chunk].first]); zs.avail_in = (uInt)idat[chunk].second;
        }
        const int rc = inflate(&zs, Z_FINISH);
        if (zs.avail_out == 0) ok = false;                  // more pixels than the header announced
        else if (rc == Z_STREAM_END) ended = true;
        else if (rc == Z_BUF_ERROR || rc == Z_OK) ok = zs.avail_in == 0 && chunk + 1 < idat.size();   // needs the next IDAT chunk
        else ok = false;                                    // Z_DATA_ERROR: checksum mismatch / corrupt stream
    }
    inflateEnd(&zs);
    return ok;
}

static bool read_png(const std::vector<uint8_t> &b, cv::Mat &out)
{
    return png_decode(b, [&](int w, int h, int *pitch) { out.create(h, w); *pitch = (int)out.step; return out.ptr(0); });
}

bool ReadImageGray(const std::string &path, cv::Mat &out)
{
    std::vector<uint8_t> b;
    if (!read_file(path, b) || b.size() < 8) return false;
    if (b[0] == 'P' && b[1] == '5') return read_pgm(b, out);
    return read_png(b, out);
}

// The same decode straight into caller-owned rows (`pitch` bytes apart) of a w x h image; a file of
// another size is refused.  Used by the batched runner's decoder threads on page-locked memory.
bool ReadImageGrayInto(const std::string &path, uint8_t *dst, int pitch, int w, int h)
{
    std::vector<uint8_t> b;
    if (!dst || !read_file(path, b) || b.size() < 8) return false;
    if (b[0] == 'P' && b[1] == '5') {
        cv::Mat m;
        if (!read_pgm(b, m) || m.cols != w || m.rows != h) return false;
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * pitch, m.ptr(y), (size_t)w);
        return true;
    }
    return png_decode(b, [&](int fw, int fh, int *p) -> uint8_t * { *p = pitch; return (fw == w && fh == h) ? dst : nullptr; });
}

// ---- System ------------------------------------------------------------------------------------
bool System::s_defer_yaml_outputs = false;
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
    // additive keys pose_file / tracks_file (the headless stand-in for Tracking::displayTracking, src/tracking.cpp:345-382).
    // A System that only probes the sequence or hands its records to a sink (RunSplitPairs) must not open -- and truncate --
    // the files the YAML names: it remembers the names instead.
    if (Config::Has("pose_file")) yaml_pose_file_ = Config::Get<std::string>("pose_file");
    if (Config::Has("tracks_file")) yaml_tracks_file_ = Config::Get<std::string>("tracks_file");
    if (!s_defer_yaml_outputs) {
        if (!yaml_pose_file_.empty()) SetPoseFile(yaml_pose_file_);
        if (!yaml_tracks_file_.empty()) SetTracksFile(yaml_tracks_file_);
    }
    if (Config::Has("fill_features") && Config::Get<int>("fill_features") != 0) tracking_->SetFillFeatures(true);
    batch_size_ = Config::Has("batch_size") ? Config::Get<int>("batch_size") : 1;
    decode_threads_ = Config::Has("decode_threads") ? Config::Get<int>("decode_threads") : 0;    // 0: the usable cores
    stream_depth_ = Config::Has("stream_depth") ? Config::Get<int>("stream_depth") : 0;           // 0: Step_ros is synchronous
    if (stream_depth_ > 0 && tracks_file_)
        LZB_LOG("WARNING", "tracks_file is set together with stream_depth: the pipelined stream keeps poses only, the tracks file "
                           "will hold its header and nothing else (use the per-frame loop or batch_size for track dumps)");
}

System::~System()
{
    StreamRelease();
    for (int q = 0; q < 2; q++)
        for (int cam = 0; cam < 2; cam++) if (batch_pin_[q][cam]) svo_host_free(nullptr, batch_pin_[q][cam]);
    if (pose_file_) fclose(pose_file_);
    if (tracks_file_) fclose(tracks_file_);
}

// cores this process may run on (the affinity mask, not the machine's core count), at most 64
static int usable_cores()
{
    cpu_set_t set;
    int n = 0;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : n > 64 ? 64 : n;
}

// consecutive stereo frames present under dataset_path_ (both cameras, .png or .pgm), from index 0
int System::CountFrames() const
{
    auto have = [&](int index) {
        char name[32];
        struct stat st;
        for (int cam = 0; cam < 2; cam++) {
            bool ok = false;
            for (const char *ext : {"png", "pgm"}) {
                snprintf(name, sizeof(name), "/image_%d/%06d.%s", cam, index, ext);
                if (stat((dataset_path_ + name).c_str(), &st) == 0) { ok = true; break; }
            }
            if (!ok) return false;
        }
        return true;
    };
    int n = 0;
    while (have(n)) n++;
    return n;
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
    if (batch_size_ > 1) {
        RunBatched(batch_size_, decode_threads_);
        Shutdown();
        return;
    }
    if (stream_depth_ > 0) {                               // the per-frame loop through the pipelined stream
        for (;;) {
            Frame::Ptr f = NextFrame_kitti();
            if (f == nullptr || !Step_ros(f)) break;
        }
        // whatever was launched is collected even when a later push or submission failed (its poses are complete on the
        // GPU); the failure is kept for the caller's exit code
        std::vector<Pose4x4> poses;
        if (!StreamFlush()) run_failed_ = true;
        StreamPoll(poses, true);
        for (const auto &P : poses) WritePoseRow(P.m);
        if (stream_.failed) run_failed_ = true;
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
    if (stream_depth_ > 0) {
        // pipelined: the frame joins the current micro-batch; poses that have completed meanwhile go to the pose file.
        // The return value says the frame was ACCEPTED (its own pose follows stream_depth frames later).
        if (!StreamPush(new_frame)) return false;
        std::vector<Pose4x4> poses;
        StreamPoll(poses, false);
        for (const auto &P : poses) WritePoseRow(P.m);
        return true;
    }
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
    if (!ReadStereo(frame_base_, l0, r0)) { LZB_LOG("WARNING", "cannot find images at index %d", frame_base_); return; }
    const int w = l0.cols, h = l0.rows;
    if (r0.cols != w || r0.rows != h) { LZB_LOG("ERROR", "left/right size mismatch at index %d", 0); return; }
    LZB_PHASE("first frame read (size known)");
    const int pitch = (w + 255) / 256 * 256;                // the library's staging pitch: one copy per camera
    const size_t fbytes = (size_t)pitch * h;
    uint8_t *(&pin)[2][2] = batch_pin_;
    const size_t pin_bytes = fbytes * (size_t)(B + 1);
    if (batch_pin_bytes_ < pin_bytes) {                     // (a second run with larger chunks: start over)
        for (int q = 0; q < 2; q++)
            for (int cam = 0; cam < 2; cam++) if (pin[q][cam]) { svo_host_free(nullptr, pin[q][cam]); pin[q][cam] = nullptr; }
        batch_pin_bytes_ = 0;
    }
    const bool have_pins = batch_pin_bytes_ >= pin_bytes;
    // Start-up, overlapped: page-locking the four chunk buffers (0.25 ms per MB) runs on a thread of its own BESIDE the
    // context creation (HIP runtime start + one device allocation), and chunk 0 is decoded as soon as ITS two buffers
    // are there -- on a 1000-frame run these three were, one after the other, half of the wall time.
    std::atomic<int> pinned(0);                             // buffers page-locked so far (2 = chunk 0's pair), -1 = failed
    std::thread pin_thread([&]() {
        for (int k = 0; k < 2; k++)
            for (int cam = 0; cam < 2; cam++) {
                if (!have_pins && svo_host_alloc(nullptr, pin_bytes, (void **)&pin[k][cam]) != SVO_OK) { pinned.store(-1); return; }
                pinned.fetch_add(1);
            }
    });
    // decode_threads < 1: every core this process may use.  A work item is ONE image (left and right of a
    // frame are inflated side by side), handed out through a counter, decoded straight into the page-locked chunk.
    const int T = decode_threads < 1 ? usable_cores() : decode_threads;
    // decodes frames first .. first+count-1 into slots slot0.. of buffer k; returns how many
    // consecutive frames (from `first`) were read
    auto decode = [&](int k, int slot0, int first, int count) -> int {
        first += frame_base_;                               // SetFrameRange: this System's frame 0 is the sequence's frame_base_
        if (frame_last_ >= 0 && first + count - 1 > frame_last_) count = frame_last_ - first + 1;
        if (count <= 0) return 0;
        std::vector<char> ok((size_t)count * 2, 0);
        std::atomic<int> next_item(0);
        auto work = [&]() {
            char name[32];
            for (;;) {
                const int it = next_item.fetch_add(1);
                if (it >= 2 * count) break;
                const int i = it >> 1, cam = it & 1;
                uint8_t *dst = pin[k][cam] + (size_t)(slot0 + i) * fbytes;
                for (const char *ext : {"png", "pgm"}) {
                    snprintf(name, sizeof(name), "/image_%d/%06d.%s", cam, first + i, ext);
                    if (ReadImageGrayInto(dataset_path_ + name, dst, pitch, w, h)) { ok[(size_t)it] = 1; break; }
                }
            }
        };
        std::vector<std::thread> pool;
        const int nt = T < 2 * count ? T : 2 * count;
        for (int t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (auto &th : pool) th.join();
        int n = 0;
        while (n < count && ok[(size_t)2 * n] && ok[(size_t)2 * n + 1]) n++;
        return n;
    };
    auto upload = [&](int k, int n) {
        svo_ctx *c = tracking_->Context();
        int rc = svo_upload_frames(c, k, pin[k][0], pin[k][1], pitch, (int64_t)fbytes, n);
        if (rc != SVO_OK) LZB_LOG("ERROR", "svo_upload_frames: %s", svo_last_error(c));
        return rc == SVO_OK;
    };

    int cur_n = 0;
    double decode0_seconds = 0;
    std::thread first_decode([&]() {
        while (pinned.load() >= 0 && pinned.load() < 2) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        if (pinned.load() >= 2) cur_n = decode(0, 0, 0, B + 1);
        decode0_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    const bool have_ctx = tracking_->EnsureBatchContext(w, h, B);
    LZB_PHASE("svo_create done");
    first_decode.join();
    LZB_PHASE("chunk 0 decoded");
    pin_thread.join();
    LZB_PHASE("page-locked buffers allocated");
    svo_ctx *ctx = have_ctx ? tracking_->Context() : nullptr;
    if (!ctx || pinned.load() < 0) {
        if (pinned.load() < 0) LZB_LOG("ERROR", "svo_host_alloc failed (page-locked frame buffers)");
        for (int q = 0; q < 2; q++)
            for (int cam = 0; cam < 2; cam++) if (pin[q][cam]) { svo_host_free(nullptr, pin[q][cam]); pin[q][cam] = nullptr; }
        batch_pin_bytes_ = 0;
        return;
    }
    batch_pin_bytes_ = pin_bytes;
    // the loop's own clock: chunk 0's decode (it ran beside the context creation: its own duration is added at the
    // end) + everything from here to the last pose row
    const auto t_loop = std::chrono::steady_clock::now();
    int k = 0, next = 0;
    next = cur_n;
    current_image_index_ = cur_n > 0 ? 1 : 0;
    if (cur_n > 0 && !record_sink_) WritePose();            // frame 0: StereoInit_f2f, pose = identity
    std::vector<svo_step_result> recs;
    auto flush = [&](double seconds, int pairs) {           // the oldest outstanding chunk's records -> pose file
        recs.clear();
        if (!tracking_->CollectUploaded(recs)) return false;
        if (getenv("LZB_VIO_VERBOSE")) LZB_LOG("INFO", "VO cost time: %f seconds for %d pairs", seconds, pairs);
        for (const auto &r : recs) {
            if (record_sink_) record_sink_->push_back(r); else WritePoseRow(r.pose);
            current_image_index_++;
        }
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
        if (bg.joinable()) bg.join();
        // chunk c+1 crosses PCIe on the copy stream beside chunk c's kernels.  Its upload is queued BEFORE the
        // records of chunk c-1 are waited for: their pose stage runs beside chunk c's kernels and may finish late
        // (its LDS-heavy blocks wait for room behind the LK launch), and a copy queued only after that wait would
        // start when chunk c is nearly done -- the GPU would idle for the length of the copy
        bool more = ok && nn >= 2;
        if (more && !upload(k ^ 1, nn)) { ok = false; more = false; }      // a failed upload is an error, not the end of the sequence
        auto t_now = std::chrono::steady_clock::now();
        if (ok && outstanding == 2) { ok = flush(std::chrono::duration<double>(t_now - t_prev).count(), B); outstanding--; }
        t_prev = t_now;
        if (!ok || !more) break;
        next += nn - 1;
        k ^= 1;
        cur_n = nn;
        chunk++;
    }
    while (ok && outstanding > 0) { ok = flush(0.0, 0); outstanding--; }
    if (!ok && cur_n >= 2) run_failed_ = true;              // an upload / launch / collect failed: the caller must not report success
    svo_sync(ctx);
    loop_seconds_ = decode0_seconds + std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    LZB_PHASE("loop done (last pose row written)");
}

// ---- pipelined stream ----------------------------------------------------------------------------
void System::StreamRelease()
{
    for (int k = 0; k < 2; k++)
        for (int cam = 0; cam < 2; cam++)
            if (stream_.pin[k][cam]) { svo_host_free(nullptr, stream_.pin[k][cam]); stream_.pin[k][cam] = nullptr; }
    stream_.active = false;
}

// the oldest outstanding micro-batch's poses -> stream_.done (block = false: only when they are ready)
bool System::StreamCollect(bool block)
{
    if (tracking_->Outstanding() == 0) return false;
    if (!block && tracking_->ResultsReady() == 0) return false;
    std::vector<svo_step_result> recs;
    if (!tracking_->CollectUploaded(recs)) { stream_.failed = true; return false; }
    for (const auto &r : recs) {
        Pose4x4 P;
        memcpy(P.m, r.pose, sizeof(P.m));
        stream_.done.push_back(P);
    }
    return true;
}

bool System::StreamSubmit()
{
    StreamState &s = stream_;
    svo_ctx *ctx = tracking_->Context();
    if (s.n < 2) return true;
    if (tracking_->Outstanding() == 2 && !StreamCollect(true)) return false;     // at most two in flight
    // after the first micro-batch the halo frame (slot 0) is carried ON THE DEVICE (Tracking::TrackUploadedAsync asks for
    // SVO_CONTINUE_CARRY_FRAME with the chain): it is neither copied on the host nor uploaded again
    const int s0 = s.chunk > 0 ? 1 : 0;
    if (svo_upload_frames_at(ctx, s.buf, s0, s.pin[s.buf][0] + (size_t)s0 * s.fbytes, s.pin[s.buf][1] + (size_t)s0 * s.fbytes, s.pitch,
                             (int64_t)s.fbytes, s.n - s0) != SVO_OK) {
        LZB_LOG("ERROR", "svo_upload_frames_at: %s", svo_last_error(ctx));
        return false;
    }
    s.uploaded[s.buf] = true;
    if (!tracking_->TrackUploadedAsync(s.buf, s.n, s.chunk > 0)) return false;
    s.chunk++;
    // the other buffer takes over: its own upload (two submissions ago) must have left the page-locked memory
    const int nb = s.buf ^ 1;
    if (s.uploaded[nb] && svo_wait_upload(ctx, nb) != SVO_OK) return false;
    s.buf = nb;
    s.n = 1;
    return true;
}

bool System::StreamAcquire(int width, int height, uint8_t **left, uint8_t **right, int *pitch)
{
    StreamState &s = stream_;
    if (s.failed || width <= 0 || height <= 0 || !left || !right || !pitch) return false;
    const int k = stream_depth_ > 0 ? stream_depth_ : 1;
    if (!s.active) {
        if (!tracking_->EnsureBatchContext(width, height, k)) return false;
        svo_ctx *ctx = tracking_->Context();
        s.w = width; s.h = height;
        s.pitch = (s.w + 255) / 256 * 256;
        s.fbytes = (size_t)s.pitch * s.h;
        for (int q = 0; q < 2; q++)
            for (int cam = 0; cam < 2; cam++)
                if (svo_host_alloc(ctx, s.fbytes * (size_t)(k + 1), (void **)&s.pin[q][cam]) != SVO_OK) {
                    LZB_LOG("ERROR", "svo_host_alloc: %s", svo_last_error(ctx));
                    StreamRelease();
                    return false;
                }
        s.active = true;
        s.buf = 0; s.n = 0; s.chunk = 0;
        Pose4x4 I = tracking_->GetPose();                   // frame 0: StereoInit_f2f only, the pose it starts from
        s.done.push_back(I);
    }
    if (width != s.w || height != s.h) {
        LZB_LOG("ERROR", "stream: frame size changed from %dx%d to %dx%d", s.w, s.h, width, height);
        return false;
    }
    *left = s.pin[s.buf][0] + (size_t)s.n * s.fbytes;
    *right = s.pin[s.buf][1] + (size_t)s.n * s.fbytes;
    *pitch = s.pitch;
    return true;
}

bool System::StreamCommit()
{
    StreamState &s = stream_;
    if (!s.active || s.failed) return false;
    const int k = stream_depth_ > 0 ? stream_depth_ : 1;
    s.n++;
    if (s.n == k + 1 && !StreamSubmit()) { s.failed = true; return false; }
    return true;
}

bool System::StreamPush(Frame::Ptr frame)
{
    if (frame == nullptr || stream_.failed) return false;
    const cv::Mat &L = frame->left_img_, &R = frame->right_img_;
    if (L.empty() || R.empty() || L.cols != R.cols || L.rows != R.rows) return false;
    uint8_t *dst[2];
    int pitch = 0;
    if (!StreamAcquire(L.cols, L.rows, &dst[0], &dst[1], &pitch)) return false;
    const cv::Mat *img[2] = {&L, &R};
    for (int cam = 0; cam < 2; cam++)
        for (int y = 0; y < L.rows; y++) memcpy(dst[cam] + (size_t)y * pitch, img[cam]->ptr(y), (size_t)L.cols);
    return StreamCommit();
}

int System::StreamPoll(std::vector<Pose4x4> &poses, bool wait)
{
    while (StreamCollect(false)) {}
    // wait: drain every outstanding micro-batch, also after a failed push / submission (a failed COLLECT ends it)
    if (wait) while (tracking_->Outstanding() > 0 && StreamCollect(true)) {}
    const int n = (int)stream_.done.size();
    poses.insert(poses.end(), stream_.done.begin(), stream_.done.end());
    stream_.done.clear();
    return n;
}

bool System::StreamFlush()
{
    if (!stream_.active || stream_.failed) return stream_.active && !stream_.failed;
    return StreamSubmit();
}

void System::CloseOutputs()
{
    if (pose_file_) { fclose(pose_file_); pose_file_ = nullptr; }
    if (tracks_file_) { fclose(tracks_file_); tracks_file_ = nullptr; }
    if (tracking_ && tracking_->Context()) svo_sync(tracking_->Context());
}

void System::Shutdown() {}
void System::Reset() {}

// ---- several sequences on the node's devices ---------------------------------------------------------
int RunSequences(const std::vector<std::string> &yamls, const std::vector<std::string> &pose_files, int n_devices,
                 std::vector<SequenceReport> *report)
{
    const int n_seq = (int)yamls.size();
    if (n_seq == 0) return 0;
    if (n_devices <= 0 && (svo_device_count(&n_devices) != SVO_OK || n_devices <= 0)) {
        LZB_LOG("ERROR", "no HIP device: %d sequences not run (there is no CPU path)", n_seq);
        return n_seq;
    }
    // Config is a process-wide singleton, as in the reference (src/config.cpp:26): the Systems are BUILT one after
    // the other on this thread -- each reads its own YAML completely in its constructor -- and only then run
    std::vector<std::unique_ptr<System>> sys((size_t)n_seq);
    std::vector<int> frames((size_t)n_seq, 0);
    std::vector<char> no_output((size_t)n_seq, 0);        // the sequence's pose file could not be opened: it is NOT run
    for (int i = 0; i < n_seq; i++) {
        std::string path = yamls[(size_t)i];
        sys[(size_t)i].reset(new System(path));
        if (i < (int)pose_files.size() && !pose_files[(size_t)i].empty() && !sys[(size_t)i]->SetPoseFile(pose_files[(size_t)i])) {
            LZB_LOG("ERROR", "cannot open %s for writing: sequence %d is not run", pose_files[(size_t)i].c_str(), i);
            no_output[(size_t)i] = 1;
        }
        frames[(size_t)i] = sys[(size_t)i]->CountFrames();
    }
    // one worker per device; a lone device gets two (one sequence decodes while the other's kernels run)
    const int n_workers = n_devices == 1 ? (n_seq > 1 ? 2 : 1) : (n_devices < n_seq ? n_devices : n_seq);
    // longest-processing-time-first deal (ties: lower sequence index, lower worker)
    std::vector<int> order((size_t)n_seq);
    for (int i = 0; i < n_seq; i++) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return frames[(size_t)a] > frames[(size_t)b]; });
    std::vector<std::vector<int>> mine((size_t)n_workers);
    std::vector<long> load((size_t)n_workers, 0);
    for (int s : order) {
        int best = 0;
        for (int wk = 1; wk < n_workers; wk++) if (load[(size_t)wk] < load[(size_t)best]) best = wk;
        mine[(size_t)best].push_back(s);
        load[(size_t)best] += frames[(size_t)s] > 1 ? frames[(size_t)s] - 1 : 0;
    }
    std::vector<SequenceReport> rep((size_t)n_seq);
    std::vector<double> busy((size_t)n_workers, 0.0);
    std::vector<std::thread> pool;
    for (int wk = 0; wk < n_workers; wk++)
        pool.emplace_back([&, wk]() {
            const int dev = wk % n_devices;
            for (int s : mine[(size_t)wk]) {
                SequenceReport &r = rep[(size_t)s];
                r.yaml = yamls[(size_t)s]; r.device = dev; r.worker = wk;
                if (no_output[(size_t)s]) { sys[(size_t)s].reset(); continue; }      // r.ok stays false: counted as failed
                const auto t0 = std::chrono::steady_clock::now();
                sys[(size_t)s]->SetDevice(dev);
                sys[(size_t)s]->Run();
                r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                r.frames = sys[(size_t)s]->FramesProcessed();
                r.ok = r.frames > 0 && r.frames == frames[(size_t)s];
                busy[(size_t)wk] += r.seconds;
                sys[(size_t)s].reset();                          // closes the pose file, releases the context
            }
        });
    for (auto &th : pool) th.join();
    int failed = 0;
    for (int wk = 0; wk < n_workers; wk++) {
        std::string list;
        for (int s : mine[(size_t)wk]) list += (list.empty() ? "" : ",") + std::to_string(s);
        fprintf(stderr, "worker %d (device %d): sequences [%s], %ld pairs, busy %.3f s\n", wk, wk % n_devices, list.c_str(),
                load[(size_t)wk], busy[(size_t)wk]);
    }
    for (const auto &r : rep) failed += r.ok ? 0 : 1;
    if (report) *report = rep;
    return failed;
}

// ---- one sequence, frame pairs dealt to several contexts ------------------------------------------------
int RunSplitPairs(const std::string &yaml, const std::string &pose_file, int n_parts, int n_devices,
                  std::vector<SequenceReport> *report)
{
    if (n_devices <= 0 && (svo_device_count(&n_devices) != SVO_OK || n_devices <= 0)) {
        LZB_LOG("ERROR", "no HIP device: %s not run (there is no CPU path)", yaml.c_str());
        return 1;
    }
    std::string path = yaml;
    // The probe and the chunk Systems hand their records to a sink: none of them may open (= truncate) the pose_file /
    // tracks_file the YAML names.  The poses of the whole sequence go to `pose_file`, or to the YAML's pose_file when the
    // command line gives none.
    System::DeferYamlOutputs(true);
    std::unique_ptr<System> probe(new System(path));
    const int n_frames = probe->CountFrames();
    const int n_pairs = n_frames - 1;
    const int default_batch = probe->BatchSize() > 1 ? probe->BatchSize() : 256;
    const std::string out_path = !pose_file.empty() ? pose_file : probe->YamlPoseFile();
    if (!probe->YamlTracksFile().empty())
        LZB_LOG("WARNING", "tracks_file is not written by --split-pairs runs (the chunks keep relative motions only)%s", "");
    probe.reset();
    if (n_pairs < 1) { System::DeferYamlOutputs(false); LZB_LOG("ERROR", "%s: fewer than two stereo frames", yaml.c_str()); return 1; }
    if (n_parts < 1) n_parts = 1;
    if (n_parts > n_pairs) n_parts = n_pairs;
    FILE *out = nullptr;
    if (!out_path.empty() && !(out = fopen(out_path.c_str(), "w"))) {
        System::DeferYamlOutputs(false);
        LZB_LOG("ERROR", "cannot open %s for writing", out_path.c_str());
        return 1;
    }
    // contiguous chunks, sizes differing by at most one (multigpu.shard_pairs); chunk c = pairs first[c] .. first[c] + n[c] - 1
    // = frames first[c] .. first[c] + n[c]
    std::vector<std::unique_ptr<System>> sys((size_t)n_parts);
    std::vector<std::vector<svo_step_result>> recs((size_t)n_parts);
    std::vector<int> first((size_t)n_parts), cnt((size_t)n_parts);
    const int base = n_pairs / n_parts, extra = n_pairs % n_parts;
    // Workers, not chunks, set the footprint: a chunk's System holds a context with its arena, four page-locked buffers and a
    // decoder pool while it runs.  Two workers a device (one decodes while the other's kernels run) pull chunks from a queue;
    // the usable cores are divided among them.  Chunk c runs on device c % n_devices whoever takes it.
    const int n_workers = std::min(n_parts, 2 * n_devices);
    const int dec_threads = std::max(1, usable_cores() / n_workers);
    for (int c = 0; c < n_parts; c++) {
        first[(size_t)c] = c * base + (c < extra ? c : extra);
        cnt[(size_t)c] = base + (c < extra ? 1 : 0);
        std::string p = yaml;
        sys[(size_t)c].reset(new System(p));             // built one after the other: Config is process-wide
        sys[(size_t)c]->SetFrameRange(first[(size_t)c], first[(size_t)c] + cnt[(size_t)c]);
        sys[(size_t)c]->SetRecordSink(&recs[(size_t)c]);
        sys[(size_t)c]->SetBatchSize(default_batch < cnt[(size_t)c] ? default_batch : (cnt[(size_t)c] > 1 ? cnt[(size_t)c] : 2));
        sys[(size_t)c]->SetDevice(c % n_devices);
        sys[(size_t)c]->SetDecodeThreads(dec_threads);
    }
    System::DeferYamlOutputs(false);
    std::vector<SequenceReport> rep((size_t)n_parts);
    std::vector<std::thread> pool;
    std::atomic<int> next_chunk{0};
    for (int wk = 0; wk < n_workers; wk++)
        pool.emplace_back([&, wk]() {
            for (int c = next_chunk.fetch_add(1); c < n_parts; c = next_chunk.fetch_add(1)) {
                SequenceReport &r = rep[(size_t)c];
                r.yaml = yaml; r.device = c % n_devices; r.worker = wk;
                const auto t0 = std::chrono::steady_clock::now();
                sys[(size_t)c]->Run();
                r.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                r.frames = (int)recs[(size_t)c].size() + 1;
                r.ok = (int)recs[(size_t)c].size() == cnt[(size_t)c] && !sys[(size_t)c]->Failed();
                if (c > 0) sys[(size_t)c].reset();           // its records are in the sink; chunk 0's context does the prefix product
            }
        });
    for (auto &th : pool) th.join();
    int failed = 0;
    for (int c = 0; c < n_parts; c++) {
        fprintf(stderr, "chunk %d (device %d, worker %d): frames %d..%d, %d pairs, %.3f s%s\n", c, rep[(size_t)c].device, rep[(size_t)c].worker,
                first[(size_t)c], first[(size_t)c] + cnt[(size_t)c], cnt[(size_t)c], rep[(size_t)c].seconds, rep[(size_t)c].ok ? "" : "  [FAILED]");
        failed += rep[(size_t)c].ok ? 0 : 1;
    }
    if (report) *report = rep;
    if (failed) { if (out) fclose(out); return failed; }
    // the gather (17 doubles a pair) and the one serial step: the prefix product, on chunk 0's context
    std::vector<double> T((size_t)n_pairs * 16), poses((size_t)n_pairs * 16);
    std::vector<int32_t> okv((size_t)n_pairs);
    size_t p = 0;
    for (int c = 0; c < n_parts; c++)
        for (const auto &r : recs[(size_t)c]) { memcpy(&T[p * 16], r.T_rel_inv, sizeof(double) * 16); okv[p] = r.ok; p++; }
    svo_ctx *ctx = sys[0]->GetTracking()->Context();
    if (!ctx || svo_chain_relative(ctx, T.data(), okv.data(), n_pairs, nullptr, poses.data(), SVO_MEM_HOST) != SVO_OK) {
        LZB_LOG("ERROR", "svo_chain_relative: %s", ctx ? svo_last_error(ctx) : "no context");
        if (out) fclose(out);
        return 1;
    }
    if (out) {
        const double I[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        for (int i = 0; i < 12; i++) fprintf(out, "%.9e%c", I[i], i == 11 ? '\n' : ' ');
        for (int q = 0; q < n_pairs; q++)
            for (int i = 0; i < 12; i++) fprintf(out, "%.9e%c", poses[(size_t)q * 16 + i], i == 11 ? '\n' : ' ');
        fclose(out);
    }
    return 0;
}

}  // namespace lzb_vio

// System.h -- facade + KITTI reader (reference include/lzb_vio/System.h:19-47).  Same public
// surface: System(std::string&), Run(), Step(), Step_ros(Frame::Ptr).  Additive: SetPoseFile().
#pragma once
#ifndef lzb_vio_System_H
#define lzb_vio_System_H

#include "lzb_vio/common_include.h"
#include "lzb_vio/frame.h"
#include "lzb_vio/parameter.h"
#include "lzb_vio/tracking.h"

namespace lzb_vio {

class System {
public:
    explicit System(std::string &config_path);
    ~System();
    void Run();
    bool Step();
    bool Step_ros(Frame::Ptr new_frame);

    // additive: batched runner.  Frames are decoded by `decode_threads`
    // threads into page-locked memory while the GPU tracks the previous batch of `batch_size` frame
    // pairs; poses come out exactly as Run() would produce them.  Run() uses it when the YAML has
    // batch_size > 1 (additive keys batch_size / decode_threads; absent = the reference's per-frame loop).
    void RunBatched(int batch_size, int decode_threads);

    // additive: a PIPELINED stream behind Step_ros (reference src/System.cpp:60-74 feeds one externally supplied frame per
    // call and blocks until its pose exists).  Consecutive pairs are independent, so a live stream may
    // trade latency for throughput: frames are gathered into micro-batches of `stream_depth` pairs in page-locked
    // memory; a full micro-batch is uploaded and launched WITHOUT waiting (up to two are in flight, the pose chain
    // continues on the device), and the poses come back `stream_depth` to 3 x `stream_depth` frames late, byte for byte
    // the per-frame loop's.  YAML key `stream_depth: k` (k >= 1) routes Step_ros and Run() through it.
    //   StreamPush  : hands one frame over (the images are copied; the frame may be reused at once)
    //   StreamPoll  : appends the poses (row-major 4x4, frame order, frame 0 = identity included) completed since the
    //                 last call; wait = true blocks until everything submitted so far is there
    //   StreamFlush : submits the partial micro-batch, if any; follow it with StreamPoll(poses, true)
    bool StreamPush(Frame::Ptr frame);
    //   StreamAcquire / StreamCommit : StreamPush without its copy, for a producer that can write the images where they
    //                 are uploaded from (a camera driver, a decoder): Acquire hands out the page-locked rows of the NEXT
    //                 frame (`pitch` bytes apart, width x height), Commit says they are complete
    bool StreamAcquire(int width, int height, uint8_t **left, uint8_t **right, int *pitch);
    bool StreamCommit();
    int StreamPoll(std::vector<Pose4x4> &poses, bool wait = false);
    bool StreamFlush();
    int StreamDepth() const { return stream_depth_; }
    void SetStreamDepth(int k) { if (!stream_.active) stream_depth_ = k; }

    // additive: write one KITTI-format pose row (12 numbers of [R|t]) per frame
    bool SetPoseFile(const std::string &path);
    // additive: per-frame track dump (text), the headless replacement of displayTracking
    bool SetTracksFile(const std::string &path);
    Tracking::Ptr GetTracking() { return tracking_; }
    int FramesProcessed() const { return current_image_index_; }
    // additive: seconds RunBatched spent from its first decode to its last pose row (decode + H2D + tracking + pose
    // file; process start, context creation and buffer allocation excluded); 0 for the per-frame loop
    double LoopSeconds() const { return loop_seconds_; }
    // additive: the HIP device this System's context lives on (before the first frame; default 0), and the
    // sequence it will read: the dataset directory and the number of consecutive stereo frames found there
    void SetDevice(int device) { tracking_->SetDevice(device); }
    // additive (SURVEY.md 8e granularity 2 / 8f rank 1, RunSplitPairs below): this System tracks only the frames
    // first .. last of its sequence -- a contiguous chunk of frame pairs with its one-frame halo -- through the batched
    // runner, and hands the step records (relative motions) to `sink` instead of writing pose rows
    void SetFrameRange(int first, int last) { frame_base_ = first; frame_last_ = last; }
    void SetRecordSink(std::vector<svo_step_result> *sink) { record_sink_ = sink; }
    void SetBatchSize(int b) { batch_size_ = b; }
    void SetDecodeThreads(int t) { decode_threads_ = t; }
    // additive: the output files the YAML names (pose_file / tracks_file keys; empty: none), and a switch for constructors
    // that must not open them (a probe, or chunk Systems whose records go to a sink): set it, construct, clear it
    const std::string &YamlPoseFile() const { return yaml_pose_file_; }
    const std::string &YamlTracksFile() const { return yaml_tracks_file_; }
    static void DeferYamlOutputs(bool on) { s_defer_yaml_outputs = on; }
    int BatchSize() const { return batch_size_; }
    bool Failed() const { return run_failed_; }
    void CloseOutputs();                                     // flushes and closes the pose / tracks files, waits for the device (fast exit)              // a stream / batch submission failed: the pose file is short
    const std::string &DatasetPath() const { return dataset_path_; }
    int CountFrames() const;

private:
    TrackingStatus GetFrontendStatus() const { return tracking_->GetStatus(); }
    Frame::Ptr NextFrame_kitti();
    void Shutdown();
    void Reset();
    void WritePose();
    void WritePoseRow(const double *pose16);
    void WriteTracks();
    bool ReadStereo(int index, cv::Mat &left, cv::Mat &right);

    std::string config_file_path_;
    Parameter::Ptr init_parameter_ = nullptr;
    Sensors::Ptr sensors_ = nullptr;
    Tracking::Ptr tracking_ = nullptr;
    int current_image_index_ = 0;
    bool inited_ = false;
    std::string dataset_path_;
    FILE *pose_file_ = nullptr, *tracks_file_ = nullptr;
    std::string yaml_pose_file_, yaml_tracks_file_;
    static bool s_defer_yaml_outputs;
    // additive YAML keys batch_size / decode_threads, read once in the constructor (Config is process-wide:
    // another System may have loaded ITS file by the time Run() is called)
    int batch_size_ = 1, decode_threads_ = 0;
    double loop_seconds_ = 0;
    int frame_base_ = 0, frame_last_ = -1;                   // SetFrameRange (last < 0: to the end of the sequence)
    std::vector<svo_step_result> *record_sink_ = nullptr;
    bool run_failed_ = false;
    // RunBatched's page-locked chunk buffers: kept for the next run of this System, released with it (un-pinning half a
    // gigabyte is a fifth of a short run's wall time, and a process about to exit need not do it)
    uint8_t *batch_pin_[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
    size_t batch_pin_bytes_ = 0;
    // pipelined stream (StreamPush / StreamPoll)
    int stream_depth_ = 0;
    struct StreamState {
        bool active = false, failed = false;
        int w = 0, h = 0, pitch = 0;
        size_t fbytes = 0;
        uint8_t *pin[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
        int buf = 0, n = 0, chunk = 0;         // current page-locked buffer, frames in it, micro-batches launched
        bool uploaded[2] = {false, false};
        std::vector<Pose4x4> done;             // completed poses not yet handed out
    } stream_;
    bool StreamSubmit();
    bool StreamCollect(bool block);
    void StreamRelease();
};

// 8-bit grayscale image readers used by NextFrame_kitti: binary PGM (P5) and PNG (8-bit gray or
// RGB/RGBA converted with the BT.601 weights cv::imread(IMREAD_GRAYSCALE) uses; zlib inflate).
bool ReadImageGray(const std::string &path, cv::Mat &out);
// the same straight into caller-owned rows (`pitch` bytes apart) of a w x h image; a file of another size is refused
bool ReadImageGrayInto(const std::string &path, uint8_t *dst, int pitch, int w, int h);

// additive: several sequences -- one YAML each, as `run_kitti_stereo a.yaml b.yaml ...` -- dealt
// longest-first to the HIP devices of the node (svo_device_count; n_devices > 0 uses that many), one worker thread +
// System + context per device running its sequences back to back; on a single device two workers share the card, so
// one sequence's image decode overlaps the other's kernels.  pose_files[i] (may be empty) receives sequence i's poses,
// byte for byte what a single-sequence run writes.  Returns 0, or the number of sequences that could not be run.
struct SequenceReport { std::string yaml; int device = 0, worker = 0, frames = 0; double seconds = 0; bool ok = false; };
int RunSequences(const std::vector<std::string> &yamls, const std::vector<std::string> &pose_files, int n_devices,
                 std::vector<SequenceReport> *report);

// additive (SURVEY.md 8e granularity 2 / 8f rank 1): ONE sequence cut into `n_parts` contiguous chunks of frame pairs with a
// one-frame halo (chunk c needs the frame before its first pair), chunk c -> its own worker thread, System and context on
// device c % n_devices; every chunk is tracked by the batched runner from the identity, its relative motions
// (svo_step_result.T_rel_inv, ok) are gathered and chained ONCE by svo_chain_relative -- the `frame_pose_ *= T.inv()`
// recurrence of reference src/tracking.cpp:318 --, and the pose file is byte for byte what the single-context run
// writes.  What `bench.py --shard pairs` does with one process per GPU; removes the sequence-length imbalance of
// RunSequences (KITTI 00-07 on 8 devices: max / mean 2.28).  `run_kitti_stereo cfg.yaml [poses] --split-pairs N [--devices D]`.
// Returns 0 on success; report (may be null) gets one entry per chunk.
int RunSplitPairs(const std::string &yaml, const std::string &pose_file, int n_parts, int n_devices,
                  std::vector<SequenceReport> *report);

}  // namespace lzb_vio
#endif

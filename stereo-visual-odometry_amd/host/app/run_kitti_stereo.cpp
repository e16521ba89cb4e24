// run_kitti_stereo <config.yaml> -- drop-in entry point (reference app/run_kitti_stereo.cpp).
// Unlike the reference it returns non-zero on a wrong argument count instead of dereferencing
// argv[1].  Optional second argument: pose file (KITTI format).
//
// Additive: run_kitti_stereo a.yaml b.yaml ... [--poses-dir DIR] [--devices N]
//   several sequences (one YAML each) dealt longest-first to the node's HIP devices, one worker thread, System and
//   context per device (lzb_vio::RunSequences; SURVEY.md 8e: "degrades to hipGetDeviceCount() devices").  A sequence's
//   poses go to its YAML's pose_file key, else to DIR/<yaml basename>.poses.txt when --poses-dir is given.
// Additive: run_kitti_stereo cfg.yaml [poses.txt] --split-pairs N [--devices D]
//   ONE sequence cut into N contiguous chunks of frame pairs (one-frame halo), a context per chunk on device chunk % D, the
//   relative motions chained once (lzb_vio::RunSplitPairs; SURVEY.md 8e granularity 2): same pose file, no length imbalance.
#include "lzb_vio/System.h"
#include <unistd.h>

// Orderly teardown is the default: the System is destroyed (pose / tracks files closed, context freed), the HIP runtime's
// exit handlers run.  LZB_VIO_FAST_EXIT=1 opts into leaving the process the moment its output is on disk -- un-pinning half
// a gigabyte of frame buffers and tearing the runtime down is a fifth of a short run's wall time and nothing the operating
// system does not do anyway: the files are closed by the System's destructor-equivalent (CloseOutputs) and the device is
// synchronised first, so nothing buffered can be lost.
static int finish(lzb_vio::System *vo, int code)
{
    if (getenv("LZB_VIO_FAST_EXIT")) {
        if (vo) vo->CloseOutputs();                          // fclose of the pose / tracks files + svo_sync
        fflush(nullptr);
        LZB_PHASE("exit (teardown left to the OS)");
        _exit(code);
    }
    delete vo;
    LZB_PHASE("System destroyed (context freed)");
    return code;
}

static bool is_yaml(const std::string &s)
{
    auto ends = [&](const char *e) { const size_t n = strlen(e); return s.size() >= n && s.compare(s.size() - n, n, e) == 0; };
    return ends(".yaml") || ends(".yml");
}

int main(int argc, char **argv)
{
    std::vector<std::string> yamls, rest;
    std::string poses_dir;
    int devices = 0, split = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        if (a == "--poses-dir" && i + 1 < argc) poses_dir = argv[++i];
        else if (a == "--devices" && i + 1 < argc) devices = atoi(argv[++i]);
        else if (a == "--split-pairs" && i + 1 < argc) split = atoi(argv[++i]);
        else if (is_yaml(a)) yamls.push_back(a);
        else rest.push_back(a);
    }
    if (yamls.size() >= 2 && rest.empty()) {
        std::vector<std::string> pose_files;
        for (const auto &y : yamls) {
            std::string f;
            if (!poses_dir.empty()) {
                const size_t slash = y.find_last_of('/');
                f = poses_dir + "/" + (slash == std::string::npos ? y : y.substr(slash + 1)) + ".poses.txt";
            }
            pose_files.push_back(f);
        }
        std::vector<lzb_vio::SequenceReport> rep;
        const int failed = lzb_vio::RunSequences(yamls, pose_files, devices, &rep);
        for (const auto &r : rep)
            fprintf(stderr, "%s: device %d, %d frames, %.3f s%s\n", r.yaml.c_str(), r.device, r.frames, r.seconds, r.ok ? "" : "  [FAILED]");
        return finish(nullptr, failed ? 1 : 0);
    }
    if (split > 0 && yamls.size() == 1 && rest.size() <= 1 && poses_dir.empty()) {
        std::vector<lzb_vio::SequenceReport> rep;
        const int failed = lzb_vio::RunSplitPairs(yamls[0], rest.empty() ? std::string() : rest[0], split, devices, &rep);
        return finish(nullptr, failed ? 1 : 0);
    }
    if (argc < 2 || argc > 3 || yamls.size() > 1 || !poses_dir.empty() || devices || split) {
        fprintf(stderr, "usage: %s config.yaml [poses.txt]\n       %s a.yaml b.yaml ... [--poses-dir DIR] [--devices N]\n"
                        "       %s config.yaml [poses.txt] --split-pairs N [--devices D]\n", argv[0], argv[0], argv[0]);
        return 2;
    }
    std::string config_file_path = argv[1];
    LZB_PHASE("main");
    lzb_vio::System *vo = new lzb_vio::System(config_file_path);
    LZB_PHASE("System constructed (YAML read)");
    if (argc == 3 && !vo->SetPoseFile(argv[2])) {
        fprintf(stderr, "cannot open %s for writing\n", argv[2]);
        return 2;
    }
    vo->Run();
    fprintf(stderr, "processed %d frames\n", vo->FramesProcessed());
    if (vo->LoopSeconds() > 0)
        fprintf(stderr, "batched loop: %d pairs in %.6f s\n", vo->FramesProcessed() - 1, vo->LoopSeconds());
    if (vo->Failed()) fprintf(stderr, "run FAILED: a batch or stream submission failed, the pose file is incomplete\n");
    return finish(vo, vo->Failed() ? 1 : 0);
}

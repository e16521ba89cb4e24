// run_kitti_stereo <config.yaml> -- drop-in entry point (reference app/run_kitti_stereo.cpp).
// Unlike the reference it returns non-zero on a wrong argument count instead of dereferencing
// argv[1].  Optional second argument: pose file (KITTI format).
#include "lzb_vio/System.h"

int main(int argc, char **argv)
{
    if (argc < 2 || argc > 3) {
        fprintf(stderr, "usage: %s config.yaml [poses.txt]\n", argv[0]);
        return 2;
    }
    std::string config_file_path = argv[1];
    lzb_vio::System *vo = new lzb_vio::System(config_file_path);
    if (argc == 3 && !vo->SetPoseFile(argv[2])) {
        fprintf(stderr, "cannot open %s for writing\n", argv[2]);
        return 2;
    }
    vo->Run();
    fprintf(stderr, "processed %d frames\n", vo->FramesProcessed());
    delete vo;
    return 0;
}

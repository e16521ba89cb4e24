#!/usr/bin/env python3
"""Driver of tools/gpu/lk_stamps.sh: 256 S0 pairs through svo_track_batch with a -DSVO_LK_STAMP=k library; prints the
cycles all waves spent in section k, the number of times the section ran and the kernel's time with the stamps in."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

NAMES = {0: "level control + I-tile staging + J requests", 1: "patch build (4 slots)",
         2: "J stores + A reduction + 2x2 set-up + next level's I requests", 4: "iteration: control (floor, weights, restage test)",
         5: "iteration: slot pixel work (J reads, bilinear + mismatch dot products)", 6: "iteration: reduce-scatter + solve + convergence",
         8: "a whole cv::calcOpticalFlowPyrLK call of four points"}


def main():
    import torch
    k = int(sys.argv[1])
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    lib = pkg.load_library()
    if not hasattr(lib, "svo_debug_lk_stamps"):
        raise SystemExit("the library was not built with -DSVO_LK_STAMP=k (run tools/gpu/lk_stamps.sh)")
    dev = torch.device("cuda", 0)
    W, H, B = 1241, 376, 256
    cache = "/tmp/lk_stamp_frames.pt"
    if os.path.exists(cache):
        L, R = torch.load(cache)
        L, R = L.to(dev), R.to(dev)
    else:
        seq = synth.StereoSequence(width=W, height=H, n_frames=B + 1, seed=20200710, device=dev)
        L = torch.zeros((B + 1, H, W), dtype=torch.uint8, device=dev)
        R = torch.zeros((B + 1, H, W), dtype=torch.uint8, device=dev)
        for f in range(B + 1):
            L[f], R[f] = seq.render(f)
        torch.save((L.cpu(), R.cpu()), cache)
    P1, P2 = synth.proj_matrices()
    c = pkg.Context(W, H, device=0, max_batch=B, P1=P1, P2=P2)
    out = (C.c_ulonglong * 2)()
    c.track_batch(L, R)
    lib.svo_debug_lk_stamps(out, 1)                # warm-up launch discarded
    c.enable_timing(True)
    c.track_batch(L, R)
    c.sync()
    ms = dict(c.get_timing()).get("lk")
    lib.svo_debug_lk_stamps(out, 1)
    print(f"section {k}: {out[0] / 1e9:8.3f} G wave-cycles in {out[1] / 1e6:8.3f} M passes ({out[0] / max(out[1], 1):7.0f} cycles per pass); "
          f"kernel {ms:.2f} ms with these stamps   [{NAMES[k]}]", flush=True)
    c.close()


if __name__ == "__main__":
    main()

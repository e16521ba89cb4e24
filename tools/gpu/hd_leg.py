#!/usr/bin/env python3
"""The bench's hd leg alone (BASELINE config #4: 1920x1080, exactly 2000 corners per frame), for the profiler:
python3 tools/gpu/hd_leg.py [batch=128] [steps=4]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)
    seq = synth.StereoSequence(width=1920, height=1080, n_frames=2 * B + 1, seed=1, device=dev)
    L = torch.zeros((2 * B + 1, 1080, 1920), dtype=torch.uint8, device=dev)
    R = torch.zeros_like(L)
    for f in range(2 * B + 1):
        L[f], R[f] = seq.render(f)
    P1, P2 = seq.proj()
    ctx, el, st_ms, recs, _ = bench.run_leg(pkg, torch, dev, L, R, 1920, 1080, B, steps, 1,
                                            dict(P1=P1, P2=P2, max_keypoints=1 << 16, fast_keep_strongest=2000))
    print(f"hd leg: {B * steps / el:.0f} pairs/s, {1e3 * el / steps:.3f} ms/step, stages {st_ms}, ok {int(recs['ok'].sum())}/{B}")
    ctx.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Distribution of RANSAC iteration counts after the adaptive stop (svo_step_result.ransac_iters) over one
batch of S0 pairs, per track mode: how much of a 448-hypothesis phase is speculative work.
Usage: python tools/ransac_iters_hist.py [pairs=256]"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)
    W, H = 1241, 376
    seq = synth.StereoSequence(width=W, height=H, n_frames=B + 1, seed=20200710, device=dev)
    L = torch.zeros((B + 1, H, W), dtype=torch.uint8, device=dev)
    R = torch.zeros((B + 1, H, W), dtype=torch.uint8, device=dev)
    for f in range(B + 1):
        L[f], R[f] = seq.render(f)
    P1, P2 = seq.proj()
    for mode in ("lk", "orb"):
        kw = dict(P1=P1, P2=P2)
        if mode == "orb":
            kw.update(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
        c = pkg.Context(W, H, device=0, max_batch=B, **kw)
        r = c.track_batch(L, R)
        c.close()
        it = r["ransac_iters"].astype(int)
        edges = [0, 16, 32, 64, 96, 128, 192, 256, 384, 500, 10 ** 9]
        hist = {f"<= {edges[i + 1]}": int(((it > edges[i]) & (it <= edges[i + 1])).sum()) for i in range(len(edges) - 1)}
        print(json.dumps({"mode": mode, "pairs": B, "ok": int(r["ok"].sum()), "mean_tracks": float(r["n_tracked"].mean()),
                          "mean_inliers": float(r["n_inliers"].mean()), "ransac_iters": {"mean": float(it.mean()), "median": float(np.median(it)),
                                                                                        "max": int(it.max()), "hist": hist}}))


if __name__ == "__main__":
    main()

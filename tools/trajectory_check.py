#!/usr/bin/env python3
"""End-to-end accuracy on a rendered sequence with ground truth: tracks N
frames of the synthetic corridor in batches, in LK and ORB mode, and compares every estimated
relative motion and the accumulated trajectory with the renderer's ground truth.
Usage: python tools/trajectory_check.py [n_frames=301] [batch=100]      One JSON line per mode."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 301
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)
    W, H = 1241, 376
    seq = synth.StereoSequence(width=W, height=H, n_frames=n, seed=20200710, device=dev)
    L = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    R = torch.zeros((n, H, W), dtype=torch.uint8, device=dev)
    for f in range(n):
        L[f], R[f] = seq.render(f)
    P1, P2 = seq.proj()
    gt_wc = seq.poses_wc().numpy()                     # camera-to-world of every frame
    gt0 = np.linalg.inv(gt_wc[0])
    for mode in ("lk", "orb"):
        kw = dict(P1=P1, P2=P2)
        if mode == "orb":
            kw.update(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
        c = pkg.Context(W, H, device=0, max_batch=B, **kw)
        pose = np.eye(4)
        recs = []
        for f0 in range(0, n - 1, B):
            f1 = min(f0 + B, n - 1)
            r = c.track_batch(L[f0:f1 + 1], R[f0:f1 + 1], pose0=pose)
            recs.append(r)
            pose = r["pose"][-1].reshape(4, 4)
        res = np.concatenate(recs)
        c.close()
        te, re_ = [], []
        for t in range(1, n):
            if not res["ok"][t - 1]:
                continue
            Tg = seq.relative_gt(t).numpy()
            Te = np.eye(4)
            Te[:3, :3] = res["R"][t - 1].reshape(3, 3)
            Te[:3, 3] = res["tvec"][t - 1]
            te.append(np.linalg.norm(Te[:3, 3] - Tg[:3, 3]))
            dR = Te[:3, :3] @ Tg[:3, :3].T
            re_.append(np.degrees(np.arccos(np.clip((np.trace(dR) - 1) / 2, -1, 1))))
        est_end = res["pose"][-1].reshape(4, 4)        # frame_pose_: camera n-1 in frame-0 coordinates
        gt_end = gt0 @ gt_wc[n - 1]
        path = float(np.sum(np.linalg.norm(np.diff(gt_wc[:, :3, 3], axis=0), axis=1)))
        print(json.dumps({"mode": mode, "frames": n, "pairs_ok": int(res["ok"].sum()), "pairs": n - 1,
                          "mean_inliers": round(float(res["n_inliers"].mean()), 1),
                          "rel_translation_err_m": {"mean": round(float(np.mean(te)), 5), "max": round(float(np.max(te)), 5)},
                          "rel_rotation_err_deg": {"mean": round(float(np.mean(re_)), 5), "max": round(float(np.max(re_)), 5)},
                          "path_length_m": round(path, 2),
                          "end_point_drift_m": round(float(np.linalg.norm(est_end[:3, 3] - gt_end[:3, 3])), 4),
                          "drift_percent_of_path": round(100 * float(np.linalg.norm(est_end[:3, 3] - gt_end[:3, 3])) / path, 4)}),
              flush=True)


if __name__ == "__main__":
    main()

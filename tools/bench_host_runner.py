#!/usr/bin/env python3
"""End-to-end run_kitti_stereo throughput (decode + H2D + tracking + pose file) on a synthetic
KITTI-layout dataset written to a scratch directory: the reference's per-frame loop (batch_size
absent) against the batched runner (batch_size / decode_threads), for PGM and PNG frames.
Usage: python tools/bench_host_runner.py [n_frames=300] [scratch_dir=/tmp/svo_ds]"""
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from PIL import Image
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    scratch = sys.argv[2] if len(sys.argv) > 2 else "/tmp/svo_ds"
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    host = os.path.join(entry.PKG_DIR, "host")
    subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
    import test_host_api as tha
    seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=torch.device("cuda", 0))
    for fmt in ("pgm", "png"):
        d = os.path.join(scratch, fmt)
        for cam in (0, 1):
            os.makedirs(os.path.join(d, f"image_{cam}"), exist_ok=True)
        for t in range(n):
            l, r = [x.cpu().numpy() for x in seq.render(t)]
            for cam, im in ((0, l), (1, r)):
                path = os.path.join(d, f"image_{cam}", f"{t:06d}.{fmt}")
                if fmt == "pgm":
                    tha._write_pgm(path, im)
                else:
                    Image.fromarray(im).save(path, compress_level=3)
        tha._write_yaml(os.path.join(d, "online.yaml"), d, fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy)
        base = open(os.path.join(d, "online.yaml"), encoding="utf-8").read()
        with open(os.path.join(d, "batched.yaml"), "w", encoding="utf-8") as f:
            f.write(base + "batch_size: 128\n")       # decode_threads absent: every core the process may use
        poses = {}
        for name in (("batched",) if os.environ.get("SVO_SKIP_ONLINE") else ("online", "batched")):
            t0 = time.perf_counter()
            r = subprocess.run([os.path.join(host, "run_kitti_stereo"), os.path.join(d, name + ".yaml"),  # (orderly teardown: the default)
                                os.path.join(d, name + ".txt")], capture_output=True)
            el = time.perf_counter() - t0
            assert r.returncode == 0, r.stderr.decode()
            poses[name] = np.loadtxt(os.path.join(d, name + ".txt"))
            print(json.dumps({"frames": fmt, "runner": name, "n_frames": n, "seconds": round(el, 3),
                              "pairs_per_s_incl_startup": round((n - 1) / el, 1)}), flush=True)
        if "online" in poses:
            assert poses["online"].shape == poses["batched"].shape
            print(json.dumps({"frames": fmt, "max_pose_difference_online_vs_batched":
                              float(np.abs(poses["online"] - poses["batched"]).max())}), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Phase log of a short run_kitti_stereo run (LZB_VIO_TIMING=1): where the wall time of 1025 frames goes.
Usage: python tools/e2e_phases.py [n_frames=1025] [batch=256] [fmt=pgm] [repeats=3]"""
import importlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    from PIL import Image
    import test_host_api as tha
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1025
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    fmt = sys.argv[3] if len(sys.argv) > 3 else "pgm"
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    host = os.path.join(entry.PKG_DIR, "host")
    d = tempfile.mkdtemp(prefix="svo_phases_", dir="/dev/shm")
    try:
        seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=torch.device("cuda", 0))
        for cam in (0, 1):
            os.makedirs(os.path.join(d, f"image_{cam}"))
        for t in range(n):
            for cam, im in enumerate(x.cpu().numpy() for x in seq.render(t)):
                path = os.path.join(d, f"image_{cam}", f"{t:06d}.{fmt}")
                if fmt == "pgm":
                    tha._write_pgm(path, im)
                else:
                    Image.fromarray(im).save(path, compress_level=3)
        tha._write_yaml(os.path.join(d, "cfg.yaml"), d, fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy)
        with open(os.path.join(d, "cfg.yaml"), "a", encoding="utf-8") as f:
            f.write(f"batch_size: {batch}\n")
        for r in range(reps):
            t0 = time.perf_counter()
            p = subprocess.run([os.path.join(host, "run_kitti_stereo"), os.path.join(d, "cfg.yaml"), os.path.join(d, "poses.txt")],
                               capture_output=True, env=dict(os.environ, LZB_VIO_TIMING="1", SVO_TIMING="1"))
            el = time.perf_counter() - t0
            print(f"--- run {r}: {el:.3f} s wall, {(n - 1) / el:.0f} pairs/s, rc {p.returncode}")
            print(p.stderr.decode())
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()

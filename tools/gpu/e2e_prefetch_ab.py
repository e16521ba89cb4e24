#!/usr/bin/env python3
"""run_kitti_stereo on an n-frame S0 sequence from PNG and PGM files, best of 3, alternating an environment switch of the
runner -- written for round 6's decode-ahead experiment (LZB_VIO_PREFETCH = 1 / 0 in that build; dropped, DESIGN.md section 6:
the shipped runner ignores the variable, so the script now simply measures the whole-process time four times per format).
usage: e2e_prefetch_ab.py [n=4541]"""
import importlib, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry
import torch
from PIL import Image
import test_host_api as tha
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4541
entry.load_package()
synth = importlib.import_module(entry.PKG_NAME + ".synth")
exe = os.path.join(entry.PKG_DIR, "host", "run_kitti_stereo")
root = tempfile.mkdtemp(prefix="svo_ab_", dir="/dev/shm")
try:
    seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=torch.device("cuda", 0))
    for fmt in ("png", "pgm"):
        for cam in (0, 1):
            os.makedirs(os.path.join(root, fmt, f"image_{cam}"))
    def write(job):
        t, cam, im = job
        tha._write_pgm(os.path.join(root, "pgm", f"image_{cam}", f"{t:06d}.pgm"), im)
        Image.fromarray(im).save(os.path.join(root, "png", f"image_{cam}", f"{t:06d}.png"), compress_level=3)
    with ThreadPoolExecutor(16) as pool:
        for t0 in range(0, n, 128):
            fl, fr = seq.render_range(t0, min(n, t0 + 128))
            fl, fr = fl.cpu().numpy(), fr.cpu().numpy()
            list(pool.map(write, [(t0 + i, cam, a[i]) for i in range(fl.shape[0]) for cam, a in ((0, fl), (1, fr))]))
    for fmt in ("png", "pgm"):
        d = os.path.join(root, fmt)
        tha._write_yaml(os.path.join(d, "cfg.yaml"), d, fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy)
        with open(os.path.join(d, "cfg.yaml"), "a", encoding="utf-8") as f:
            f.write("batch_size: 256\n")
        for pre in ("1", "0", "1", "0"):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                r = subprocess.run([exe, os.path.join(d, "cfg.yaml"), os.path.join(d, "poses.txt")], capture_output=True,
                                   env=dict(os.environ, LZB_VIO_FAST_EXIT="1", LZB_VIO_PREFETCH=pre))
                best = min(best, time.perf_counter() - t0)
                assert r.returncode == 0, r.stderr.decode()[-500:]
            print(f"{fmt} prefetch={pre}: best of 3 {best:.3f} s = {(n - 1) / best:.0f} pairs/s", flush=True)
finally:
    shutil.rmtree(root, ignore_errors=True)

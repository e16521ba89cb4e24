#!/usr/bin/env python3
"""Host time of the calls behind one stream micro-batch (k pairs): where a small stream depth's time goes.
python3 tools/gpu/stream_host_cost.py [k=1] [contexts=1]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def main():
    import torch
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    W, H = 1241, 376
    seq = synth.StereoSequence(width=W, height=H, n_frames=k + 1, seed=20200710, device=torch.device("cuda", 0))
    fr = [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(k + 1)]
    P1, P2 = seq.proj()
    c = pkg.Context(W, H, device=0, P1=P1, P2=P2, max_batch=k)
    c.set_overlap(True)
    pitch = (W + 255) // 256 * 256
    pin = [[c.host_frames(k + 1, pitch) for _ in range(2)] for _ in range(2)]
    for b in range(2):
        for f in range(k + 1):
            pin[b][0][f, :, :W], pin[b][1][f, :, :W] = fr[f]
    T = {n: [] for n in ("copy_frame", "upload", "track_async", "ready_poll", "collect", "wait_upload")}
    for it in range(200):
        b = it & 1
        t0 = time.perf_counter(); pin[b][0][1, :, :W] = fr[1][0]; pin[b][1][1, :, :W] = fr[1][1]; T["copy_frame"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); c.upload_frames(b, pin[b][0], pin[b][1]); T["upload"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); c.track_uploaded_async(b, k + 1); T["track_async"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); c.results_ready(); T["ready_poll"].append(time.perf_counter() - t0)
        if it >= 1:
            t0 = time.perf_counter(); c.collect_results(k); T["collect"].append(time.perf_counter() - t0)
        t0 = time.perf_counter(); c.wait_upload(b); T["wait_upload"].append(time.perf_counter() - t0)
    c.collect_results(k)
    for n, v in T.items():
        v = np.array(v[20:]) * 1e6
        print(f"{n:12s} median {np.median(v):8.1f} us   p90 {np.percentile(v, 90):8.1f} us")
    c.close()


if __name__ == "__main__":
    main()

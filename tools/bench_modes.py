#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md section 6 (bench.py stays the one-line contract):

  online      svo_add_frame latency, one stereo pair per call (SURVEY.md 8d config #2 "online"),
              frames resident in HBM and frames in host memory (H2D inside the call)
  pcie        batched throughput with the frames in PINNED HOST memory: double-buffered H2D on a copy
              stream beside the previous batch's kernels (the PCIe-inclusive rate; never bench.py's value)
  hd          config #4 stand-in: 1920x1080 synthetic stereo stream, FAST threshold raised until
              about 2000 corners per frame survive, batched
  cpu         the oracle on all host cores (OpenMP over points) next to the 1-thread figure

Usage: python tools/bench_modes.py [online] [pcie] [hd] [cpu]   (default: all).  One JSON line per mode.
"""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

W, H, PITCH = 1241, 376, 1280


def render(synth, torch, dev, w, h, pitch, n, seed=20200710):
    seq = synth.StereoSequence(width=w, height=h, n_frames=n, seed=seed, device=dev)
    L = torch.zeros((n, h, pitch), dtype=torch.uint8, device=dev)
    R = torch.zeros((n, h, pitch), dtype=torch.uint8, device=dev)
    for f in range(n):
        l, r = seq.render(f)
        L[f, :, :w] = l
        R[f, :, :w] = r
    return seq, L, R


def main():
    import torch
    modes = sys.argv[1:] or ["online", "pcie", "hd", "cpu"]
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the hot path has no CPU fallback")
    pkg = entry.load_package()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    dev = torch.device("cuda", 0)

    if "online" in modes:
        n = 60
        seq, L, R = render(synth, torch, dev, W, H, PITCH, n)
        P1, P2 = seq.proj()
        for where in ("hbm", "host"):
            ctx = pkg.Context(W, H, device=0, max_batch=1, P1=P1, P2=P2)
            fl = [L[f, :, :W] for f in range(n)] if where == "hbm" else [L[f, :, :W].cpu().numpy().copy() for f in range(n)]
            fr = [R[f, :, :W] for f in range(n)] if where == "hbm" else [R[f, :, :W].cpu().numpy().copy() for f in range(n)]
            lat, ok = [], 0
            for f in range(n):
                t0 = time.perf_counter()
                rc, r = ctx.add_frame(fl[f], fr[f])
                lat.append(time.perf_counter() - t0)
                ok += int(rc == 0)
            lat = np.array(lat[10:]) * 1e3
            print(json.dumps({"mode": "online", "frames": where, "ms_per_pair_median": round(float(np.median(lat)), 3),
                              "ms_per_pair_p90": round(float(np.percentile(lat, 90)), 3),
                              "pairs_per_s": round(1e3 / float(np.mean(lat)), 1), "ok": ok, "n": n}), flush=True)
            ctx.close()
        del L, R

    if "pcie" in modes:
        B = 256
        F = B + 1
        seq, L, R = render(synth, torch, dev, W, H, PITCH, F)
        P1, P2 = seq.proj()
        hostL, hostR = L.cpu().pin_memory(), R.cpu().pin_memory()
        bufs = [(torch.empty_like(L), torch.empty_like(R)) for _ in range(2)]
        del L, R
        ctx = pkg.Context(W, H, device=0, max_batch=B, P1=P1, P2=P2)
        main_s = torch.cuda.current_stream()
        copy_s = torch.cuda.Stream()
        ctx.set_stream(main_s.cuda_stream)
        ctx.set_overlap(True)
        results = torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        ready = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]

        def upload(k):
            with torch.cuda.stream(copy_s):
                copy_s.wait_event(done[k])          # the batch that last used this buffer has finished
                bufs[k][0].copy_(hostL, non_blocking=True)
                bufs[k][1].copy_(hostR, non_blocking=True)
                ready[k].record(copy_s)

        for k in range(2):
            done[k].record(main_s)
        upload(0)
        steps, warm = 8, 2
        t0 = None
        for i in range(steps + warm):
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            k = i & 1
            upload(k ^ 1)                            # next batch's H2D runs beside this batch's kernels
            main_s.wait_event(ready[k])
            ctx.track_batch(bufs[k][0][:, :, :W], bufs[k][1][:, :, :W], results=results)
            done[k].record(main_s)
        ctx.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        res = np.frombuffer(results.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
        print(json.dumps({"mode": "pcie", "pairs_per_s": round(B * steps / el, 1), "ms_per_step": round(1e3 * el / steps, 3),
                          "h2d_bytes_per_step": int(hostL.numel() + hostR.numel()),
                          "h2d_GBps_needed": round((hostL.numel() + hostR.numel()) * steps / el / 1e9, 2),
                          "pairs_ok_last_step": int(res["ok"].sum())}), flush=True)
        ctx.close()
        del bufs, hostL, hostR

    if "hd" in modes:
        w, h, pitch, B = 1920, 1080, 1920, 128
        F = B + 1
        seq, L, R = render(synth, torch, dev, w, h, pitch, F, seed=1)
        P1, P2 = seq.proj()
        probe = pkg.Context(w, h, device=0, max_batch=1, max_keypoints=1 << 16, P1=P1, P2=P2)
        thr, n_kp = 20, None
        for t in range(20, 200, 5):
            n_kp = len(probe.fast_detect(L[0], threshold=t))
            thr = t
            if n_kp <= 2000:
                break
        probe.close()
        ctx = pkg.Context(w, h, device=0, max_batch=B, max_keypoints=4096, fast_threshold=thr, P1=P1, P2=P2)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.set_overlap(True)
        results = torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev)
        for _ in range(2):
            ctx.track_batch(L, R, results=results)
        ctx.sync()
        ctx.enable_timing(True)
        ctx.get_timing()
        torch.cuda.synchronize()
        steps = 5
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.track_batch(L, R, results=results)
        ctx.sync()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        stage = dict(ctx.get_timing())
        res = np.frombuffer(results.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
        pts = int(res["n_prev_kps"].sum())
        lk_ms = stage.get("lk", 0.0)
        print(json.dumps({"mode": "hd", "size": [w, h], "fast_threshold": thr, "keypoints_frame0": n_kp,
                          "mean_keypoints_per_pair": round(pts / B, 1), "pairs_per_s": round(B * steps / el, 1),
                          "ms_per_step": round(1e3 * el / steps, 3), "pairs_ok": int(res["ok"].sum()),
                          "stage_ms_per_step": {k: round(v, 4) for k, v in stage.items()},
                          "lk_algorithmic_GBps": round(pts * 4 * 4257 / (lk_ms * 1e-3) / 1e9, 1) if lk_ms else None}),
              flush=True)
        ctx.close()
        del L, R

    if "cpu" in modes:
        O = entry.load_oracle()
        O.build()
        n = 6
        seq, L, R = render(synth, torch, dev, W, H, PITCH, n + 1)
        P1, P2 = seq.proj()
        fl, fr = L[:, :, :W].cpu().numpy(), R[:, :, :W].cpu().numpy()
        prm = O.make_params(P1, P2)
        for threads in (1, min(16, os.cpu_count())):      # a 1-GPU box is given 16 cores
            kps, pose = O.fast(fl[0]), np.eye(4)
            c0 = time.perf_counter()
            for t in range(1, n + 1):
                _, kps, pose = O.lk_track_step(prm, fl[t - 1], fr[t - 1], fl[t], fr[t], kps, pose, threads=threads)
            c1 = time.perf_counter()
            print(json.dumps({"mode": "cpu", "threads": threads, "host_cpus": os.cpu_count(),
                              "pairs_per_s": round(n / (c1 - c0), 2)}), flush=True)


if __name__ == "__main__":
    main()

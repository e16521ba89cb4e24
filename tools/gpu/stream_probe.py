#!/usr/bin/env python3
"""The bench's online and stream legs alone (S0 frames, LK mode): usage stream_probe.py [depths=1,2,4]"""
import sys, os, json, importlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
synth = importlib.import_module(entry.PKG_NAME + ".synth")
depths = tuple(int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(","))
dev = torch.device("cuda", 0)
n = 200
seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=dev)
fr = [seq.render(f) for f in range(n)]
L = torch.stack([f[0] for f in fr]); R = torch.stack([f[1] for f in fr])
P1, P2 = seq.proj()
octx = pkg.Context(1241, 376, device=0, max_batch=1, P1=P1, P2=P2)
fl = [L[f].cpu().numpy().copy() for f in range(64)]; frr = [R[f].cpu().numpy().copy() for f in range(64)]
lat = []
for f in range(64):
    t0 = time.perf_counter(); octx.add_frame(fl[f], frr[f]); lat.append(time.perf_counter() - t0)
octx.close()
print("online ms/pair median", round(1e3 * float(np.median(lat[8:])), 3))
r = bench.stream_leg(pkg, importlib.import_module(entry.PKG_NAME + ".stream"), L, R, 1241, 376, P1, P2, {}, depths=depths)
for k, v in r["depths"].items():
    print("stream", k, v)

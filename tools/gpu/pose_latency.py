#!/usr/bin/env python3
"""bench.pose_latency_probe on S0 frames: when are batch k's poses there once batch k + 1 is launched?  usage: pose_latency.py [B=48] [accum=sse2|exact|simd128|sse2_legacy]"""
import sys, os, json, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
synth = importlib.import_module(entry.PKG_NAME + ".synth")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
accum = sys.argv[2] if len(sys.argv) > 2 else "sse2"
dev = torch.device("cuda", 0)
seq = synth.StereoSequence(width=1241, height=376, n_frames=B + 1, seed=20200710, device=dev)
fr = [seq.render(f) for f in range(B + 1)]
L = torch.stack([f[0] for f in fr]); R = torch.stack([f[1] for f in fr])
P1, P2 = seq.proj()
acc = {"exact": 0, "sse2": pkg.LK_ACCUM_SSE2, "simd128": pkg.LK_ACCUM_SIMD128, "sse2_legacy": pkg.LK_ACCUM_SSE2_LEGACY}[accum]
print(accum, json.dumps(bench.pose_latency_probe(pkg, L, R, 1241, 376, B, dict(P1=P1, P2=P2, lk_accum=acc), reps=5)))

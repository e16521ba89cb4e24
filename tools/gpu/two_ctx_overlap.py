import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
import __graft_entry__ as entry
import torch
pkg = entry.load_package(); synth = importlib.import_module(entry.PKG_NAME + ".synth")
dev = torch.device("cuda", 0)
W, H, B = 1241, 376, 256
cache = "/tmp/s0_frames_c2.pt"
if os.path.exists(cache):
    L, R = torch.load(cache)
    L, R = L.to(dev), R.to(dev)
else:
    seq = synth.StereoSequence(width=W, height=H, n_frames=2 * B + 1, seed=20200710, device=dev)
    L = torch.zeros((2 * B + 1, H, 1280), dtype=torch.uint8, device=dev); R = torch.zeros_like(L)
    for f in range(2 * B + 1):
        l, r = seq.render(f); L[f, :, :W] = l; R[f, :, :W] = r
seq = synth.StereoSequence(width=W, height=H, n_frames=2, seed=20200710)
P1, P2 = seq.proj()
def run(nctx, steps=12):
    cs = [pkg.Context(W, H, device=0, max_batch=B, P1=P1, P2=P2) for _ in range(nctx)]
    ss = [torch.cuda.Stream() for _ in range(nctx)]
    bufs = [[torch.zeros((B, pkg.STEP_DTYPE.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)] for _ in range(nctx)]
    for c, s in zip(cs, ss):
        c.set_stream(s.cuda_stream); c.set_overlap(True)
    def one(k):
        i = k % nctx; ch = k % 2
        cs[i].track_batch(L[ch * B:ch * B + B + 1, :, :W], R[ch * B:ch * B + B + 1, :, :W], results=bufs[i][(k // nctx) & 1])
    for k in range(4): one(k)
    for c in cs: c.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps): one(k)
    for c in cs: c.sync()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    for c in cs: c.close()
    return B * steps / el, 1e3 * el / steps
for n in (1, 2, 1, 2):
    print(n, "contexts:", run(n))

"""Online (one pair per call) loop for kernel traces: python3 tools/gpu/online_loop.py [lk|orb] [n_frames]"""
import sys, importlib, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as e
pkg = e.load_package(); synth = importlib.import_module(e.PKG_NAME + '.synth')
mode = sys.argv[1] if len(sys.argv) > 1 else "lk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda', 0)
seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=dev)
fr = [seq.render(t) for t in range(n)]
P1, P2 = seq.proj()
kw = dict(P1=P1, P2=P2)
if mode == "orb": kw.update(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=100.0)
c = pkg.Context(1241, 376, device=0, max_batch=1, **kw)
torch.cuda.synchronize()
for l, r in fr:
    rc, res = c.add_frame(l, r)
print("done", int(res['n_tracked']), int(res['ransac_iters']), int(res['lm_iters']))
c.close()

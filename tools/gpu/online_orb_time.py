"""ORB-mode online path: median wall time of svo_add_frame per stereo pair (host frames), result on the host.
usage: python3 tools/gpu/online_orb_time.py [n_frames]"""
import sys, importlib, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import __graft_entry__ as e
pkg = e.load_package(); synth = importlib.import_module(e.PKG_NAME + '.synth')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seq = synth.StereoSequence(width=1241, height=376, n_frames=n, seed=20200710, device=torch.device('cuda', 0))
fr = [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(n)]
P1, P2 = seq.proj()
c = pkg.Context(1241, 376, device=0, max_batch=1, P1=P1, P2=P2, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=100.0)
ts, ok = [], 0
for t, (l, r) in enumerate(fr):
    t0 = time.perf_counter()
    rc, res = c.add_frame(l, r)
    ts.append(time.perf_counter() - t0)
    ok += int(rc == 0)
ts = np.array(ts[4:]) * 1e3
print("orb online ms per pair: median %.3f p90 %.3f (%d frames, %d ok) env=%s" % (np.median(ts), np.percentile(ts, 90), n, ok,
      {k: v for k, v in os.environ.items() if k.startswith("SVO_ORB")}))
c.close()

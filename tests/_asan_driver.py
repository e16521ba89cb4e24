"""Driver run under AddressSanitizer + UBSan (tests/test_oracle_sanitizers.py): every oracle entry
point on synthetic stereo frames and on adversarial textures.  argv[1] = instrumented library."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as e  # noqa: E402
from stress_lk_parity import texture  # noqa: E402

O = e.load_oracle()
e.load_package()
lib = C.CDLL(sys.argv[1])
lib.orc_rng_next.restype = C.c_uint32
O._LIB = lib
synth = importlib.import_module(e.PKG_NAME + ".synth")
rng = np.random.default_rng(0)
seq = synth.StereoSequence(width=320, height=128, n_frames=3, seed=7)
fr = [tuple(x.numpy() for x in seq.render(t)) for t in range(3)]
prm = O.make_params(*seq.proj())
kp, pose = O.fast(fr[0][0]), np.eye(4)
for t in (1, 2):
    r, kp, pose = O.lk_track_step(prm, *fr[t - 1], *fr[t], kp, pose, threads=2)
prm_o = O.make_params(*seq.proj(), min_t2=0.05 ** 2, max_t2=100.0)
feats = [(O.orb_extract(L)[:2], O.orb_extract(R)[:2]) for L, R in fr]
pose = np.eye(4)
for t in (1, 2):
    (kL, dL), (kR, dR) = feats[t - 1]
    (k2, d2), _ = feats[t]
    r, pose = O.orb_track_step(prm_o, kL, dL, kR, dR, k2, d2, pose)
for kind in range(4):
    w, h = int(rng.integers(160, 260)), int(rng.integers(120, 180))
    img = texture(rng, h, w, kind)
    for thr in (1, 20, 200):
        O.fast(img, thr=thr)
        O.fast(img, thr=thr, nms=False)
    O.orb_extract(img, nfeatures=1000, cap=16384)
    J = np.roll(img, (2, -3), (0, 1))
    pts = np.stack([rng.uniform(-15, w + 15, 500), rng.uniform(-15, h + 15, 500)], 1).astype(np.float32)
    O.lk_track(img, J, pts, threads=2)
P1, P2 = [np.asarray(p, np.float64).reshape(3, 4) for p in seq.proj()]
x1 = np.stack([rng.uniform(0, 320, 200), rng.uniform(0, 128, 200)], 1).astype(np.float32)
O.triangulate(P1, P2, x1, x1 - np.float32([3.0, 0.0]))
X = np.stack([rng.uniform(-5, 5, 60), rng.uniform(-2, 2, 60), rng.uniform(5, 20, 60)], 1).astype(np.float32)
u = (P1[:, :3] @ X.T).T
O.pnp_ransac(X, (u[:, :2] / u[:, 2:]).astype(np.float32), P1[:, :3])
O.pnp_ransac(X[:5], (u[:5, :2] / u[:5, 2:]).astype(np.float32), P1[:, :3])
print("sanitizer run complete")

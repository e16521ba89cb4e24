#!/usr/bin/env python3
"""Generates tests/golden/stereo_quad_160x96.npz: a rendered 160x96 stereo quadruple (two
consecutive stereo frames of the synthetic corridor) and the outputs of the CPU oracle on it
(FAST keypoints, circular LK tracks, triangulated points, PnP pose, ORB keypoints/descriptors of the
first left image).  The reference has no fixtures; these vectors were produced by
oracle/ in the build container and pin BOTH the oracle (against accidental change) and the HIP path.
Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

O = entry.load_oracle()
O.build()
entry.load_package()
import importlib  # noqa: E402
synth = importlib.import_module(entry.PKG_NAME + ".synth")

seq = synth.StereoSequence(width=160, height=96, n_frames=2, seed=42, scales=(0.5, 2.0, 8.0), supersample=2)
(L0, R0), (L1, R1) = [tuple(x.numpy() for x in seq.render(t)) for t in range(2)]
P1, P2 = seq.proj()
kp = O.fast(L0)
prm = O.make_params(P1, P2)
res, kp1, pose = O.lk_track_step(prm, L0, R0, L1, R1, kp, np.eye(4), want_tracks=True)
X = O.triangulate(P1, P2, res["tracks"][0], res["tracks"][1])
okp, odesc, oper = O.orb_extract(L0, nlevels=3, nfeatures=300)
np.savez_compressed(os.path.join(os.path.dirname(__file__), "stereo_quad_160x96.npz"),
                    L0=L0, R0=R0, L1=L1, R1=R1, P1=np.array(P1), P2=np.array(P2),
                    fast_kp=kp, fast_kp_next=kp1, tracks=res["tracks"], X=X,
                    n_inliers=res["n_inliers"], rvec=res["rvec"], tvec=res["tvec"], pose=pose,
                    orb_kp=okp, orb_desc=odesc, orb_per_level=oper)
print("keypoints", len(kp), "tracks", res["n_tracked"], "inliers", res["n_inliers"], "orb", len(okp), "ok", res["ok"])

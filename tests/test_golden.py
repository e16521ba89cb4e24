"""Committed golden vectors (tests/golden/stereo_quad_160x96.npz, made by make_golden.py from the CPU
oracle in the build container).  CPU: the oracle still reproduces them bit for bit.  GPU (-m gpu):
the HIP path reproduces them too, without the oracle in the loop."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stereo_quad_160x96.npz"))


def test_oracle_reproduces_golden(oracle):
    kp = oracle.fast(G["L0"])
    assert kp.tobytes() == G["fast_kp"].tobytes()
    prm = oracle.make_params(G["P1"], G["P2"])
    res, kp1, pose = oracle.lk_track_step(prm, G["L0"], G["R0"], G["L1"], G["R1"], kp, np.eye(4), want_tracks=True)
    assert res["ok"] == 1 and kp1.tobytes() == G["fast_kp_next"].tobytes()
    assert res["tracks"].tobytes() == G["tracks"].tobytes()
    assert oracle.triangulate(G["P1"], G["P2"], res["tracks"][0], res["tracks"][1]).tobytes() == G["X"].tobytes()
    assert res["n_inliers"] == int(G["n_inliers"])
    assert np.array_equal(res["rvec"], G["rvec"]) and np.array_equal(res["tvec"], G["tvec"])
    assert np.array_equal(pose, G["pose"])
    okp, odesc, oper = oracle.orb_extract(G["L0"], nlevels=3, nfeatures=300)
    assert okp.tobytes() == G["orb_kp"].tobytes() and odesc.tobytes() == G["orb_desc"].tobytes()


@pytest.mark.gpu
def test_hip_reproduces_golden(pkg):
    h, w = G["L0"].shape
    c = pkg.Context(w, h, device=0, P1=G["P1"], P2=G["P2"])
    assert c.fast_detect(G["L0"]).tobytes() == G["fast_kp"].tobytes()
    for s, k in enumerate(("L0", "R0", "L1", "R1")):
        c.build_pyramid(s, G[k])
    pts = np.stack([G["fast_kp"]["x"], G["fast_kp"]["y"]], 1).astype(np.float32)
    tr = c.circular_match((0, 1, 2, 3), pts)
    assert np.stack(tr).tobytes() == G["tracks"].tobytes()
    assert c.triangulate(G["P1"], G["P2"], tr[0], tr[1]).tobytes() == G["X"].tobytes()
    c.add_frame(G["L0"], G["R0"])
    rc, r = c.add_frame(G["L1"], G["R1"])
    assert rc == 0 and int(r["n_inliers"]) == int(G["n_inliers"])
    assert np.abs(r["tvec"] - G["tvec"]).max() < 1e-9 and np.abs(r["rvec"] - G["rvec"]).max() < 1e-9
    assert np.linalg.norm(c.get_pose() - G["pose"]) / np.linalg.norm(G["pose"]) < 1e-9
    c.close()
    c = pkg.Context(w, h, device=0, track_mode=pkg.MODE_ORB, orb_nlevels=3, orb_nfeatures=300)
    okp, odesc, oper = c.orb_extract(G["L0"])
    assert okp.tobytes() == G["orb_kp"].tobytes() and odesc.tobytes() == G["orb_desc"].tobytes()
    c.close()

"""The OpenCV cross-check through C++ (tools/cv_crosscheck.cpp) for a box that has OpenCV's headers and libraries but no
cv2 module: `make -C tools cv_crosscheck` builds only where pkg-config knows an OpenCV; here it does not, and the test
skips with that reason.  Compares cv::FAST, the four chained cv::calcOpticalFlowPyrLK calls (status bytes in every
accumulation order; coordinates bit for bit in at least one), cv::triangulatePoints (either restated system) and
cv::solvePnPRansac (inlier set; pose under one of the refit forks) with the oracle on the golden quadruple."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import conftest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stereo_quad_160x96.npz"))
TOOLS = os.path.join(conftest.ROOT, "tools")


def _have_opencv():
    if not shutil.which("pkg-config"):
        return False
    return any(subprocess.run(["pkg-config", "--exists", n]).returncode == 0 for n in ("opencv4", "opencv"))


@pytest.mark.skipif(not _have_opencv(), reason="pkg-config knows no OpenCV (opencv4 / opencv) in this environment: nothing to "
                                                "cross-check the oracle against (PARITY UNPINNED)")
def test_oracle_against_the_c_plus_plus_callees(oracle, tmp_path):
    subprocess.check_call(["make", "-C", TOOLS, "cv_crosscheck"])
    for k in ("L0", "R0", "L1", "R1"):
        with open(tmp_path / f"{k}.pgm", "wb") as f:
            f.write(b"P5\n%d %d\n255\n" % (G[k].shape[1], G[k].shape[0]) + G[k].tobytes())
    pts = np.stack([G["fast_kp"]["x"], G["fast_kp"]["y"]], 1).astype(np.float32)
    pts.tofile(tmp_path / "pts.f32")
    np.concatenate([G["P1"], G["P2"]]).astype(np.float64).tofile(tmp_path / "P.f64")
    for k, i in (("x1", 0), ("x2", 1), ("x3", 3)):
        np.ascontiguousarray(G["tracks"][i]).tofile(tmp_path / f"{k}.f32")
    subprocess.check_call([os.path.join(TOOLS, "cv_crosscheck"), str(tmp_path)])
    print("OpenCV", open(tmp_path / "version.txt").read())
    mine = oracle.fast(G["L0"])
    fk = np.fromfile(tmp_path / "fast.f32", np.float32).reshape(-1, 3)
    assert np.array_equal(fk, np.stack([mine["x"], mine["y"], mine["response"]], 1))
    imgs = {"L0": G["L0"], "R0": G["R0"], "L1": G["L1"], "R1": G["R1"]}
    chain = (("L0", "R0"), ("R0", "R1"), ("R1", "L1"), ("L1", "L0"))
    exact_in = {}
    for mode in range(5):
        old = oracle.set_lk_accum(mode)
        try:
            cur, same = pts, True
            for c, (a, b) in enumerate(chain):
                nxt, st = oracle.lk_track(imgs[a], imgs[b], cur)
                want = np.fromfile(tmp_path / f"lk_{c}.f32", np.float32).reshape(-1, 2)
                wst = np.fromfile(tmp_path / f"lk_{c}.u8", np.uint8)
                same = same and np.array_equal(st, wst) and np.array_equal(nxt[wst == 1], want[wst == 1])
                cur = want                                   # follow OpenCV's chain so that one differing call does not hide the next
            exact_in[mode] = same
        finally:
            oracle.set_lk_accum(old)
    print("LK accumulation orders reproducing this OpenCV bit for bit:", [m for m, ok in exact_in.items() if ok])
    assert any(exact_in.values()), exact_in
    X4 = np.fromfile(tmp_path / "X4.f32", np.float32).reshape(4, -1)
    X3 = (X4[:3] / np.where(X4[3] != 0, X4[3], 1)).T
    ok = []
    for v in (0, 1):
        oracle.set_opencv_compat(oracle.COMPAT_TRIANGULATE, v)
        ok.append(np.abs(oracle.triangulate(G["P1"], G["P2"], G["tracks"][0], G["tracks"][1]) - X3).max() <= 1e-6 * np.abs(X3).max())
    oracle.set_opencv_compat(oracle.COMPAT_TRIANGULATE, 0)
    assert any(ok), "neither triangulation system reproduces this OpenCV"
    pose = np.fromfile(tmp_path / "pnp.f64", np.float64)
    inl = np.fromfile(tmp_path / "pnp_inliers.i32", np.int32)
    K = G["P1"].reshape(3, 4)[:, :3].copy()
    best = None
    for v in range(4):
        oracle.set_opencv_compat(oracle.COMPAT_PNP_REFIT, v)
        r = oracle.pnp_ransac(X3.astype(np.float32), G["tracks"][3], K)
        assert np.array_equal(np.flatnonzero(r["mask"]), np.sort(inl))
        e = max(np.abs(r["rvec"] - pose[:3]).max(), np.abs(r["tvec"] - pose[3:]).max())
        best = e if best is None else min(best, e)
    oracle.set_opencv_compat(oracle.COMPAT_PNP_REFIT, 0)
    assert best < 1e-9, best

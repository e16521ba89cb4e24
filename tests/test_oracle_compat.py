"""The oracle's OpenCV-version forks (oracle/geom.c orc_set_opencv_compat), CPU only.

The reference asks for "OpenCV 3" without a patch release (/root/reference/CMakeLists.txt:17) and the callees it uses
changed inside the 3.x line.  Value 0 of every knob is the canonical choice the HIP path is held to; these tests pin what
the OTHER values compute with answers known independently of the oracle: numpy's SVD for the 6 x 4 triangulation system,
planted poses for the refit variants, analytic sub-pixel shifts for the LK lane orders.  None of it is checked against an
OpenCV binary here (there is none in this image): tests/test_cv_crosscheck.py does that wherever cv2 can be imported."""
import os

import numpy as np
import pytest

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "stereo_quad_160x96.npz"))


@pytest.fixture()
def compat(oracle):
    """set(knob, value) with every knob back at 0 afterwards (the switches are process-wide)."""
    touched = []

    def set_(knob, value):
        touched.append(knob)
        oracle.set_opencv_compat(knob, value)
    yield set_
    for k in touched:
        oracle.set_opencv_compat(k, 0)


def _K(P1):
    return np.asarray(P1, np.float64).reshape(3, 4)[:, :3].copy()


def test_knobs_default_to_the_canonical_choice(oracle):
    for k in (oracle.COMPAT_TRIANGULATE, oracle.COMPAT_PNP_REFIT, oracle.COMPAT_PNP_MINIMAL, oracle.COMPAT_LK_LANES):
        assert oracle.lib().orc_get_opencv_compat(k) == 0


def test_triangulate_6x4_is_the_null_vector_of_the_six_row_system(oracle, compat):
    """C8 = 1: rows x P[2] - P[0], y P[2] - P[1], x P[1] - y P[0] per view; the answer is the right singular vector of
    the smallest singular value -- checked against LAPACK on the golden tracks (whose rays do not meet exactly)."""
    P = [np.asarray(G[k], np.float64).reshape(3, 4) for k in ("P1", "P2")]
    x1, x2 = G["tracks"][0], G["tracks"][1]
    compat(oracle.COMPAT_TRIANGULATE, 1)
    X = oracle.triangulate(G["P1"], G["P2"], x1, x2)
    for i in range(0, len(x1), 7):
        A = []
        for Pj, (x, y) in zip(P, (x1[i].astype(np.float64), x2[i].astype(np.float64))):
            A += [x * Pj[2] - Pj[0], y * Pj[2] - Pj[1], x * Pj[1] - y * Pj[0]]
        v = np.linalg.svd(np.array(A))[2][-1]
        want = v[:3] / v[3]
        assert np.abs(X[i] - want).max() <= 2e-5 * np.abs(want).max(), (i, X[i], want)
    # the canonical 4 x 4 system is a different least-squares problem: same points only when the rays meet
    compat(oracle.COMPAT_TRIANGULATE, 0)
    assert oracle.triangulate(G["P1"], G["P2"], x1, x2).tobytes() == G["X"].tobytes()


def test_triangulation_forks_agree_on_consistent_projections(oracle, compat):
    rng = np.random.default_rng(3)
    Xw = np.c_[rng.uniform(-8, 8, 64), rng.uniform(-2, 2, 64), rng.uniform(6, 40, 64)]
    P = [np.asarray(G[k], np.float64).reshape(3, 4) for k in ("P1", "P2")]
    uv = []
    for Pj in P:
        h = (Pj @ np.c_[Xw, np.ones(64)].T).T
        uv.append((h[:, :2] / h[:, 2:]).astype(np.float32))
    got = {}
    for v in (0, 1):
        compat(oracle.COMPAT_TRIANGULATE, v)
        got[v] = oracle.triangulate(G["P1"], G["P2"], uv[0], uv[1])
        # float32 image points: the depth of a 0.54 m rig is known to ~1e-3 relative at these ranges
        assert np.abs(got[v] - Xw).max(axis=1).max() <= 2e-2 * np.abs(Xw).max()
    assert np.abs(got[0] - got[1]).max() <= 2e-2 * np.abs(Xw).max()


def _planted(n=120, seed=5, noise=0.05, outliers=12):
    rng = np.random.default_rng(seed)
    K = _K(G["P1"])
    Xw = np.c_[rng.uniform(-8, 8, n), rng.uniform(-2, 2, n), rng.uniform(6, 40, n)].astype(np.float32)
    rv = np.array([0.01, -0.02, 0.005]); t = np.array([0.03, -0.01, -0.9])
    th = np.linalg.norm(rv); k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    h = (K @ (R @ Xw.T.astype(np.float64) + t[:, None])).T
    uv = (h[:, :2] / h[:, 2:] + rng.normal(0, noise, (n, 2))).astype(np.float32)
    uv[:outliers] += rng.uniform(5, 30, (outliers, 2)).astype(np.float32)
    return Xw, uv, K, R, t


def test_pnp_refit_forks_reach_the_same_minimum(oracle, compat):
    """C9: the refit from the best hypothesis (0), from the last evaluated one (2: 3.4) and from zeros (3) minimise the
    same reprojection error over the same inlier set; without a refit (1: before 3.3) the answer is the winning EPnP."""
    Xw, uv, K, R, t = _planted()
    ref = oracle.pnp_ransac(Xw, uv, K)
    assert ref["ok"] and ref["n_inliers"] >= 100 and np.abs(ref["tvec"] - t).max() < 5e-3
    for v in (2, 3):
        compat(oracle.COMPAT_PNP_REFIT, v)
        r = oracle.pnp_ransac(Xw, uv, K)
        assert r["mask"].tobytes() == ref["mask"].tobytes() and r["ransac_iters"] == ref["ransac_iters"]
        assert np.abs(r["tvec"] - ref["tvec"]).max() < 1e-6 and np.abs(r["rvec"] - ref["rvec"]).max() < 1e-7, v
    compat(oracle.COMPAT_PNP_REFIT, 1)
    r = oracle.pnp_ransac(Xw, uv, K)
    assert r["lm_iters"] == 0 and r["mask"].tobytes() == ref["mask"].tobytes()
    assert 1e-6 < np.abs(r["tvec"] - ref["tvec"]).max() < 0.2        # a five-point model, not the inlier optimum


def test_pnp_five_points_direct_return(oracle, compat):
    """C10: with exactly five points 3.4 returns the kernel's EPnP pose and calls every point an inlier; the canonical
    path refines that pose by LM.  Both see the same five inliers."""
    Xw, uv, K, R, t = _planted(n=5, noise=0.0, outliers=0)
    a = oracle.pnp_ransac(Xw, uv, K)
    compat(oracle.COMPAT_PNP_MINIMAL, 1)
    b = oracle.pnp_ransac(Xw, uv, K)
    assert a["ok"] and b["ok"] and a["n_inliers"] == b["n_inliers"] == 5 and b["lm_iters"] == 0 and a["lm_iters"] >= 1
    assert np.abs(b["tvec"] - t).max() < 1e-2 and np.abs(a["tvec"] - t).max() < 1e-2


@pytest.mark.parametrize("mode", [3, 4])
def test_lk_lane_orders_track_like_the_exact_sums(oracle, compat, mode):
    """C11: the two upstream SIMD blocks restated whole (3 = legacy CV_SSE2, 4 = CV_SIMD128) differ from the exact sums
    only in the rounding of the five window sums: same status bytes, coordinates within 1e-3 px, most of them identical."""
    kp = G["fast_kp"]
    pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    ref, st_ref = oracle.lk_track(G["L0"], G["R0"], pts)
    compat(oracle.COMPAT_LK_LANES, mode)
    out, st = oracle.lk_track(G["L0"], G["R0"], pts)
    assert (st == st_ref).mean() > 0.995
    both = (st == 1) & (st_ref == 1)
    assert np.abs(out[both] - ref[both]).max() < 1e-3
    same = (out[both] == ref[both]).all(axis=1).mean()
    assert 0.5 < same < 1.0, same                       # float order matters in the last bits, and only there

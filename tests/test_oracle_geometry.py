"""Known-answer tests pinning the oracle's geometry restatement (CPU only): Jacobi SVD vs LAPACK,
triangulation of projected known points, Rodrigues vs scipy + finite differences, the OpenCV MWC
RNG recurrence, EPnP / RANSAC-PnP recovering planted poses, motion gates."""
import numpy as np
from scipy.spatial.transform import Rotation

K = np.array([[718.856, 0, 607.193], [0, 718.856, 185.216], [0, 0, 1.0]])
P1 = np.hstack([K, np.zeros((3, 1))])
P2 = np.hstack([K, K @ np.array([[-0.537], [0], [0]])])


def _project(P, X):
    x = (P @ np.hstack([X, np.ones((len(X), 1))]).T).T
    return x[:, :2] / x[:, 2:3]


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 1.6, n), rng.uniform(5, 40, n)], 1)


def test_jacobi_svd_matches_lapack(oracle):
    rng = np.random.default_rng(0)
    for (m, n) in [(3, 3), (4, 4), (6, 4), (6, 5), (12, 12), (6, 6)]:
        A = rng.normal(size=(m, n))
        if m == n == 12:
            A = A.T @ A                                     # symmetric PSD like EPnP's M^T M
        W, U, Vt = oracle.jacobi_svd(A)
        assert np.allclose(W, np.linalg.svd(A, compute_uv=False), rtol=1e-11, atol=1e-12)
        assert (np.diff(W) <= 1e-12).all()
        assert np.allclose(U @ np.diag(W) @ Vt, A, atol=1e-10)
        assert np.allclose(Vt @ Vt.T, np.eye(n), atol=1e-12)
        assert np.allclose(U.T @ U, np.eye(n), atol=1e-12)


def test_triangulate_known_points(oracle):
    X = _scene(200, 1)
    x1 = _project(P1, X).astype(np.float32)
    x2 = _project(P2, X).astype(np.float32)
    got, got4 = oracle.triangulate(P1, P2, x1, x2, want4=True)
    # float32 pixel coordinates limit depth accuracy: relative error ~ Z * 1e-5 / disparity
    assert np.abs(got - X).max() < 0.05
    assert got4.shape == (4, 200)
    assert np.allclose(got4[:3] / got4[3], got.T, rtol=1e-5)
    # rectified rig: depth == fx * b / disparity
    d = x1[:, 0] - x2[:, 0]
    assert np.allclose(got[:, 2], 718.856 * 0.537 / d, rtol=2e-3)


def test_rodrigues_roundtrip_and_jacobian(oracle):
    rng = np.random.default_rng(2)
    for _ in range(20):
        r = rng.normal(size=3) * rng.uniform(0.001, 1.5)
        R, J = oracle.rodrigues_vec2mat(r, jac=True)
        assert np.allclose(R, Rotation.from_rotvec(r).as_matrix(), atol=1e-13)
        rb = oracle.rodrigues_mat2vec(R)                 # |r| may exceed pi: compare as rotations
        assert np.allclose(Rotation.from_rotvec(rb).as_matrix(), R, atol=1e-10)
        if np.linalg.norm(r) < 3.0:
            assert np.allclose(rb, r, atol=1e-10)
        eps = 1e-6
        for i in range(3):
            dr = np.zeros(3)
            dr[i] = eps
            num = (oracle.rodrigues_vec2mat(r + dr) - oracle.rodrigues_vec2mat(r - dr)) / (2 * eps)
            assert np.allclose(J[i], num.reshape(9), atol=1e-8)
    R0, J0 = oracle.rodrigues_vec2mat(np.zeros(3), jac=True)
    assert np.array_equal(R0, np.eye(3))
    assert J0[0, 5] == -1 and J0[0, 7] == 1 and J0[1, 2] == 1 and J0[1, 6] == -1
    assert np.allclose(oracle.rodrigues_mat2vec(np.eye(3)), 0)
    # rotation by pi about an axis (the s < 1e-5, c < 0 branch)
    ax = np.array([0.6, -0.64, 0.48])
    Rpi = Rotation.from_rotvec(ax / np.linalg.norm(ax) * np.pi).as_matrix()
    rp = oracle.rodrigues_mat2vec(Rpi)
    assert np.allclose(Rotation.from_rotvec(rp).as_matrix(), Rpi, atol=1e-9)


def test_rng_is_opencv_mwc(oracle):
    """cv::RNG: state = (uint32)state * 4164903690 + (state >> 32), output = low 32 bits."""
    st = 0xFFFFFFFFFFFFFFFF
    exp = []
    for _ in range(16):
        st = ((st & 0xFFFFFFFF) * 4164903690 + (st >> 32)) & 0xFFFFFFFFFFFFFFFF
        exp.append(st & 0xFFFFFFFF)
    assert oracle.rng_sequence(16) == exp
    assert exp[0] == (0xFFFFFFFF * 4164903690 + 0xFFFFFFFF) & 0xFFFFFFFF


def _pose(seed):
    rng = np.random.default_rng(seed)
    r = rng.normal(size=3) * 0.03
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-1.2, -0.6)])
    return r, t


def test_epnp_recovers_planted_pose(oracle):
    for n, seed in [(5, 1), (6, 2), (12, 3), (40, 4)]:
        X = _scene(n, seed)
        r, t = _pose(seed)
        R = Rotation.from_rotvec(r).as_matrix()
        us = _project(np.hstack([K @ R, (K @ t)[:, None]]), X)
        Re, te = oracle.epnp(X, us, K[0, 0], K[1, 1], K[0, 2], K[1, 2])
        assert np.allclose(Re @ Re.T, np.eye(3), atol=1e-9) and np.linalg.det(Re) > 0
        if n >= 6:                                       # 5 points: exact only up to EPnP's N<=3 betas
            assert np.allclose(Re, R, atol=1e-5) and np.allclose(te, t, atol=1e-4)
        else:
            assert np.abs(_project(np.hstack([K @ Re, (K @ te)[:, None]]), X) - us).max() < 0.5


def test_pnp_ransac_planted_pose_with_outliers(oracle):
    n = 400
    X = _scene(n, 11)
    r, t = _pose(11)
    R = Rotation.from_rotvec(r).as_matrix()
    rng = np.random.default_rng(5)
    x = _project(np.hstack([K @ R, (K @ t)[:, None]]), X) + rng.normal(scale=0.05, size=(n, 2))
    out = rng.choice(n, 100, replace=False)
    x[out] += rng.uniform(5, 40, (100, 2)) * rng.choice([-1, 1], (100, 2))
    res = oracle.pnp_ransac(X.astype(np.float32), x.astype(np.float32), K)
    assert res["ok"] == 1
    assert res["mask"][out].sum() == 0 and res["n_inliers"] >= 280
    assert res["n_inliers"] == res["mask"].sum()
    assert np.allclose(res["rvec"], r, atol=2e-4) and np.allclose(res["tvec"], t, atol=3e-3)
    assert np.allclose(res["R"], Rotation.from_rotvec(res["rvec"]).as_matrix(), atol=1e-12)
    # adaptive stop: with 75 % inliers far fewer than 500 hypotheses are needed
    assert 1 <= res["ransac_iters"] < 100 and 0 <= res["best_iter"] < res["ransac_iters"]
    assert 1 <= res["lm_iters"] <= 20


def test_pnp_ransac_edge_cases(oracle):
    X = _scene(5, 3)
    r, t = _pose(3)
    R = Rotation.from_rotvec(r).as_matrix()
    x = _project(np.hstack([K @ R, (K @ t)[:, None]]), X)
    # exactly 5 points: one model, everything an inlier
    res = oracle.pnp_ransac(X.astype(np.float32), x.astype(np.float32), K)
    assert res["ok"] == 1 and res["n_inliers"] == 5 and res["ransac_iters"] == 1
    # exactly 4 points: solvePnPRansac's kernel is P3P, one model from all four, all inliers, LM refit on them
    res = oracle.pnp_ransac(X[:4].astype(np.float32), x[:4].astype(np.float32), K)
    assert res["ok"] == 1 and res["n_inliers"] == 4 and res["ransac_iters"] == 1 and res["mask"].tolist() == [1, 1, 1, 1]
    assert np.abs(res["tvec"] - t).max() < 1e-3 and np.abs(res["rvec"] - r).max() < 1e-4
    # fewer than 4 points: cv::solvePnPRansac would assert -> no solution
    res = oracle.pnp_ransac(X[:3].astype(np.float32), x[:3].astype(np.float32), K)
    assert res["ok"] == 0 and res["n_inliers"] == 0 and np.array_equal(res["R"], np.eye(3))
    # pure garbage: no hypothesis reaches 5 inliers -> failure, zero inliers
    rng = np.random.default_rng(0)
    Xg = _scene(60, 9).astype(np.float32)
    xg = rng.uniform(0, 1200, (60, 2)).astype(np.float32)
    res = oracle.pnp_ransac(Xg, xg, K, iterations=50)
    assert res["ransac_iters"] == 50
    assert (res["ok"] == 0 and res["n_inliers"] == 0) or res["n_inliers"] >= 5


def test_gate_and_accumulate(oracle):
    r, t = np.array([0.01, -0.02, 0.005]), np.array([0.02, -0.01, -0.9])
    R = Rotation.from_rotvec(r).as_matrix()
    rc, pose, Ti = oracle.gate_and_accumulate(R, t, np.eye(4))
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    assert rc == 1 and np.allclose(Ti, np.linalg.inv(T), atol=1e-14) and np.allclose(pose, Ti)
    rc2, pose2, _ = oracle.gate_and_accumulate(R, t, pose)
    assert rc2 == 1 and np.allclose(pose2, Ti @ Ti, atol=1e-13)
    # rotation gate: any |euler| >= 0.1 rad
    Rbig = Rotation.from_euler("y", 0.12).as_matrix()
    assert oracle.gate_and_accumulate(Rbig, t, np.eye(4))[0] == -4
    # translation gates (LK mode: 0.0005^2 < |t|^2 < 100), pose untouched on failure
    rc3, pose3, _ = oracle.gate_and_accumulate(R, np.array([0, 0, 1e-4]), pose)
    assert rc3 == -5 and np.array_equal(pose3, pose)
    assert oracle.gate_and_accumulate(R, np.array([0, 0, 10.0]), np.eye(4))[0] == -5


def test_full_step_recovers_synthetic_motion(oracle, small_seq):
    seq, frames = small_seq
    prm = oracle.make_params(*seq.proj())
    pose = np.eye(4)
    kps = oracle.fast(frames[0][0])
    for t in range(1, 4):
        res, kps_next, pose = oracle.lk_track_step(prm, *frames[t - 1], *frames[t], kps, pose)
        assert res["ok"] == 1 and res["fail_stage"] == 0
        assert res["n_tracked"] >= 30 and res["n_inliers"] >= 0.3 * res["n_tracked"]
        gt = seq.relative_gt(t).numpy()
        assert np.abs(res["tvec"] - gt[:3, 3]).max() < 0.08
        assert np.abs(Rotation.from_matrix(res["R"]).as_rotvec()
                      - Rotation.from_matrix(gt[:3, :3]).as_rotvec()).max() < 5e-3
        kps = kps_next
    gt_pose = (np.linalg.inv(seq.poses_wc()[0].numpy()) @ seq.poses_wc()[3].numpy())
    assert np.abs(pose[:3, 3] - gt_pose[:3, 3]).max() < 0.2


def test_quartic_solver_against_numpy_roots(oracle):
    """polynom_solver.cpp's solve_deg4 (behind P3P): the REAL roots of random quartics, against numpy.roots."""
    import ctypes as C
    rng = np.random.default_rng(0)
    lib = oracle.lib()
    for it in range(600):
        roots = rng.uniform(-3, 3, 4)
        co = (np.poly(roots) if it % 3 else np.poly([roots[0], roots[1], 1 + 2j, 1 - 2j]).real) * rng.uniform(0.5, 2)
        x = (C.c_double * 4)()
        n = lib.orc_solve_deg4(*[C.c_double(v) for v in co], x)
        want = np.sort([r.real for r in np.roots(co) if abs(r.imag) < 1e-9])
        got = np.sort(list(x)[:n])
        assert len(want) == len(got) and (not len(want) or np.abs(want - got).max() <= 1e-6 * max(1.0, np.abs(want).max())), (it, co)


def test_p3p_recovers_planted_poses(oracle):
    """cv::solvePnP(SOLVEPNP_P3P) as restated in oracle/p3p.c: four exact projections -> the planted pose.  Gao's closed form
    is not a numerically gentle method (nor is upstream's): most solves are at 1e-9, a few per cent at 1e-3 -- the RANSAC path
    follows it with an LM refit on the four points."""
    import ctypes as C
    rng = np.random.default_rng(0)
    lib = oracle.lib()
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    errs = []
    for it in range(400):
        Xw = np.c_[rng.uniform(-8, 8, 4), rng.uniform(-2, 2, 4), rng.uniform(6, 40, 4)]
        rv, tt = rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3)
        R = Rotation.from_rotvec(rv).as_matrix()
        Xc = (R @ Xw.T).T + tt
        us = np.c_[fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy]
        Ro, to = (C.c_double * 9)(), (C.c_double * 3)()
        ok = lib.orc_p3p4((C.c_double * 12)(*Xw.reshape(-1)), (C.c_double * 8)(*us.reshape(-1)), C.c_double(fx), C.c_double(fy),
                          C.c_double(cx), C.c_double(cy), Ro, to)
        assert ok == 1
        errs.append(max(np.abs(np.array(Ro).reshape(3, 3) - R).max(), np.abs(np.array(to) - tt).max()))
    errs = np.array(errs)
    assert np.median(errs) < 1e-8 and (errs < 1e-6).mean() > 0.85 and (errs < 1e-2).mean() > 0.97, (np.median(errs), (errs < 1e-6).mean())

"""GPU parity tests (run with -m gpu): triangulation, RANSAC-EPnP + LM pose solver, the fused online
step and the batched step, HIP through the C-ABI vs the CPU oracle.

Bars: triangulated points, RANSAC winner / iteration counts / inlier masks: BIT-EXACT (IEEE f64
+,-,*,/,sqrt in the same order on both sides).  Pose after the LM refit: relative Frobenius error
<= 1e-4 (BASELINE.json north_star); the refit's reductions run in a different order on the GPU and
use device sin/cos/acos, so a few ulp are expected -- the tests also assert the much tighter 1e-9
that is actually observed."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation

pytestmark = pytest.mark.gpu

K = np.array([[718.856, 0, 607.193], [0, 718.856, 185.216], [0, 0, 1.0]])
P1 = np.hstack([K, np.zeros((3, 1))])
P2 = np.hstack([K, K @ np.array([[-0.537], [0], [0]])])
POSE_TOL = 1e-4          # north_star: pose within 1e-4 relative Frobenius
TIGHT = 1e-9


def relfro(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _project(P, X):
    x = (P @ np.hstack([X, np.ones((len(X), 1))]).T).T
    return x[:, :2] / x[:, 2:3]


def _scene(n, seed):
    rng = np.random.default_rng(seed)
    return np.stack([rng.uniform(-8, 8, n), rng.uniform(-2, 1.6, n), rng.uniform(5, 40, n)], 1)


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def ctx(pkg, tc):
    c = pkg.Context(416, 128, device=0)
    yield c
    c.close()


def test_triangulate_parity_bit_exact(ctx, oracle, tc):
    rng = np.random.default_rng(0)
    X = _scene(3000, 1)
    x1 = (_project(P1, X) + rng.normal(scale=0.3, size=(3000, 2))).astype(np.float32)
    x2 = (_project(P2, X) + rng.normal(scale=0.3, size=(3000, 2))).astype(np.float32)
    x2[:5] = x1[:5]                                   # zero disparity: w ~ 0 corner case
    ref = oracle.triangulate(P1, P2, x1, x2)
    got = ctx.triangulate(P1, P2, x1, x2)
    assert got.tobytes() == ref.tobytes()
    got_d = ctx.triangulate(P1, P2, tc.from_numpy(x1).cuda(), tc.from_numpy(x2).cuda())
    assert got_d.cpu().numpy().tobytes() == ref.tobytes()
    assert ctx.triangulate(P1, P2, x1[:0], x2[:0]).shape == (0, 3)


def _planted(n, seed, n_out, noise=0.05):
    X = _scene(n, seed)
    rng = np.random.default_rng(seed + 100)
    r = rng.normal(size=3) * 0.03
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.05, 0.05), rng.uniform(-1.2, -0.6)])
    R = Rotation.from_rotvec(r).as_matrix()
    x = _project(np.hstack([K @ R, (K @ t)[:, None]]), X) + rng.normal(scale=noise, size=(n, 2))
    out = rng.choice(n, n_out, replace=False)
    x[out] += rng.uniform(5, 40, (n_out, 2)) * rng.choice([-1, 1], (n_out, 2))
    return X.astype(np.float32), x.astype(np.float32), r, t


def _check_pnp(got, ref):
    assert got["ok"] == ref["ok"]
    assert got["ransac_iters"] == ref["ransac_iters"] and got["best_iter"] == ref["best_iter"]
    assert got["n_inliers"] == ref["n_inliers"]
    assert np.array_equal(got["mask"], ref["mask"])
    if ref["ok"]:
        assert got["lm_iters"] == ref["lm_iters"]
        T_g = np.hstack([got["R"], got["tvec"][:, None]])
        T_r = np.hstack([ref["R"], ref["tvec"][:, None]])
        assert relfro(T_g, T_r) <= POSE_TOL
        assert relfro(T_g, T_r) <= TIGHT
        assert np.abs(got["rvec"] - ref["rvec"]).max() <= TIGHT


@pytest.mark.parametrize("n,seed,n_out", [(400, 11, 100), (1500, 12, 300), (64, 13, 30), (200, 14, 150),
                                          (9, 15, 0), (5, 16, 0)])
def test_pnp_ransac_parity(ctx, oracle, tc, n, seed, n_out):
    X, x, r, t = _planted(n, seed, n_out)
    ref = oracle.pnp_ransac(X, x, K)
    got = ctx.pnp_ransac(X, x, K)
    _check_pnp(got, ref)
    got_d = ctx.pnp_ransac(tc.from_numpy(X).cuda(), tc.from_numpy(x).cuda(), K)
    _check_pnp(got_d, ref)
    if n >= 64 and n_out < n // 2:
        assert ref["ok"] == 1 and np.abs(ref["tvec"] - t).max() < 0.02


def test_pnp_refit_svd_route_parity(ctx, oracle, monkeypatch):
    """The refit solves its normal equations by Cholesky and keeps the reference's SVD solve (run by four
    waves) for rank-deficient systems; SVO_REFIT_SVD=1 sends ordinary data down that route."""
    monkeypatch.setenv("SVO_REFIT_SVD", "1")
    for n, seed, n_out in [(400, 11, 100), (64, 13, 30), (9, 15, 0)]:
        X, x, r, t = _planted(n, seed, n_out)
        _check_pnp(ctx.pnp_ransac(X, x, K), oracle.pnp_ransac(X, x, K))


def test_pnp_ransac_many_rounds_and_failure(ctx, oracle, tc):
    """Low inlier ratio -> the adaptive stop needs several 64-hypothesis rounds; garbage -> failure."""
    X, x, r, t = _planted(300, 21, 225, noise=0.02)          # 25 % inliers
    ref = oracle.pnp_ransac(X, x, K)
    got = ctx.pnp_ransac(X, x, K)
    assert ref["ransac_iters"] > 64
    _check_pnp(got, ref)
    rng = np.random.default_rng(0)
    Xg = _scene(60, 9).astype(np.float32)
    xg = rng.uniform(0, 1200, (60, 2)).astype(np.float32)
    _check_pnp(ctx.pnp_ransac(Xg, xg, K, iterations=130), oracle.pnp_ransac(Xg, xg, K, iterations=130))
    # fewer than 4 points: no solution (cv::solvePnPRansac would assert), identity rotation, zero inliers
    res = ctx.pnp_ransac(X[:3], x[:3], K)
    assert res["ok"] == 0 and res["n_inliers"] == 0 and np.array_equal(res["R"], np.eye(3))


def test_pnp_ransac_four_points_take_the_p3p_kernel(ctx, oracle):
    """npoints == 4: cv::solvePnPRansac's minimal solver is P3P (round 5; reachable with num_features_tracking = 4, reference
    src/tracking.cpp:274, 485): one model from the four points, every point an inlier, LM refit on them.  Discrete fields equal
    the oracle's; the pose to 1e-6 (pow / acos / cos of the device library seed the refit with other last bits, and four points
    leave the refit's minimum shallow)."""
    n_ok = 0
    for seed in range(24):
        X, x, r, t = _planted(4, 100 + seed, 0, noise=0.0 if seed % 2 else 0.05)
        ref = oracle.pnp_ransac(X, x, K)
        got = ctx.pnp_ransac(X, x, K)
        assert got["ok"] == ref["ok"] and got["n_inliers"] == ref["n_inliers"] and got["ransac_iters"] == ref["ransac_iters"], seed
        assert np.array_equal(got["mask"], ref["mask"]), seed
        if ref["ok"]:
            n_ok += 1
            assert got["n_inliers"] == 4 and abs(got["lm_iters"] - ref["lm_iters"]) <= 1, seed
            assert np.abs(got["tvec"] - ref["tvec"]).max() <= 1e-6 * max(1.0, np.abs(ref["tvec"]).max()), seed
            assert np.abs(got["rvec"] - ref["rvec"]).max() <= 1e-6, seed
    assert n_ok >= 20


def test_num_features_tracking_four_is_accepted_three_refused(pkg):
    c = pkg.Context(416, 128, device=0, num_features_tracking=4)
    c.close()
    with pytest.raises(pkg.SvoError):
        pkg.Context(416, 128, device=0, num_features_tracking=3)


@pytest.mark.parametrize("n,n_out,iterations", [(6, 3, 500), (7, 4, 500), (12, 9, 500), (40, 33, 500), (300, 255, 500),
                                                (300, 255, 130), (25, 21, 700)])
def test_pnp_ransac_orb_mode_first_phase_is_the_whole_search(pkg, oracle, tc, n, n_out, iterations):
    """An ORB-mode context draws the subsets of the whole iterationsCount (<= 512) up front, the 64 lanes together, from
    the constant cv::RNG(-1) stream; with a handful of points nearly every subset redraws indices and the 4096-draw
    stretch runs out (6 points: ~8.7 draws per subset), so the serial generator takes over mid-phase; 700 iterations
    need a second phase drawn by it.  Few inliers: the search never stops early.  Everything must equal the oracle's
    serial loop: iteration count, winner, mask, pose."""
    c = pkg.Context(416, 128, device=0, track_mode=pkg.MODE_ORB, min_move2=0.0, max_move2=1e9)
    X, x, r, t = _planted(n, 40 + n, n_out, noise=0.02)
    ref = oracle.pnp_ransac(X, x, K, iterations=iterations)
    got = c.pnp_ransac(X, x, K, iterations=iterations)
    assert ref["ransac_iters"] > min(iterations, 64) or n <= 7
    _check_pnp(got, ref)
    c.close()


def _check_step(g, r, first=False):
    assert int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"]
    assert int(g["n_cur_kps"]) == r["n_cur_kps"]
    if first:
        return
    assert int(g["n_prev_kps"]) == r["n_prev_kps"]
    assert int(g["n_tracked"]) == r["n_tracked"]
    if r["fail_stage"] in (0, 3, 4, 5):
        assert int(g["n_inliers"]) == r["n_inliers"]
        Tg = np.hstack([g["R"].reshape(3, 3), g["tvec"][:, None]])
        Tr = np.hstack([r["R"], r["tvec"][:, None]])
        assert relfro(Tg, Tr) <= POSE_TOL and relfro(Tg, Tr) <= TIGHT
    if r["ok"]:
        assert relfro(g["T_rel_inv"].reshape(4, 4), r["T_rel_inv"]) <= TIGHT


def _oracle_sequence(oracle, seq, frames):
    prm = oracle.make_params(*seq.proj())
    pose = np.eye(4)
    kps = oracle.fast(frames[0][0])
    out = []
    for t in range(1, len(frames)):
        res, kps, pose = oracle.lk_track_step(prm, *frames[t - 1], *frames[t], kps, pose)
        out.append((res, pose.copy()))
    return out


def test_online_add_frame_parity(pkg, oracle, tc, small_seq):
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s)
    ref = _oracle_sequence(oracle, seq, frames)
    rc, g0 = c.add_frame(*frames[0])
    assert rc == 0 and g0["ok"] == 1 and g0["n_cur_kps"] == len(oracle.fast(frames[0][0]))
    for t in range(1, len(frames)):
        # alternate host / device inputs
        fr = frames[t] if t % 2 else tuple(tc.from_numpy(x).cuda() for x in frames[t])
        rc, g = c.add_frame(*fr)
        r, pose = ref[t - 1]
        assert rc == (0 if r["ok"] else r["fail_stage"])
        _check_step(g, r)
        assert relfro(g["pose"].reshape(4, 4), pose) <= POSE_TOL
        assert relfro(c.get_pose(), pose) <= TIGHT
    # reset -> INITING again
    c.reset()
    assert np.array_equal(c.get_pose(), np.eye(4))
    rc, g = c.add_frame(*frames[1])
    assert g["n_prev_kps"] == 0 and g["ok"] == 1
    c.close()


def test_track_batch_parity(pkg, oracle, tc, small_seq):
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    F = len(frames)
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, max_batch=F - 1)
    pitch = 512
    L = tc.zeros((F, h, pitch), dtype=tc.uint8, device="cuda")
    R = tc.zeros((F, h, pitch), dtype=tc.uint8, device="cuda")
    for f in range(F):
        L[f, :, :w] = tc.from_numpy(frames[f][0]).cuda()
        R[f, :, :w] = tc.from_numpy(frames[f][1]).cuda()
    res = c.track_batch(L[:, :, :w], R[:, :, :w])
    ref = _oracle_sequence(oracle, seq, frames)
    assert len(res) == F - 1
    for p in range(F - 1):
        _check_step(res[p], ref[p][0])
        assert relfro(res[p]["pose"].reshape(4, 4), ref[p][1]) <= TIGHT
    # a seeded initial pose and device-resident results
    pose0 = np.eye(4)
    pose0[:3, 3] = [1.0, 2.0, 3.0]
    dres = tc.zeros((F - 1, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
    c.track_batch(L[:, :, :w], R[:, :, :w], pose0=pose0, results=dres)
    c.sync()
    res2 = np.frombuffer(dres.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    assert relfro(res2[-1]["pose"].reshape(4, 4), pose0 @ ref[-1][1]) <= TIGHT
    c.close()


def test_track_batch_overlap_mode(pkg, oracle, tc, small_seq):
    """Overlap mode (pose stage of batch k beside the front end of batch k+1) changes no result."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    F = len(frames)
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, max_batch=F - 1)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    ref = c.track_batch(L, R)                                     # host results, stream order
    c.set_overlap(True)
    outs = [tc.zeros((F - 1, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda") for _ in range(3)]
    for o in outs:                                                # back-to-back batches, no host sync between
        c.track_batch(L, R, results=o)
    c.sync()
    for o in outs:
        got = np.frombuffer(o.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
        assert got.tobytes() == ref.tobytes()
    # stage calls and the online path wait for a pending pose stage by themselves
    c.track_batch(L, R, results=outs[0])
    X = np.random.default_rng(0).uniform(1, 5, (50, 3)).astype(np.float32)
    x = (X[:, :2] / X[:, 2:] * 100 + 200).astype(np.float32)
    assert c.triangulate(P1, P2, x, x + np.float32([5, 0])).shape == (50, 3)
    c.wait_results()
    got = np.frombuffer(outs[0].cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    assert got.tobytes() == ref.tobytes()
    c.set_overlap(False)
    c.close()


def test_failure_stages_parity(pkg, oracle, tc, small_seq):
    """Frames that make the reference return false: too few corners, static scene (|t| gate)."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    prm = oracle.make_params(P1s, P2s)
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s)
    flat = np.full((h, w), 90, np.uint8)
    # (a) current frame has < 30 corners -> stage 1; the next step then tracks from 0 features -> stage 2
    c.add_frame(*frames[0])
    rc, g = c.add_frame(flat, flat)
    kps = oracle.fast(frames[0][0])
    r, kps2, pose = oracle.lk_track_step(prm, *frames[0], flat, flat, kps, np.eye(4))
    assert r["fail_stage"] == 1 and rc == 1
    _check_step(g, r)
    rc, g = c.add_frame(*frames[1])
    r, kps3, pose = oracle.lk_track_step(prm, flat, flat, *frames[1], kps2, pose)
    assert r["fail_stage"] == 2 and rc == 2
    _check_step(g, r)
    # (b) identical consecutive frames: (near-)zero motion; whatever the oracle decides, the GPU agrees
    rc, g = c.add_frame(*frames[1])
    r, kps4, pose2 = oracle.lk_track_step(prm, *frames[1], *frames[1], kps3, pose)
    assert rc == (0 if r["ok"] else r["fail_stage"])
    _check_step(g, r)
    assert relfro(c.get_pose(), pose2) <= TIGHT
    c.close()
    # (c) translation gate (stage 5) and rotation gate (stage 4) with tightened windows
    for kw, stage in [(dict(min_move2=4.0, max_move2=100.0), 5), (dict(min_move2=0.0, max_move2=1e-3), 5)]:
        c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, **kw)
        prm2 = oracle.make_params(P1s, P2s, min_t2=kw["min_move2"], max_t2=kw["max_move2"])
        c.add_frame(*frames[0])
        rc, g = c.add_frame(*frames[1])
        r, _, pose3 = oracle.lk_track_step(prm2, *frames[0], *frames[1], oracle.fast(frames[0][0]), np.eye(4))
        assert r["fail_stage"] == stage and rc == stage
        _check_step(g, r)
        assert np.array_equal(c.get_pose(), np.eye(4)) and np.array_equal(pose3, np.eye(4))
        c.close()
    # (d) inlier-ratio failure (stage 3) with an impossible rate
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, inlier_rate=1.01)
    prm3 = oracle.make_params(P1s, P2s, inlier_rate=1.01)
    c.add_frame(*frames[0])
    rc, g = c.add_frame(*frames[1])
    r, _, _ = oracle.lk_track_step(prm3, *frames[0], *frames[1], oracle.fast(frames[0][0]), np.eye(4))
    assert r["fail_stage"] == 3 and rc == 3
    _check_step(g, r)
    c.close()


def test_host_frame_batches_match_track_batch(pkg, oracle, tc, small_seq):
    """svo_upload_frames + svo_track_uploaded (page-locked host frames, copy stream) == svo_track_batch
    on HBM-resident frames, for both device buffers and for a padded host pitch."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_batch=8, P1=P1, P2=P2)
    L = tc.from_numpy(np.stack([f[0] for f in frames])).cuda()
    R = tc.from_numpy(np.stack([f[1] for f in frames])).cuda()
    ref = c.track_batch(L, R)
    for buf, pitch in ((0, (w + 255) // 256 * 256), (1, w + 3)):
        hl, hr = c.host_frames(len(frames), pitch), c.host_frames(len(frames), pitch)
        for i, (l, r) in enumerate(frames):
            hl[i, :, :w] = l
            hr[i, :, :w] = r
        c.upload_frames(buf, hl, hr)
        got = c.track_uploaded(buf, len(frames))
        c.wait_upload(buf)
        c.host_free(hl)
        c.host_free(hr)
        assert got.tobytes() == ref.tobytes()
    with pytest.raises(pkg.SvoError):
        c.track_uploaded(0, len(frames) + 1)
    c.close()


def test_set_pose_seeds_the_online_chain(pkg, tc, small_seq):
    """svo_set_pose: a context created in the middle of a run (e.g. rebuilt for another frame size by the
    host mirror) continues frame_pose_ instead of restarting from the identity."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    a = pkg.Context(w, h, device=0, P1=P1, P2=P2)
    for f in frames[:3]:
        a.add_frame(*f)
    mid = a.get_pose().copy()
    rc, ra = a.add_frame(*frames[3])
    b = pkg.Context(w, h, device=0, P1=P1, P2=P2)
    b.set_pose(mid)
    rc0, r0 = b.add_frame(*frames[2])                      # initialises only: the pose is the seed
    assert rc0 == 0 and np.array_equal(b.get_pose(), mid) and np.array_equal(r0["pose"].reshape(4, 4), mid)
    rcb, rb = b.add_frame(*frames[3])
    assert rcb == rc and rb["pose"].tobytes() == ra["pose"].tobytes() and np.array_equal(b.get_pose(), a.get_pose())
    b.reset()
    assert np.array_equal(b.get_pose(), np.eye(4))
    a.close()
    b.close()


@pytest.mark.parametrize("overlap", [False, True])
def test_async_uploaded_chunks_continue_the_chain_on_the_device(pkg, tc, small_seq, overlap):
    """svo_track_uploaded_async with two batches outstanding and continue_chain: three one-pair chunks
    (one-frame halo) launched ahead of their predecessors' records give, record for record, what ONE
    svo_track_batch over the four frames gives -- poses included (the seed of chunk c is chunk c-1's last
    pose, read on the device)."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_batch=4, P1=P1, P2=P2)
    L = tc.from_numpy(np.stack([f[0] for f in frames])).cuda()
    R = tc.from_numpy(np.stack([f[1] for f in frames])).cuda()
    pose0 = np.eye(4)
    pose0[:3, 3] = [4.0, -1.0, 2.0]
    whole = c.track_batch(L, R, pose0=pose0)
    c.set_overlap(overlap)
    pitch = (w + 255) // 256 * 256
    chunks = []
    for k in range(3):
        hl, hr = c.host_frames(2, pitch), c.host_frames(2, pitch)
        for i in range(2):
            hl[i, :, :w] = frames[k + i][0]
            hr[i, :, :w] = frames[k + i][1]
        chunks.append((hl, hr))
    got = []
    c.upload_frames(0, *chunks[0])
    c.track_uploaded_async(0, 2, pose0=pose0)
    for k in (1, 2):
        c.upload_frames(k & 1, *chunks[k])
        c.track_uploaded_async(k & 1, 2, continue_chain=True)     # launched before chunk k-1 was collected
        got.append(c.collect_results(1))
    got.append(c.collect_results(1))
    with pytest.raises(pkg.SvoError):
        c.collect_results(1)                                       # nothing outstanding any more
    got = np.concatenate(got)
    assert got.tobytes() == whole.tobytes()
    # a third launch without a collect is refused, not silently queued over live results
    c.upload_frames(0, *chunks[0])
    c.track_uploaded_async(0, 2)
    c.upload_frames(1, *chunks[1])
    c.track_uploaded_async(1, 2, continue_chain=True)
    with pytest.raises(pkg.SvoError):
        c.track_uploaded_async(0, 2, continue_chain=True)
    c.collect_results(1)
    c.collect_results(1)
    for hl, hr in chunks:
        c.host_free(hl)
        c.host_free(hr)
    c.close()


def test_pair_sharding_chains_to_the_whole_sequence(pkg, tc, small_seq):
    """Chunks of frame pairs tracked independently (one-frame halo) + svo_chain_relative over the
    gathered relative motions == the whole sequence tracked in one batch."""
    import importlib
    import conftest
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_batch=8, P1=P1, P2=P2)
    L = tc.from_numpy(np.stack([f[0] for f in frames])).cuda()
    R = tc.from_numpy(np.stack([f[1] for f in frames])).cuda()
    whole = c.track_batch(L, R)
    for world in (2, 3):
        Ts, oks = [], []
        for rank in range(world):
            first, nf = mg.shard_pairs(len(frames), world, rank)
            if nf < 2:
                continue
            r = c.track_batch(L[first:first + nf], R[first:first + nf])
            Ts.append(r["T_rel_inv"])
            oks.append(r["ok"])
        T, ok = np.concatenate(Ts), np.concatenate(oks).astype(np.int32)
        assert T.tobytes() == whole["T_rel_inv"].tobytes() and np.array_equal(ok, whole["ok"])
        got = c.chain_relative(T, ok)
        assert got.tobytes() == whole["pose"].reshape(-1, 16).tobytes()          # same association order
        got_d = c.chain_relative(tc.from_numpy(T).cuda(), tc.from_numpy(ok).cuda())
        c.sync()                                             # device operands: stream-ordered, not host-synchronous
        assert got_d.cpu().numpy().tobytes() == got.tobytes()
        ref = mg.chain_relative(tc.from_numpy(T), tc.from_numpy(ok)).numpy().reshape(-1, 16)
        assert np.abs(ref - got).max() < 1e-12
    # failed pairs are skipped, pose0 seeds the chain
    ok2 = np.array([1, 0, 1], np.int32)
    p0 = np.eye(4); p0[0, 3] = 5.0
    got = c.chain_relative(whole["T_rel_inv"], ok2, pose0=p0).reshape(-1, 4, 4)
    ref = p0 @ whole["T_rel_inv"][0].reshape(4, 4)
    assert np.abs(got[0] - ref).max() < 1e-12 and np.array_equal(got[1], got[0])
    assert np.abs(got[2] - ref @ whole["T_rel_inv"][2].reshape(4, 4)).max() < 1e-12
    c.close()


def test_capacity_overflow_fails_the_pair_loudly(pkg, tc, small_seq):
    """cv::FAST is uncapped; a frame with more corners than max_keypoints must not be tracked as a
    truncated set: the pair reports SVO_FAIL_CAPACITY (6), ok = 0, and the pose chain skips it."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_keypoints=64, P1=P1, P2=P2)
    c.add_frame(*frames[0])
    rc, r = c.add_frame(*frames[1])
    assert rc == 6 and int(r["ok"]) == 0 and int(r["fail_stage"]) == 6 and int(r["n_prev_kps"]) > 64
    assert np.array_equal(c.get_pose(), np.eye(4))
    c.close()


@pytest.mark.parametrize("mode", ["lk", "orb"])
def test_ragged_batch_equals_online_sequence(pkg, tc, small_seq, mode):
    """A batch with a featureless frame and a repeated frame in the middle (0 keypoints, zero motion,
    very different keypoint counts per pair) gives record-for-record what the online path gives
    frame by frame (itself checked against the oracle above), in both track modes; failed pairs are
    skipped by the pose chain."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    flat = np.full((h, w), 90, np.uint8)
    fs = [frames[0], frames[1], (flat, flat), frames[2], frames[2], frames[3]]
    kw = dict(P1=P1, P2=P2)
    if mode == "orb":
        kw.update(track_mode=pkg.MODE_ORB, orb_nlevels=4, orb_nfeatures=600, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    c = pkg.Context(w, h, device=0, max_batch=8, **kw)
    online = []
    for l, r in fs:
        rc, rec = c.add_frame(l, r)
        online.append((rc, rec, c.get_pose().copy()))
    c.reset()
    L = tc.from_numpy(np.stack([f[0] for f in fs])).cuda()
    R = tc.from_numpy(np.stack([f[1] for f in fs])).cuda()
    got = c.track_batch(L, R)
    assert len(got) == len(fs) - 1
    stages = []
    for i, g in enumerate(got):
        rc, rec, pose = online[i + 1]
        for k in ("ok", "fail_stage", "n_prev_kps", "n_cur_kps", "n_tracked", "n_inliers"):
            assert int(g[k]) == int(rec[k]), (i, k)
        assert np.abs(g["pose"].reshape(4, 4) - pose).max() < 1e-9
        assert rc == (0 if int(rec["ok"]) else int(rec["fail_stage"]))
        stages.append(int(g["fail_stage"]))
    assert stages[1] != 0 and stages[2] != 0            # into and out of the featureless frame
    assert int(got[0]["ok"]) == 1 and int(got[4]["ok"]) == 1
    assert np.array_equal(got[2]["pose"], got[0]["pose"])         # the chain skipped the two failures
    c.close()


def test_new_entry_points_reject_bad_arguments(pkg, tc, small_seq):
    """Hard errors (SVO_ERR_ARG) instead of silent truncation or undefined behaviour."""
    import ctypes as C
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_batch=2, P1=P1, P2=P2)
    with pytest.raises(pkg.SvoError):
        c.frame_keypoints(0)                                   # no frame added yet
    c.add_frame(*frames[0])
    with pytest.raises(pkg.SvoError):
        c.frame_keypoints(1)                                   # LK mode detects on the left image only
    with pytest.raises(pkg.SvoError):
        c.frame_keypoints(0, cap=8)                            # capacity too small: error, not truncation
    assert len(c.last_tracks()[0]) == 0
    c.add_frame(*frames[1])
    with pytest.raises(pkg.SvoError):
        c.last_tracks(cap=4)
    hl = c.host_frames(4)
    with pytest.raises(pkg.SvoError):
        c.upload_frames(0, hl, hl)                             # 4 frames > max_batch + 1
    with pytest.raises(pkg.SvoError):
        c.upload_frames(2, hl[:2], hl[:2])                     # buffer index
    with pytest.raises(pkg.SvoError):
        c.track_uploaded(1, 2)                                 # nothing uploaded into buffer 1
    c.host_free(hl)
    assert c.lib.svo_chain_relative(c.h, None, None, 3, None, None, pkg.MEM_HOST) < 0
    assert c.chain_relative(np.zeros((0, 16)), np.zeros(0, np.int32)).shape == (0, 16)
    c.close()
    with pytest.raises(pkg.SvoError):
        pkg.Context(w, h, device=0, track_mode=pkg.MODE_ORB, max_keypoints=1 << 15).orb_extract(frames[0][0])
    with pytest.raises(pkg.SvoError):
        pkg.Context(w, h, device=0, track_mode=pkg.MODE_ORB, orb_nfeatures=40000, orb_nlevels=1).orb_extract(frames[0][0])


def test_fused_step_with_exactly_four_tracks_runs_p3p(pkg, oracle, small_seq):
    """The whole frame step down the P3P branch: num_features_tracking = 4 and a feature_match_error chosen so that EXACTLY
    four point chains survive deleteBadmatchFeatures (the reference then calls cv::solvePnPRansac with four points,
    src/tracking.cpp:274, 485).  Counts, failure stage and inlier mask as the oracle's; pose to 1e-6."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    imgs = [*frames[0], *frames[1]]
    kps = oracle.fast(imgs[0])
    pts = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    pyr = [oracle.PyramidHandle(im) for im in imgs]
    chain, cur, outs, sts = ((0, 1), (1, 3), (3, 2), (2, 0)), pts, [pts], []          # L1 -> R1 -> R2 -> L2 -> L1'
    for a, b in chain:
        nxt, st = oracle.lk_track(pyr[a], pyr[b], cur)
        outs.append(nxt); sts.append(st); cur = nxt
    ok = np.all([s == 1 for s in sts], axis=0) & np.all([(o >= 0).all(axis=1) for o in outs], axis=0)
    dy = np.maximum(np.abs(outs[0][:, 1] - outs[1][:, 1]), np.abs(outs[2][:, 1] - outs[3][:, 1])).astype(np.float64)
    cand = np.sort(dy[ok])
    assert len(cand) > 8 and cand[3] < cand[4]
    err = 0.5 * (cand[3] + cand[4])                          # exactly four chains pass |y0 - y1|, |y2 - y3| <= err
    prm = oracle.make_params(P1s, P2s, feature_match_error=err, num_features_tracking=4)
    r, _, pose = oracle.lk_track_step(prm, *imgs, kps, np.eye(4), want_tracks=True)
    assert r["n_tracked"] == 4
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, feature_match_error=err, num_features_tracking=4)
    c.add_frame(*frames[0])
    rc, g = c.add_frame(*frames[1])
    assert rc == (0 if r["ok"] else r["fail_stage"])
    for k in ("ok", "fail_stage", "n_prev_kps", "n_cur_kps", "n_tracked", "n_inliers", "ransac_iters"):
        assert int(g[k]) == int(r[k]) if k in r else True, k
    assert int(g["n_tracked"]) == 4
    t1l, t1r, t2r, t2l, inl = c.last_tracks()
    assert np.stack([t1l, t1r, t2r, t2l]).tobytes() == r["tracks"].tobytes()
    if r["ok"]:
        assert inl.tolist() == [1, 1, 1, 1]
        assert np.abs(g["tvec"] - r["tvec"]).max() <= 1e-6 and np.abs(g["rvec"] - r["rvec"]).max() <= 1e-6
        assert relfro(c.get_pose(), pose) <= 1e-6
    c.close()

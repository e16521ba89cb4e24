"""Size-independent properties of the HIP path at BASELINE.json's full frame size (1241x376, and one
1920x1080 case), checked WITHOUT the oracle in the loop: sortedness, idempotence, round trips,
stability of the compaction, batch == sequence of online steps, chunk-chaining == whole-sequence."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W, H = 1241, 376


@pytest.fixture(scope="module")
def tc():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


@pytest.fixture(scope="module")
def kitti_frames(synth, tc):
    seq = synth.StereoSequence(width=W, height=H, n_frames=5, seed=20200710, device=tc.device("cuda", 0))
    fr = [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(5)]
    return seq, fr


def test_fast_output_is_sorted_thresholded_and_locally_maximal(pkg, tc, kitti_frames):
    seq, fr = kitti_frames
    c = pkg.Context(W, H, device=0, max_keypoints=1 << 17)        # room for the un-suppressed corner set
    img = fr[0][0]
    kp = c.fast_detect(img)
    assert len(kp) > 1000
    x, y, r = kp["x"].astype(int), kp["y"].astype(int), kp["response"]
    order = y.astype(np.int64) * W + x
    assert np.all(np.diff(order) > 0)                              # cv::FAST order: row-major, no duplicates
    assert x.min() >= 3 and x.max() < W - 3 and y.min() >= 3 and y.max() < H - 3
    assert r.min() >= 20 and np.all(r == np.round(r))             # score = largest passing threshold >= thr
    # strict 3x3 non-max suppression: no two kept keypoints are 8-neighbours
    score = np.zeros((H, W), np.float32)
    score[y, x] = r
    no_nms = c.fast_detect(img, nonmax=False, cap=1 << 17)
    full = np.zeros((H, W), np.float32)
    full[no_nms["y"].astype(int), no_nms["x"].astype(int)] = 1
    assert np.all(full[y, x] == 1)                                 # NMS output is a subset of the corners
    kept = score > 0
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dx or dy:                                           # a strict maximum has no kept 8-neighbour
                assert not np.any(kept & np.roll(kept, (dy, dx), (0, 1)))
    # the detector is a pure function of the image: same answer for host and device inputs, twice
    assert c.fast_detect(tc.from_numpy(img).cuda()).tobytes() == kp.tobytes() == c.fast_detect(img).tobytes()
    c.close()


def test_pyramid_levels_halve_and_constant_images_stay_constant(pkg, tc):
    c = pkg.Context(W, H, device=0)
    img = np.full((H, W), 137, np.uint8)
    c.build_pyramid(0, img)
    w, h = W, H
    for l in range(c.num_levels):
        lvl = c.read_pyramid_level(0, l)
        assert lvl.shape == (h, w) and np.all(lvl == 137)
        w, h = (w + 1) // 2, (h + 1) // 2
    c.close()


def test_lk_self_tracking_is_the_identity(pkg, tc, kitti_frames):
    """Tracking an image against itself returns the input points (first iteration has zero mismatch)."""
    seq, fr = kitti_frames
    c = pkg.Context(W, H, device=0)
    img = fr[0][0]
    c.build_pyramid(0, img)
    c.build_pyramid(1, img)
    kp = c.fast_detect(img)
    pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32) + np.float32(0.25)
    out, st = c.lk_track(0, 1, pts)
    ok = st.astype(bool)
    assert ok.mean() > 0.95
    assert np.abs(out[ok] - pts[ok]).max() < 1e-3
    c.close()


def test_circular_match_is_a_stable_subset_with_consistent_geometry(pkg, tc, kitti_frames):
    seq, fr = kitti_frames
    c = pkg.Context(W, H, device=0)
    for s, im in enumerate((fr[0][0], fr[0][1], fr[1][0], fr[1][1])):
        c.build_pyramid(s, im)
    kp = c.fast_detect(fr[0][0])
    pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    t1l, t1r, t2r, t2l = c.circular_match((0, 1, 2, 3), pts)
    m = len(t1l)
    assert 500 < m <= len(pts)
    # stable compaction: the kept t1_left points are a subsequence of the input, in order
    idx = np.flatnonzero((pts[:, None, :] == t1l[None, :, :]).all(2).any(1))
    assert len(idx) == m and np.array_equal(pts[idx], t1l)
    # the filter's own predicate holds for everything that was kept
    assert np.all(np.abs(t1l[:, 1] - t1r[:, 1]) <= 3.0) and np.all(np.abs(t2r[:, 1] - t2l[:, 1]) <= 3.0)
    assert (t1l >= 0).all() and (t1r >= 0).all() and (t2r >= 0).all() and (t2l >= 0).all()
    assert np.median(t1l[:, 0] - t1r[:, 0]) > 0                    # positive disparity on a rectified rig
    c.close()


def test_triangulation_and_pnp_round_trip(pkg, tc, kitti_frames):
    seq, fr = kitti_frames
    P1, P2 = [np.asarray(p, np.float64).reshape(3, 4) for p in seq.proj()]
    c = pkg.Context(W, H, device=0, P1=P1, P2=P2)
    rng = np.random.default_rng(3)
    X = np.stack([rng.uniform(-8, 8, 3000), rng.uniform(-2, 2, 3000), rng.uniform(6, 40, 3000)], 1)
    Xh = np.concatenate([X, np.ones((len(X), 1))], 1)
    x1 = (P1 @ Xh.T).T
    x2 = (P2 @ Xh.T).T
    x1, x2 = (x1[:, :2] / x1[:, 2:]).astype(np.float32), (x2[:, :2] / x2[:, 2:]).astype(np.float32)
    Xr = c.triangulate(P1, P2, x1, x2)
    assert np.abs(Xr - X).max() / np.abs(X).max() < 2e-3          # float32 pixel coordinates in, float32 points out
    # a planted motion comes back from solvePnPRansac on exact projections
    ang = 0.02
    Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t = np.array([0.05, -0.02, -0.9])
    Xc = (Rm @ X.T).T + t
    u = (P1[:, :3] @ Xc.T).T
    u = (u[:, :2] / u[:, 2:]).astype(np.float32)
    r = c.pnp_ransac(X.astype(np.float32), u, P1[:, :3])
    assert r["n_inliers"] > 0.9 * len(X)
    assert np.abs(np.asarray(r["tvec"]) - t).max() < 2e-3 and np.abs(np.asarray(r["R"]).reshape(3, 3) - Rm).max() < 1e-4
    c.close()


@pytest.mark.parametrize("mode", ["lk", "orb"])
def test_batch_equals_online_and_chunks_chain_to_the_whole(pkg, tc, kitti_frames, synth, mode):
    import importlib
    import conftest
    mg = importlib.import_module(conftest.entry.PKG_NAME + ".multigpu")
    seq, fr = kitti_frames
    P1, P2 = seq.proj()
    kw = dict(P1=P1, P2=P2)
    if mode == "orb":
        kw.update(track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    c = pkg.Context(W, H, device=0, max_batch=4, **kw)
    L = tc.from_numpy(np.stack([f[0] for f in fr])).cuda()
    R = tc.from_numpy(np.stack([f[1] for f in fr])).cuda()
    whole = c.track_batch(L, R)
    assert whole["ok"].all() and (whole["n_inliers"] > 5).all()
    c.reset()
    for t, (l, r) in enumerate(fr):                                 # the same frames one by one
        rc, rec = c.add_frame(l, r)
        if t:
            assert rc == 0 and int(rec["n_inliers"]) == int(whole["n_inliers"][t - 1])
            assert np.abs(c.get_pose() - whole["pose"][t - 1].reshape(4, 4)).max() < 1e-9
    # two chunks with a one-frame halo, chained from their relative motions
    Ts, oks = [], []
    for rank in range(2):
        first, nf = mg.shard_pairs(len(fr), 2, rank)
        r = c.track_batch(L[first:first + nf], R[first:first + nf])
        Ts.append(r["T_rel_inv"]); oks.append(r["ok"])
    got = c.chain_relative(np.concatenate(Ts), np.concatenate(oks).astype(np.int32))
    assert got.tobytes() == whole["pose"].reshape(-1, 16).tobytes()
    # ground truth: one metre forward per frame on this sequence, recovered within centimetres
    gt = np.linalg.inv(seq.poses_wc()[0].numpy()) @ seq.poses_wc()[len(fr) - 1].numpy()
    assert np.abs(whole["pose"][-1].reshape(4, 4)[:3, 3] - gt[:3, 3]).max() < 0.1
    c.close()


def test_orb_and_matcher_properties(pkg, tc, kitti_frames):
    seq, fr = kitti_frames
    c = pkg.Context(W, H, device=0, track_mode=pkg.MODE_ORB)
    kps, desc, per = c.orb_extract(fr[0][0])
    assert 1900 <= len(kps) <= 2100 and per[:8].sum() == len(kps)
    assert np.all(np.diff(kps["octave"]) >= 0)                     # levels are emitted in order
    assert (kps["angle"] >= 0).all() and (kps["angle"] < 360).all()
    assert (kps["x"] >= 16).all() and (kps["y"] >= 16).all()       # EDGE_THRESHOLD - 3 border at every level
    for l in range(8):                                              # level l size = round(size / 1.2^l)
        lvl = c.orb_read_level(l)
        assert abs(lvl.shape[1] - W / 1.2 ** l) < 1.0 and abs(lvl.shape[0] - H / 1.2 ** l) < 1.0
    # matching a descriptor set against itself: every row finds itself (or an identical earlier row) at distance 0
    idx, dist = c.match_hamming(desc, desc)
    assert np.all(dist == 0) and np.all(idx <= np.arange(len(desc)))
    assert np.array_equal(desc[idx], desc)
    # extraction is deterministic and independent of where the image lives
    k2, d2, _ = c.orb_extract(tc.from_numpy(fr[0][0]).cuda())
    assert k2.tobytes() == kps.tobytes() and d2.tobytes() == desc.tobytes()
    c.close()


def test_hd_frame_properties(pkg, tc, synth):
    """BASELINE config #4 frame size: the same invariants at 1920x1080."""
    w, h = 1920, 1080
    seq = synth.StereoSequence(width=w, height=h, n_frames=2, seed=1, device=tc.device("cuda", 0))
    fr = [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(2)]
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, max_keypoints=1 << 15, P1=P1, P2=P2)
    kp = c.fast_detect(fr[0][0])
    order = kp["y"].astype(np.int64) * w + kp["x"].astype(np.int64)
    assert len(kp) > 2000 and np.all(np.diff(order) > 0)
    c.add_frame(*fr[0])
    rc, rec = c.add_frame(*fr[1])
    assert rc == 0 and int(rec["n_prev_kps"]) == len(kp) and int(rec["n_tracked"]) > 1000
    gt = seq.relative_gt(1).numpy()
    assert np.abs(rec["tvec"] - gt[:3, 3]).max() < 0.05
    c.close()

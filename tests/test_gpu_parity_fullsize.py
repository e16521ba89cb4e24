"""Whole frame steps, HIP vs the CPU oracle, at BASELINE.json's FULL frame sizes (-m gpu):
  * config #2: 1241x376 S0 frames, track_mode LK_stereof2f_pnp -- online (svo_add_frame) and batched
    (svo_track_batch), three consecutive pairs;
  * config #3: the same frames, track_mode ORB_stereof2f_pnp (nFeatures 2000, 8 levels);
  * config #4: 1920x1080, about 2000 corners per frame.
Every stage decision is compared: FAST / ORB keypoint counts, the matched tracks (bit-exact), the
triangulated points (bit-exact), RANSAC's winning hypothesis index, iteration count after the
adaptive stop and inlier mask (bit-exact: at ~2500 tracks the solver runs several 64-hypothesis
rounds), LM iteration count, fail_stage, and the pose within 1e-4 relative Frobenius (north_star),
with the bound actually observed (1e-9) asserted as well.
Reference call sites: src/tracking.cpp:258-344 (LK step), :168-249 (ORB step)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
POSE_TOL, TIGHT = 1e-4, 1e-9


def relfro(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def _render(synth, tc, w, h, n, seed):
    seq = synth.StereoSequence(width=w, height=h, n_frames=n, seed=seed, device=tc.device("cuda", 0))
    return seq, [tuple(x.cpu().numpy() for x in seq.render(t)) for t in range(n)]


@pytest.fixture(scope="module")
def kitti(synth, tc):
    return _render(synth, tc, 1241, 376, 4, 20200710)


def _K(P1):
    return np.asarray(P1, np.float64).reshape(3, 4)[:, :3].copy()


def _oracle_lk_steps(oracle, seq, frames, **prm_kw):
    """Per pair: (step result incl. tracks, 3-D points, RANSAC record incl. mask, pose after the step)."""
    P1, P2 = seq.proj()
    prm = oracle.make_params(P1, P2, **prm_kw)
    kps = oracle.fast(frames[0][0], thr=prm.fast_thr)
    pose = np.eye(4)
    out = []
    for t in range(1, len(frames)):
        res, kps, pose = oracle.lk_track_step(prm, *frames[t - 1], *frames[t], kps, pose, want_tracks=True, threads=8)
        X = oracle.triangulate(P1, P2, res["tracks"][0], res["tracks"][1])
        pnp = oracle.pnp_ransac(X, res["tracks"][3], _K(P1), iterations=prm.iterations, reproj_err=prm.reproj_err,
                                confidence=prm.confidence)
        assert pnp["n_inliers"] == res["n_inliers"]            # the step is exactly this composition
        out.append((res, X, pnp, pose.copy()))
    return out


def _check_record(g, r, pnp):
    assert int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"]
    assert int(g["n_prev_kps"]) == r["n_prev_kps"] and int(g["n_cur_kps"]) == r["n_cur_kps"]
    assert int(g["n_tracked"]) == r["n_tracked"] and int(g["n_inliers"]) == r["n_inliers"]
    assert int(g["ransac_iters"]) == pnp["ransac_iters"] and int(g["lm_iters"]) == pnp["lm_iters"]
    Tg = np.hstack([g["R"].reshape(3, 3), g["tvec"][:, None]])
    Tr = np.hstack([r["R"], r["tvec"][:, None]])
    assert relfro(Tg, Tr) <= POSE_TOL and relfro(Tg, Tr) <= TIGHT
    if r["ok"]:
        assert relfro(g["T_rel_inv"].reshape(4, 4), r["T_rel_inv"]) <= TIGHT


def _check_online_lk(pkg, oracle, seq, frames, ref, **ctx_kw):
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, **ctx_kw)
    rc, g0 = c.add_frame(*frames[0])
    assert rc == 0
    for t in range(1, len(frames)):
        r, X, pnp, pose = ref[t - 1]
        rc, g = c.add_frame(*frames[t])
        assert rc == (0 if r["ok"] else r["fail_stage"])
        _check_record(g, r, pnp)
        t1l, t1r, t2r, t2l, inl = c.last_tracks()
        for got, want in zip((t1l, t1r, t2r, t2l), r["tracks"]):
            assert got.tobytes() == want.tobytes()                       # matched tracks: bit-exact
        assert inl.tobytes() == pnp["mask"].tobytes()                    # RANSAC inlier mask: bit-exact
        # the solver stage on the same tracks: winner index and 3-D points
        Xg = c.triangulate(P1, P2, t1l, t1r)
        assert Xg.tobytes() == X.tobytes()
        sg = c.pnp_ransac(Xg, t2l, _K(P1), iterations=c.cfg.iterations, reproj_err=c.cfg.reproj_err,
                          confidence=c.cfg.confidence)
        assert sg["best_iter"] == pnp["best_iter"] and sg["ransac_iters"] == pnp["ransac_iters"]
        assert np.array_equal(sg["mask"], pnp["mask"])
        assert relfro(c.get_pose(), pose) <= POSE_TOL and relfro(c.get_pose(), pose) <= TIGHT
    c.close()


def test_lk_whole_steps_kitti_size_online_and_batched(pkg, oracle, tc, kitti):
    seq, frames = kitti
    ref = _oracle_lk_steps(oracle, seq, frames)
    assert all(r["ok"] for r, _, _, _ in ref)
    assert min(r["n_tracked"] for r, _, _, _ in ref) > 1000          # thousands of tracks per pair
    assert max(p["ransac_iters"] for _, _, p, _ in ref) >= 1
    _check_online_lk(pkg, oracle, seq, frames, ref)
    # svo_track_batch on the same frames, stream order and overlap mode
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=len(frames) - 1)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    for p, (r, X, pnp, pose) in enumerate(ref):
        _check_record(res[p], r, pnp)
        assert relfro(res[p]["pose"].reshape(4, 4), pose) <= TIGHT
    c.set_overlap(True)
    dres = tc.zeros((len(ref), pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
    c.track_batch(L, R, results=dres)
    c.sync()
    assert np.frombuffer(dres.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE).tobytes() == res.tobytes()
    c.close()


def test_lk_whole_steps_low_inlier_ratio_many_ransac_rounds(pkg, oracle, tc, kitti):
    """A 0.05 px reprojection threshold leaves few inliers among ~2500 tracks: the adaptive stop then needs
    several 64-hypothesis rounds (hundreds of EPnP hypotheses) -- the place a divergence would hide."""
    seq, frames = kitti
    ref = _oracle_lk_steps(oracle, seq, frames[:3], reproj_err=0.05)
    assert max(p["ransac_iters"] for _, _, p, _ in ref) > 64
    _check_online_lk(pkg, oracle, seq, frames[:3], ref, reproj_err=0.05)


def test_lk_whole_steps_dense_corners_two_points_per_wave(pkg, oracle, tc, kitti):
    """A low FAST threshold gives a pair several thousand corners: the online launch then packs two or three
    points into a wave (lk_kernel spreads a single pair's points over the resident waves, one per wave up
    to 3072 points), and a batch of four pairs takes the ordinary four-points-per-wave path."""
    seq, frames = kitti
    thr = 8
    n0 = len(oracle.fast(frames[0][0], thr=thr))
    assert 3500 <= n0 <= 16000
    ref = _oracle_lk_steps(oracle, seq, frames[:3], fast_thr=thr)
    _check_online_lk(pkg, oracle, seq, frames[:3], ref, fast_threshold=thr, max_keypoints=16384)
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=4, fast_threshold=thr, max_keypoints=16384)
    fr5 = [frames[0], frames[1], frames[2], frames[1], frames[2]]     # pairs (0,1), (1,2), (2,1), (1,2): four items
    L = tc.stack([tc.from_numpy(f[0]) for f in fr5]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in fr5]).cuda()
    res = c.track_batch(L, R)
    for p in (0, 1):
        _check_record(res[p], ref[p][0], ref[p][2])
    assert res[3]["T_rel_inv"].tobytes() == res[1]["T_rel_inv"].tobytes()     # the same pair again: the same bits
    c.close()


def test_lk_whole_step_hd_stress_size(pkg, oracle, tc, synth):
    """BASELINE config #4: 1920x1080, FAST threshold raised so that a frame has about 2000 corners."""
    seq, frames = _render(synth, tc, 1920, 1080, 3, 1)
    thr = 40
    n0 = len(oracle.fast(frames[0][0], thr=thr))
    assert 800 <= n0 <= 6000
    ref = _oracle_lk_steps(oracle, seq, frames, fast_thr=thr)
    assert ref[0][0]["n_tracked"] > 300
    _check_online_lk(pkg, oracle, seq, frames, ref, fast_threshold=thr)


def test_orb_whole_steps_kitti_size(pkg, oracle, tc, kitti):
    """BASELINE config #3 at full size: ORB extraction of both images (byte-exact keypoints and descriptors),
    Hamming matches + filter, triangulation, RANSAC-PnP, gates, pose; online and batched."""
    seq, frames = kitti
    frames = frames[:3]
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    prm = oracle.make_params(P1, P2, min_t2=0.05 ** 2, max_t2=10.0 ** 2)
    feats = [(oracle.orb_extract(L)[:2], oracle.orb_extract(R)[:2]) for L, R in frames]
    kw = dict(P1=P1, P2=P2, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    c = pkg.Context(w, h, device=0, **kw)
    pose = np.eye(4)
    refs = []
    for t, fr in enumerate(frames):
        rc, g = c.add_frame(*fr)
        for side in (0, 1):
            k, d = c.frame_keypoints(side, with_descriptors=True)
            assert k.tobytes() == feats[t][side][0].tobytes() and d.tobytes() == feats[t][side][1].tobytes()
        if t == 0:
            continue
        (kL, dL), (kR, dR) = feats[t - 1]
        (k2, d2), _ = feats[t]
        r, pose = oracle.orb_track_step(prm, kL, dL, kR, dR, k2, d2, pose)
        t2l_r, t1l_r, t1r_r = oracle.orb_robust_match(kL, dL, kR, dR, k2, d2)
        X = oracle.triangulate(P1, P2, t1l_r, t1r_r)
        pnp = oracle.pnp_ransac(X, t2l_r, _K(P1))
        refs.append((r, pnp, pose.copy()))
        assert rc == (0 if r["ok"] else r["fail_stage"])
        _check_record(g, r, pnp)
        t1l, t1r, _, t2l, inl = c.last_tracks()
        assert t1l.tobytes() == t1l_r.tobytes() and t1r.tobytes() == t1r_r.tobytes() and t2l.tobytes() == t2l_r.tobytes()
        assert inl.tobytes() == pnp["mask"].tobytes()
        assert relfro(c.get_pose(), pose) <= TIGHT
    assert refs[0][0]["n_tracked"] > 50 and all(r["ok"] for r, _, _ in refs)
    c.close()
    c = pkg.Context(w, h, device=0, max_batch=2, **kw)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    for p, (r, pnp, pose) in enumerate(refs):
        _check_record(res[p], r, pnp)
        assert relfro(res[p]["pose"].reshape(4, 4), pose) <= TIGHT
    c.close()


def test_orb_capacity_overflow_fails_the_pairs_loudly(pkg, tc, small_seq):
    """ORB mode must not track a silently truncated set either: more keypoints than max_keypoints
    (nFeatures > max_keypoints), or more FAST candidates than a level's capacity, fail every pair that
    uses the image with SVO_FAIL_CAPACITY (6) in the online AND the batched path; the pose chain skips it."""
    from conftest import rand_image
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    kw = dict(P1=P1, P2=P2, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2, orb_nlevels=3)
    # (a) nFeatures 600 against max_keypoints 64
    c = pkg.Context(w, h, device=0, max_keypoints=64, max_batch=3, orb_nfeatures=600, **kw)
    c.add_frame(*frames[0])
    rc, r = c.add_frame(*frames[1])
    assert rc == 6 and int(r["ok"]) == 0 and int(r["fail_stage"]) == 6
    assert np.array_equal(c.get_pose(), np.eye(4))
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    assert [int(x) for x in res["fail_stage"]] == [6, 6, 6]
    with pytest.raises(pkg.SvoError):
        c.orb_extract(frames[0][0])                                    # the stage call reports it as an error
    c.close()
    # (b) a dense random frame in the middle of a batch overflows the per-level candidate capacity
    #     (4 * max_keypoints): only the pairs that touch it fail, the others track normally
    c = pkg.Context(w, h, device=0, max_keypoints=384, max_batch=4, orb_nfeatures=200, **kw)
    dense = rand_image(h, w, 3, blocks=False)
    fs = [frames[0], frames[1], (dense, dense), frames[2], frames[3]]
    L = tc.stack([tc.from_numpy(np.ascontiguousarray(f[0])) for f in fs]).cuda()
    R = tc.stack([tc.from_numpy(np.ascontiguousarray(f[1])) for f in fs]).cuda()
    res = c.track_batch(L, R)
    stages = [int(x) for x in res["fail_stage"]]
    assert stages[1] == 6 and stages[2] == 6 and stages[0] != 6 and stages[3] != 6
    # the flags are per batch: the same context tracks clean frames afterwards
    res2 = c.track_batch(L[:2], R[:2])
    assert int(res2[0]["fail_stage"]) == stages[0] and res2[0].tobytes() == res[0].tobytes()
    c.close()


def test_queued_batches_keep_their_own_seed_pose(pkg, tc, small_seq):
    """svo_track_batch with device results never synchronises the host, so many batches can be queued
    before the first one runs: each must chain from the pose0 IT was given (the seed travels by value in
    the launch).  Six queued calls with distinct seeds, stream order and overlap mode."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    F = len(frames)
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=F - 1)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    base = c.track_batch(L, R)                                   # identity seed, host results
    assert int(base[-1]["ok"]) == 1
    c.set_stream(tc.cuda.current_stream().cuda_stream)
    for overlap in (False, True):
        c.set_overlap(overlap)
        seeds, outs = [], []
        for k in range(6):
            p0 = np.eye(4)
            p0[:3, 3] = [10.0 * (k + 1), -3.0 * k, 0.5 * k]
            seeds.append(p0)
            outs.append(tc.zeros((F - 1, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda"))
        # ~50 ms of spinning in front of the first launch: all six calls are queued before any of them runs
        tc.cuda._sleep(120_000_000)
        for k in range(6):
            c.track_batch(L, R, pose0=seeds[k], results=outs[k])
        c.sync()
        tc.cuda.synchronize()
        for k in range(6):
            got = np.frombuffer(outs[k].cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
            for p in range(F - 1):
                want = seeds[k] @ base[p]["pose"].reshape(4, 4)
                assert np.abs(got[p]["pose"].reshape(4, 4) - want).max() < 1e-9, (overlap, k, p)
    c.close()

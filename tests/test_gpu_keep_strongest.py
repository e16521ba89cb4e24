"""svo_config.fast_keep_strongest and svo_get_batch_tracks (-m gpu).

BASELINE config #4 is "1920x1080, 2000 features": SURVEY.md 8(d) words it as EXACTLY the 2000 highest-response
FAST corners per frame (ties: raster order first).  The reference tracks every cv::FAST corner
(src/tracking.cpp:94-113); the additive option keeps the N strongest ON THE DEVICE so that the fused entry points
(svo_add_frame / svo_track_batch) run that configuration.  Oracle side: np.argsort(-response, kind="stable")[:N],
then the ordinary LK step on those corners."""
import numpy as np
import pytest

from test_gpu_parity_fullsize import POSE_TOL, TIGHT, _K, _render, relfro

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


def _strongest(kps, n):
    return kps[np.sort(np.argsort(-kps["response"], kind="stable")[:n])] if len(kps) > n else kps


def _oracle_pairs(oracle, seq, frames, keep):
    P1, P2 = seq.proj()
    prm = oracle.make_params(P1, P2)
    out, pose = [], np.eye(4)
    for t in range(1, len(frames)):
        allk = oracle.fast(frames[t - 1][0])
        sel = _strongest(allk, keep)
        res, cur, _ = oracle.lk_track_step(prm, *frames[t - 1], *frames[t], sel, np.eye(4), want_tracks=True, threads=8)
        X = oracle.triangulate(P1, P2, res["tracks"][0], res["tracks"][1])
        pnp = oracle.pnp_ransac(X, res["tracks"][3], _K(P1))
        if res["ok"]:
            pose = pose @ res["T_rel_inv"]
        out.append((res, len(sel), min(len(cur), keep), pnp, pose.copy(), len(allk)))
    return out


def _check_pairs(c, recs, ref, tracks_of):
    for p, (r, n_prev, n_cur, pnp, pose, n_all) in enumerate(ref):
        g = recs[p]
        assert int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"]
        assert int(g["n_prev_kps"]) == n_prev and int(g["n_cur_kps"]) == n_cur          # the KEPT counts
        assert int(g["n_tracked"]) == r["n_tracked"] and int(g["n_inliers"]) == r["n_inliers"]
        assert int(g["ransac_iters"]) == pnp["ransac_iters"] and int(g["lm_iters"]) == pnp["lm_iters"]
        t1l, t1r, t2r, t2l, inl = tracks_of(p)
        for got, want in zip((t1l, t1r, t2r, t2l), r["tracks"]):
            assert got.tobytes() == want.tobytes(), p
        assert inl.tobytes() == pnp["mask"].tobytes(), p
        assert relfro(g["pose"].reshape(4, 4), pose) <= TIGHT * 10


@pytest.mark.parametrize("keep", [31, 37, 150, 100000])
def test_keep_strongest_small_batched_and_online(pkg, oracle, tc, small_seq, keep):
    """416x128: cuts at several depths of the response histogram (FAST scores are small integers: a cut almost
    always falls inside a run of equal responses, which the raster order breaks) and one larger than the corner
    count (nothing dropped)."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    ref = _oracle_pairs(oracle, seq, frames, keep)
    n_all = ref[0][5]
    assert n_all > 150
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=len(frames) - 1, fast_keep_strongest=keep)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    _check_pairs(c, res, ref, c.batch_tracks)
    with pytest.raises(pkg.SvoError):
        c.batch_tracks(len(frames) - 1)                       # not a pair of the last launch
    c.close()
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, fast_keep_strongest=keep)
    c.add_frame(*frames[0])
    kp = c.frame_keypoints()
    want = _strongest(oracle.fast(frames[0][0]), keep)
    assert kp.tobytes() == want.tobytes()                     # the kept corners, raster order, responses intact
    for t in range(1, len(frames)):
        rc, g = c.add_frame(*frames[t])
        r = ref[t - 1][0]
        assert int(g["n_tracked"]) == r["n_tracked"] and int(g["n_inliers"]) == r["n_inliers"]
        got = c.last_tracks()
        for a, b in zip(got[:4], r["tracks"]):
            assert a.tobytes() == b.tobytes()
    c.close()


def test_batch_tracks_default_mode(pkg, oracle, tc, small_seq):
    """svo_get_batch_tracks on an ordinary batch (every corner tracked): tracks and inlier masks of every pair."""
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    ref = _oracle_pairs(oracle, seq, frames, 1 << 30)
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=len(frames) - 1)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    c.set_overlap(True)
    dres = tc.zeros((len(ref), pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
    tc.cuda.synchronize()
    c.track_batch(L, R, results=dres)
    tr = [c.batch_tracks(p) for p in range(len(ref))]         # waits for the side stream's pose stage
    c.sync()
    res = np.frombuffer(dres.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE)
    _check_pairs(c, res, ref, lambda p: tr[p])
    c.close()


def test_hd_batch_on_exactly_2000_strongest(pkg, oracle, tc, synth):
    """BASELINE config #4 through the FUSED path: 1920x1080, fast_keep_strongest = 2000, svo_track_batch."""
    seq, frames = _render(synth, tc, 1920, 1080, 3, 1)
    ref = _oracle_pairs(oracle, seq, frames, 2000)
    assert all(n_prev == 2000 and n_all > 2000 for _, n_prev, _, _, _, n_all in ref)
    assert ref[0][0]["n_tracked"] > 300 and all(r["ok"] for r, *_ in ref)
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_batch=2, max_keypoints=1 << 16, fast_keep_strongest=2000)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    _check_pairs(c, res, ref, c.batch_tracks)
    for p, (r, *_rest) in enumerate(ref):
        assert relfro(res[p]["T_rel_inv"].reshape(4, 4), r["T_rel_inv"]) <= min(POSE_TOL, TIGHT)
    c.close()

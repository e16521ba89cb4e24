"""GPU parity tests (-m gpu) for the ORB path (BASELINE config #3): HIP through
the C-ABI vs oracle/orb.c.  Pyramids, keypoints (position, angle, response, octave, size),
256-bit descriptors and Hamming matches are all integer / single-precision with a fixed operation
order: the bar is BIT-EXACT."""
import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.mark.parametrize("w,h,seed", [(1241, 376, 3), (416, 128, 4), (333, 201, 5)])
def test_orb_extract_parity(pkg, oracle, tc, w, h, seed):
    img = rand_image(h, w, seed)
    # the random block image is far denser in corners than a real frame: room for 4 * 8192 candidates per level
    ctx = pkg.Context(w, h, device=0, track_mode=pkg.MODE_ORB, max_keypoints=8192)
    kps, desc, per = ctx.orb_extract(img)
    rk, rd, rper = oracle.orb_extract(img)
    for l in range(8):
        assert np.array_equal(ctx.orb_read_level(l), oracle.orb_pyramid_level(img, l)), f"pyramid level {l}"
    for l in range(8):
        assert ctx.orb_read_candidates(l, cap=32768).tobytes() == oracle.orb_candidates(img, l).tobytes(), f"candidates level {l}"
    assert per.tolist() == rper.tolist()
    assert len(kps) == len(rk) > 100
    assert kps.tobytes() == rk.tobytes()
    assert desc.tobytes() == rd.tobytes()
    # device-resident image gives the same answer
    k2, d2, _ = ctx.orb_extract(tc.from_numpy(img).cuda())
    assert k2.tobytes() == rk.tobytes() and d2.tobytes() == rd.tobytes()
    ctx.close()


def test_orb_extract_synthetic_and_flat(pkg, oracle, tc, small_seq):
    seq, frames = small_seq
    h, w = frames[0][0].shape
    ctx = pkg.Context(w, h, device=0, track_mode=pkg.MODE_ORB)
    for L, R in frames[:2]:
        for im in (L, R):
            kps, desc, per = ctx.orb_extract(im)
            rk, rd, rper = oracle.orb_extract(im)
            assert kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes()
    flat = np.full((h, w), 77, np.uint8)
    assert len(ctx.orb_extract(flat)[0]) == 0
    # weak texture: cells fall back to the minimum FAST threshold (7)
    rng = np.random.default_rng(0)
    weak = (100 + rng.integers(0, 12, (h // 4, w // 4))).astype(np.uint8).repeat(4, 0).repeat(4, 1)[:h, :w]
    kps, desc, per = ctx.orb_extract(weak)
    rk, rd, rper = oracle.orb_extract(weak)
    assert kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes()
    assert len(rk) > 0 and rk["response"].min() < 20
    ctx.close()


def test_match_hamming_parity(pkg, oracle, tc):
    rng = np.random.default_rng(2)
    ctx = pkg.Context(416, 128, device=0, track_mode=pkg.MODE_ORB)
    t = rng.integers(0, 256, (2011, 32), dtype=np.uint8)
    q = np.concatenate([t[rng.permutation(2011)[:900]], rng.integers(0, 256, (1100, 32), dtype=np.uint8)])
    q[3, 0] ^= 1
    t[1500] = t[10]                                           # duplicate row: the first minimum must win
    ridx, rdist = oracle.match_hamming(q, t)
    idx, dist = ctx.match_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    idx_d, dist_d = ctx.match_hamming(tc.from_numpy(q).cuda(), tc.from_numpy(t).cuda())
    assert np.array_equal(idx_d.cpu().numpy(), ridx) and np.array_equal(dist_d.cpu().numpy(), rdist)
    i0, d0 = ctx.match_hamming(q[:0], t)
    assert len(i0) == 0
    ctx.close()


POSE_TOL, TIGHT = 1e-4, 1e-9


def relfro(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300)


def _oracle_orb_sequence(oracle, seq, frames, minmove=0.05, maxmove=10.0):
    prm = oracle.make_params(*seq.proj(), min_t2=minmove ** 2, max_t2=maxmove ** 2)
    feats = [(oracle.orb_extract(L)[:2], oracle.orb_extract(R)[:2]) for L, R in frames]
    pose = np.eye(4)
    out = []
    for t in range(1, len(frames)):
        (kL, dL), (kR, dR) = feats[t - 1]
        (k2, d2), _ = feats[t]
        res, pose = oracle.orb_track_step(prm, kL, dL, kR, dR, k2, d2, pose)
        out.append((res, pose.copy()))
    return out, feats


def _check_orb_step(g, r):
    assert int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"]
    assert int(g["n_prev_kps"]) == r["n_prev_kps"] and int(g["n_cur_kps"]) == r["n_cur_kps"]
    assert int(g["n_tracked"]) == r["n_tracked"]
    if r["fail_stage"] != 2:
        assert int(g["n_inliers"]) == r["n_inliers"]
        Tg = np.hstack([g["R"].reshape(3, 3), g["tvec"][:, None]])
        Tr = np.hstack([r["R"], r["tvec"][:, None]])
        assert relfro(Tg, Tr) <= POSE_TOL and relfro(Tg, Tr) <= TIGHT


def test_orb_mode_online_and_batch_parity(pkg, oracle, tc, synth):
    """Tracking::ORB_StereoF2F_PnP_Track end to end (shipped default track_mode of config/default.yaml)."""
    seq = synth.StereoSequence(width=832, height=256, n_frames=4, seed=5)
    frames = [tuple(x.numpy() for x in seq.render(t)) for t in range(4)]
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    ref, feats = _oracle_orb_sequence(oracle, seq, frames)
    assert all(r["ok"] for r, _ in ref) and ref[0][0]["n_tracked"] > 20
    kw = dict(P1=P1s, P2=P2s, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=10.0 ** 2)
    c = pkg.Context(w, h, device=0, **kw)
    rc, g0 = c.add_frame(*frames[0])
    assert rc == 0 and g0["n_cur_kps"] == len(feats[0][0][0])
    for t in range(1, 4):
        fr = frames[t] if t % 2 else tuple(tc.from_numpy(x).cuda() for x in frames[t])
        rc, g = c.add_frame(*fr)
        r, pose = ref[t - 1]
        assert rc == (0 if r["ok"] else r["fail_stage"])
        _check_orb_step(g, r)
        assert relfro(c.get_pose(), pose) <= TIGHT
    c.close()
    # batched
    c = pkg.Context(w, h, device=0, max_batch=3, **kw)
    L = tc.stack([tc.from_numpy(f[0]) for f in frames]).cuda()
    R = tc.stack([tc.from_numpy(f[1]) for f in frames]).cuda()
    res = c.track_batch(L, R)
    for p in range(3):
        _check_orb_step(res[p], ref[p][0])
        assert relfro(res[p]["pose"].reshape(4, 4), ref[p][1]) <= TIGHT
    # overlap mode gives the same records
    c.set_overlap(True)
    dres = tc.zeros((3, pkg.STEP_DTYPE.itemsize), dtype=tc.uint8, device="cuda")
    c.track_batch(L, R, results=dres)
    c.track_batch(L, R, results=dres)
    c.sync()
    assert np.frombuffer(dres.cpu().numpy().tobytes(), dtype=pkg.STEP_DTYPE).tobytes() == res.tobytes()
    c.close()
    # frames with padded rows (pitch 864: 16-byte aligned, >= 16 bytes of padding) are read IN PLACE as pyramid level 0 -- the
    # tensors above (pitch = width) were copied into the slots first --; same records, also with the copy forced
    Lp = tc.zeros((4, h, w + 32), dtype=tc.uint8, device="cuda")
    Rp = tc.zeros((4, h, w + 32), dtype=tc.uint8, device="cuda")
    Lp[:, :, :w] = L
    Rp[:, :, :w] = R
    Lp[:, :, w:] = 0xA5                                    # (whatever lies in the padding must not matter)
    import os
    for force_copy in (False, True):
        if force_copy:
            os.environ["SVO_ORB_COPY_LEVEL0"] = "1"
        try:
            c = pkg.Context(w, h, device=0, max_batch=3, **kw)
            assert c.track_batch(Lp[:, :, :w], Rp[:, :, :w]).tobytes() == res.tobytes()
            c.close()
        finally:
            os.environ.pop("SVO_ORB_COPY_LEVEL0", None)


def test_orb_mode_failure_stages(pkg, oracle, tc, synth):
    seq = synth.StereoSequence(width=832, height=256, n_frames=2, seed=6)
    frames = [tuple(x.numpy() for x in seq.render(t)) for t in range(2)]
    h, w = frames[0][0].shape
    P1s, P2s = seq.proj()
    flat = np.full((h, w), 60, np.uint8)
    # too few matches (stage 2): the current frame has no features at all
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, track_mode=pkg.MODE_ORB, min_move2=0.05 ** 2, max_move2=100.0)
    c.add_frame(*frames[0])
    rc, g = c.add_frame(flat, flat)
    assert rc == 2 and g["n_tracked"] == 0 and g["n_cur_kps"] == 0
    rc, g = c.add_frame(*frames[1])                      # last frame has no features either
    assert rc == 2 and g["n_prev_kps"] == 0
    c.close()
    # translation gate with the configured minmove (stage 5): minmove larger than the motion
    ref, _ = _oracle_orb_sequence(oracle, seq, frames, minmove=3.0, maxmove=10.0)
    c = pkg.Context(w, h, device=0, P1=P1s, P2=P2s, track_mode=pkg.MODE_ORB, min_move2=9.0, max_move2=100.0)
    c.add_frame(*frames[0])
    rc, g = c.add_frame(*frames[1])
    assert ref[0][0]["fail_stage"] == 5 and rc == 5
    _check_orb_step(g, ref[0][0])
    assert np.array_equal(c.get_pose(), np.eye(4))
    c.close()


@pytest.mark.parametrize("cfg", [dict(nlevels=1, scale_factor=1.2, nfeatures=800, ini_th=20, min_th=7),
                                 dict(nlevels=5, scale_factor=1.5, nfeatures=1200, ini_th=30, min_th=10),
                                 dict(nlevels=8, scale_factor=1.1, nfeatures=300, ini_th=12, min_th=12),
                                 # a per-level quota whose node tables do not fit the node-parallel quadtree's LDS: the serial
                                 # wave-per-tree kernel takes over
                                 dict(nlevels=1, scale_factor=1.2, nfeatures=3000, ini_th=9, min_th=5)])
def test_orb_extract_other_configurations(pkg, oracle, tc, synth, cfg):
    """ORBextractor with non-default YAML values (nLevels, fScaleFactor, nFeatures, FAST thresholds):
    resize tables, quotas, cell grid and quadtree all follow the configuration; bit-exact."""
    seq = synth.StereoSequence(width=500, height=300, n_frames=1, seed=11, supersample=1)
    img = seq.render(0)[0].numpy()
    ctx = pkg.Context(500, 300, device=0, track_mode=pkg.MODE_ORB, orb_nlevels=cfg["nlevels"],
                      orb_scale_factor=cfg["scale_factor"], orb_nfeatures=cfg["nfeatures"],
                      orb_ini_th=cfg["ini_th"], orb_min_th=cfg["min_th"])
    kps, desc, per = ctx.orb_extract(img)
    rk, rd, rper = oracle.orb_extract(img, **cfg)
    for l in range(cfg["nlevels"]):
        assert np.array_equal(ctx.orb_read_level(l), oracle.orb_pyramid_level(img, l, scale_factor=cfg["scale_factor"],
                                                                              nlevels=cfg["nlevels"])), f"level {l}"
    assert per.tolist()[:cfg["nlevels"]] == rper.tolist()[:cfg["nlevels"]]
    assert len(rk) > 50 and kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes()
    ctx.close()


def test_orb_extract_more_candidates_than_the_lds_key_arrays(pkg, oracle, tc, synth):
    """cv::FAST is uncapped: a busy image gives a level more candidates than the quadtree kernel keeps in LDS (3456);
    the tree then partitions its keys in global scratch.  Noise on a rendered frame until level 0 has more than that
    (and less than the capacity of 4 x nFeatures), then bit-exact keypoints and descriptors."""
    seq = synth.StereoSequence(width=1241, height=376, n_frames=1, seed=17, supersample=1)
    base = seq.render(0)[0].numpy().astype(np.int32)
    rng = np.random.default_rng(5)
    noise = rng.integers(-64, 65, base.shape)
    ctx = pkg.Context(1241, 376, device=0, track_mode=pkg.MODE_ORB)
    picked = None
    for amp in (0.15, 0.25, 0.35, 0.5, 0.7, 1.0):
        img = np.clip(base + (noise * amp).astype(np.int32), 0, 255).astype(np.uint8)
        try:
            kps, desc, per = ctx.orb_extract(img)
        except RuntimeError:
            break                                   # past the candidate capacity: the context refuses loudly
        n0 = len(ctx.orb_read_candidates(0, cap=16384))
        if n0 > 3456:
            picked = (img, kps, desc, per, n0)
            break
    assert picked is not None, "no noise level put level 0 between 3456 candidates and the capacity"
    img, kps, desc, per, n0 = picked
    rk, rd, rper = oracle.orb_extract(img)
    assert per.tolist() == rper.tolist()
    assert kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes()
    ctx.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (15, 257), (16, 256), (17, 255), (700, 64), (33, 1000)])
def test_match_hamming_shapes_and_ties(pkg, oracle, tc, nq, nt):
    """Row counts around the matcher's tile sizes (16 queries per workgroup, 256 train rows per
    tile) and heavy ties (few distinct rows): the first minimum must win everywhere."""
    rng = np.random.default_rng(nq * 1000 + nt)
    base = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    t = base[rng.integers(0, 5, nt)]
    q = base[rng.integers(0, 5, nq)].copy()
    q[:, 7] ^= rng.integers(0, 2, nq).astype(np.uint8)
    ctx = pkg.Context(416, 128, device=0, track_mode=pkg.MODE_ORB)
    ridx, rdist = oracle.match_hamming(q, t)
    idx, dist = ctx.match_hamming(q, t)
    assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist)
    ctx.close()


def test_orb_pyramid_both_resize_kernels(pkg, oracle, tc, monkeypatch):
    """ComputePyramid (src/ORBextractor.cpp:1061-1085) through the row-streaming resize kernel (scale factors in (1, 2]) and
    through the LDS-staged one it replaced (SVO_ORB_RESIZE_STAGED=1; taken by itself for a scale factor whose four taps do
    not fit the streaming kernel's 8-byte window): every level byte-equal to the oracle's cv::resize restatement, keypoints
    and descriptors byte-equal to the oracle's; host image and a device image with padded rows."""
    img = rand_image(201, 333, 7)
    img_dev = tc.from_numpy(np.ascontiguousarray(np.pad(img, ((0, 0), (0, 3))))).cuda()[:, :333]     # pitch 336
    for env, sf, nl in ((None, 1.2, 8), ("SVO_ORB_RESIZE_STAGED", 1.2, 8), (None, 2.0, 4), (None, 2.7, 3), (None, 1.05, 8)):
        monkeypatch.delenv("SVO_ORB_RESIZE_STAGED", raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        ctx = pkg.Context(333, 201, device=0, track_mode=pkg.MODE_ORB, max_keypoints=8192, orb_nlevels=nl, orb_scale_factor=sf)
        rk, rd, _ = oracle.orb_extract(img, nlevels=nl, scale_factor=sf, nfeatures=2000, ini_th=20, min_th=7)
        for src in (img, img_dev):
            kps, desc, _ = ctx.orb_extract(src)
            for l in range(nl):
                assert np.array_equal(ctx.orb_read_level(l), oracle.orb_pyramid_level(img, l, scale_factor=sf, nlevels=nl)), (env, sf, l)
            assert kps.tobytes() == rk.tobytes() and desc.tobytes() == rd.tobytes(), (env, sf)
        ctx.close()


def test_orb_level0_in_place_is_not_in_the_slot(pkg, oracle, tc, synth):
    """svo_add_frame in ORB mode reads pyramid level 0 in place from the (staged) input frame: the read-back of level 0 says so
    instead of returning a stale slot; the levels built from it are the oracle's."""
    seq = synth.StereoSequence(width=832, height=256, n_frames=1, seed=5)
    left, right = (x.numpy() for x in seq.render(0))
    c = pkg.Context(832, 256, device=0, track_mode=pkg.MODE_ORB)
    c.add_frame(left, right)
    with pytest.raises(pkg.SvoError):
        c.orb_read_level(0)
    assert np.array_equal(c.orb_read_level(1), oracle.orb_pyramid_level(left, 1))     # image slot 0 = the left image
    c.orb_extract(left)                                                                 # the stage API copies level 0
    assert np.array_equal(c.orb_read_level(0), left)
    c.close()

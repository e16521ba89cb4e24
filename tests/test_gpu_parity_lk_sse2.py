"""svo_config.lk_accum = SVO_LK_ACCUM_SSE2 / _SIMD128 (-m gpu): the HIP LK kernel that accumulates A11, A12, A22, b1, b2 in
float in a lane order of upstream's x86 SIMD code (csrc/lk_sse2.hip) against the oracle in the same mode
(oracle/lk.c, orc_lk_set_accum(2) / (4)) -- the orders restated from lkpyramid.cpp for the reference's four
cv::calcOpticalFlowPyrLK calls (src/tracking.cpp:593-618; not validated against an OpenCV binary: tests/test_cv_crosscheck.py
does that where cv2 exists).  Same bars as the exact mode: points, status bytes,
tracks, RANSAC masks and iteration numbers bit for bit, the chained pose within 1e-4 (north_star) and within the
observed 1e-9.  Stage level (random images, window over the edge, one to thousands of points), BASELINE
config #1's 100 S0 pairs (batched and online), a second seed, and config #4's 1920x1080 step on exactly the
2000 strongest corners."""
import contextlib

import numpy as np
import pytest

import conftest
from test_gpu_parity_fullsize import POSE_TOL, TIGHT, _K, relfro
from test_gpu_parity_sequence import _check_batch, _check_online, _oracle_lk_sequence, _render

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tc():
    import torch
    assert torch.cuda.is_available()
    return torch


@contextlib.contextmanager
def sse2_oracle(oracle):
    old = oracle.set_lk_accum(oracle.LK_ACCUM_FLOAT_SSE)
    try:
        yield
    finally:
        oracle.set_lk_accum(old)


@contextlib.contextmanager
def accum_oracle(oracle, mode):
    old = oracle.set_lk_accum(mode)
    try:
        yield
    finally:
        oracle.set_lk_accum(old)


def _shifted_pair(h, w, seed, dx=2.3, dy=-1.4):
    """A blocky random image and a smoothly warped copy (sub-pixel shift + a little shear)."""
    img = conftest.rand_image(h + 16, w + 16, seed).astype(np.float32)
    for _ in range(2):              # soften the block edges so that LK converges
        img = (img + np.roll(img, 1, 0) + np.roll(img, -1, 0) + np.roll(img, 1, 1) + np.roll(img, -1, 1)) / 5
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float32)
    sx, sy = xs + 8 + dx + 0.004 * ys, ys + 8 + dy
    x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
    fx, fy = sx - x0, sy - y0
    nxt = (img[y0, x0] * (1 - fx) * (1 - fy) + img[y0, x0 + 1] * fx * (1 - fy) + img[y0 + 1, x0] * (1 - fx) * fy +
           img[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(img[8:8 + h, 8:8 + w] + 0.5, 0, 255).astype(np.uint8), np.clip(nxt + 0.5, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,n,seed", [(416, 128, 1, 3), (416, 128, 5, 4), (333, 201, 700, 5), (1241, 376, 3000, 6)])
def test_lk_track_sse2_points_and_status(pkg, oracle, w, h, n, seed):
    """One cv::calcOpticalFlowPyrLK call: points in the interior, on the border (windows hanging over the edge:
    the zero border of the derivative image) and outside."""
    prev, nxt = _shifted_pair(h, w, seed)
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)
    pts[: n // 4] = np.round(pts[: n // 4])                      # integer positions: zero fractional weights
    c = pkg.Context(w, h, device=0, max_keypoints=max(n, 64), lk_accum=pkg.LK_ACCUM_SSE2)
    c.build_pyramid(0, prev)
    c.build_pyramid(1, nxt)
    got, st = c.lk_track(0, 1, pts)
    c.close()
    with sse2_oracle(oracle):
        want, wst = oracle.lk_track(prev, nxt, pts)
    exact, _ = oracle.lk_track(prev, nxt, pts)
    assert st.tobytes() == wst.tobytes()
    assert got.tobytes() == want.tobytes()
    if n >= 700:
        assert wst.sum() > n // 3
        assert want.tobytes() != exact.tobytes()                 # the mode does change bits (else the test proves nothing)


def test_circular_match_sse2(pkg, oracle, small_seq):
    seq, frames = small_seq
    h, w = frames[0][0].shape
    P1, P2 = seq.proj()
    imgs = [*frames[0], *frames[1]]
    kps = oracle.fast(imgs[0])
    pts = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, lk_accum=pkg.LK_ACCUM_SSE2)
    for s, im in enumerate(imgs):
        c.build_pyramid(s, im)
    out = c.circular_match((0, 1, 2, 3), pts)
    c.close()
    with sse2_oracle(oracle):
        res, _, _ = oracle.lk_track_step(oracle.make_params(P1, P2), *imgs, kps, np.eye(4), want_tracks=True)
    assert out[0].shape[0] == res["n_tracked"] > 20
    for k in range(4):
        assert out[k].tobytes() == res["tracks"][k].tobytes()


def test_lk_accum_is_validated(pkg):
    with pytest.raises(pkg.SvoError):
        pkg.Context(416, 128, device=0, lk_accum=7)


@pytest.fixture(scope="module")
def s0_100(synth, tc):
    return _render(synth, tc, 1241, 376, 101, 20200710)


def test_lk_sse2_100_pairs_batched_and_online(pkg, oracle, tc, s0_100):
    """BASELINE config #1's 100 pairs in the x86 accumulation order: every pair's counts, tracks, RANSAC record
    and mask bit for bit, chained pose <= 1e-9."""
    seq, frames = s0_100
    with sse2_oracle(oracle):
        ref = _oracle_lk_sequence(oracle, seq, frames)
    exact = _oracle_lk_sequence(oracle, seq, frames[:9])
    assert len(ref) == 100 and all(r["ok"] for r, _, _, _ in ref)
    assert any(a[0]["tracks"].tobytes() != b[0]["tracks"].tobytes() for a, b in zip(ref, exact))
    _, worst = _check_batch(pkg, tc, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2)
    print(f"100-pair chained pose, HIP(sse2) vs oracle(sse2): max rel. Frobenius {worst:.2e}")
    _check_online(pkg, seq, frames[:41], ref[:40], lk_accum=pkg.LK_ACCUM_SSE2)


def test_lk_sse2_through_run_kitti_stereo(pkg, oracle, s0_100, tmp_path):
    """YAML `lk_accum: sse2` through the drop-in binary (per-frame loop and batched runner): the pose file
    follows the oracle's x86-order chain, and differs from the `exact` run's file."""
    import os
    import subprocess
    from test_gpu_parity_sequence import HOST, _write_pgm
    from test_host_api import _write_yaml
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    seq, frames = s0_100
    frames = frames[:17]
    with sse2_oracle(oracle):
        ref = _oracle_lk_sequence(oracle, seq, frames)
    for cam in (0, 1):
        os.makedirs(tmp_path / f"image_{cam}")
    for t, (L, R) in enumerate(frames):
        _write_pgm(tmp_path / "image_0" / f"{t:06d}.pgm", L)
        _write_pgm(tmp_path / "image_1" / f"{t:06d}.pgm", R)
    _write_yaml(tmp_path / "exact.yaml", str(tmp_path), fx=seq.fx, fy=seq.fy, cx=seq.cx, cy=seq.cy)
    txt = open(tmp_path / "exact.yaml", encoding="utf-8").read()
    open(tmp_path / "sse2.yaml", "w", encoding="utf-8").write(txt + "lk_accum: sse2\n")
    open(tmp_path / "sse2_batched.yaml", "w", encoding="utf-8").write(txt + "lk_accum: sse2\nbatch_size: 8\ndecode_threads: 4\n")
    want = np.stack([np.eye(4)] + [pose for _, _, _, pose in ref])[:, :3]
    files = {}
    for cfg in ("exact.yaml", "sse2.yaml", "sse2_batched.yaml"):
        out = tmp_path / (cfg + ".poses")
        r = subprocess.run([os.path.join(HOST, "run_kitti_stereo"), str(tmp_path / cfg), str(out)], capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        files[cfg] = open(out, "rb").read()
        if cfg != "exact.yaml":
            poses = np.loadtxt(out).reshape(-1, 3, 4)
            assert poses.shape == want.shape
            assert max(relfro(poses[t], want[t]) for t in range(len(want))) <= 1e-6      # the file holds 10 significant digits
    assert files["sse2.yaml"] == files["sse2_batched.yaml"] != files["exact.yaml"]


def test_lk_sse2_second_seed_24_pairs(pkg, oracle, tc, synth):
    seq, frames = _render(synth, tc, 1241, 376, 25, 7)
    with sse2_oracle(oracle):
        ref = _oracle_lk_sequence(oracle, seq, frames)
    assert sum(r["ok"] for r, _, _, _ in ref) >= 23
    _check_batch(pkg, tc, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2)
    _check_online(pkg, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2)


def test_hd_sse2_whole_step_on_exactly_2000_strongest_corners(pkg, oracle, tc, synth):
    """BASELINE config #4 (1920x1080, the 2000 highest-response corners) in the x86 accumulation order."""
    w, h = 1920, 1080
    seq, frames = _render(synth, tc, w, h, 3, 1)
    P1, P2 = seq.proj()
    prm = oracle.make_params(P1, P2)
    c = pkg.Context(w, h, device=0, P1=P1, P2=P2, max_keypoints=1 << 16, lk_accum=pkg.LK_ACCUM_SSE2)
    pose_g, pose_r = np.eye(4), np.eye(4)
    for t in (1, 2):
        (L0, R0), (L1, R1) = frames[t - 1], frames[t]
        ref_kp = oracle.fast(L0, thr=20)
        sel = ref_kp[np.sort(np.argsort(-ref_kp["response"], kind="stable")[:2000])]
        assert len(sel) == 2000
        for s, im in enumerate((L0, R0, L1, R1)):
            c.build_pyramid(s, im)
        pts = np.stack([sel["x"], sel["y"]], 1).astype(np.float32)
        with sse2_oracle(oracle):
            res, _, pose_r = oracle.lk_track_step(prm, L0, R0, L1, R1, sel, pose_r, want_tracks=True, threads=8)
        got = c.circular_match((0, 1, 2, 3), pts)
        assert got[0].shape[0] == res["n_tracked"] > 300
        for k in range(4):
            assert got[k].tobytes() == res["tracks"][k].tobytes()
        X = oracle.triangulate(P1, P2, res["tracks"][0], res["tracks"][1])
        Xg = c.triangulate(P1, P2, got[0], got[1])
        assert Xg.tobytes() == X.tobytes()
        pnp = oracle.pnp_ransac(X, res["tracks"][3], _K(P1))
        sg = c.pnp_ransac(Xg, got[3], _K(P1))
        assert sg["best_iter"] == pnp["best_iter"] and sg["ransac_iters"] == pnp["ransac_iters"] and sg["lm_iters"] == pnp["lm_iters"]
        assert sg["n_inliers"] == pnp["n_inliers"] == res["n_inliers"] and np.array_equal(sg["mask"], pnp["mask"])
        rc, pose_g, _ = oracle.gate_and_accumulate(sg["R"], sg["tvec"], pose_g)
        assert rc >= 0 and relfro(pose_g, pose_r) <= POSE_TOL and relfro(pose_g, pose_r) <= TIGHT * 10
    c.close()


# ---- SVO_LK_ACCUM_SIMD128 (ABI v7): the universal-intrinsic block restated whole = oracle mode 4 ----------------------------
@pytest.mark.parametrize("w,h,n,seed", [(416, 128, 5, 4), (333, 201, 700, 5), (1241, 376, 3000, 6)])
def test_lk_track_simd128_points_and_status(pkg, oracle, w, h, n, seed):
    prev, nxt = _shifted_pair(h, w, seed)
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)
    pts[: n // 4] = np.round(pts[: n // 4])
    c = pkg.Context(w, h, device=0, max_keypoints=max(n, 64), lk_accum=pkg.LK_ACCUM_SIMD128)
    c.build_pyramid(0, prev)
    c.build_pyramid(1, nxt)
    got, st = c.lk_track(0, 1, pts)
    c.close()
    with accum_oracle(oracle, oracle.LK_ACCUM_SIMD128):
        want, wst = oracle.lk_track(prev, nxt, pts)
    with sse2_oracle(oracle):
        other, _ = oracle.lk_track(prev, nxt, pts)
    assert st.tobytes() == wst.tobytes() and got.tobytes() == want.tobytes()
    if n >= 700:
        assert want.tobytes() != other.tobytes()                 # the A order does change bits against the sse2 mode


def test_lk_simd128_24_pairs_batched_and_online(pkg, oracle, tc, synth):
    seq, frames = _render(synth, tc, 1241, 376, 25, 20200710)
    with accum_oracle(oracle, oracle.LK_ACCUM_SIMD128):
        ref = _oracle_lk_sequence(oracle, seq, frames)
    assert all(r["ok"] for r, _, _, _ in ref)
    _check_batch(pkg, tc, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SIMD128)
    _check_online(pkg, seq, frames[:9], ref[:8], lk_accum=pkg.LK_ACCUM_SIMD128)


# ---- SVO_LK_ACCUM_SSE2_LEGACY (ABI v8): the older hand-written CV_SSE2 block restated whole = oracle mode 3 ---------------------
@pytest.mark.parametrize("w,h,n,seed", [(416, 128, 1, 3), (416, 128, 5, 4), (333, 201, 700, 5), (1241, 376, 3000, 6)])
def test_lk_track_sse2_legacy_points_and_status(pkg, oracle, w, h, n, seed):
    """One float add per pixel product (pixels 0, 1, then 4, 5 of a group of eight into qb0; 2, 3, then 6, 7 into qb1) instead of
    the madd pair sums of the other two orders: bit-identical to oracle mode 3."""
    prev, nxt = _shifted_pair(h, w, seed)
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-3, w + 3, n), rng.uniform(-3, h + 3, n)], 1).astype(np.float32)
    pts[: n // 4] = np.round(pts[: n // 4])
    c = pkg.Context(w, h, device=0, max_keypoints=max(n, 64), lk_accum=pkg.LK_ACCUM_SSE2_LEGACY)
    c.build_pyramid(0, prev)
    c.build_pyramid(1, nxt)
    got, st = c.lk_track(0, 1, pts)
    c.close()
    with accum_oracle(oracle, oracle.LK_ACCUM_LEGACY_SSE2):
        want, wst = oracle.lk_track(prev, nxt, pts)
    assert st.tobytes() == wst.tobytes() and got.tobytes() == want.tobytes()
    # (against the madd order the bits differ only where a b sum passes 2^24 between a pair's two products: 6 of 2 478 points on
    #  an S0 frame pair, none on this shifted pair -- the 24-pair sequence below is where the two orders part)


def test_lk_sse2_legacy_24_pairs_batched_and_online(pkg, oracle, tc, synth):
    seq, frames = _render(synth, tc, 1241, 376, 25, 20200710)
    with accum_oracle(oracle, oracle.LK_ACCUM_LEGACY_SSE2):
        ref = _oracle_lk_sequence(oracle, seq, frames)
    with sse2_oracle(oracle):
        ref2 = _oracle_lk_sequence(oracle, seq, frames)
    assert all(r["ok"] for r, _, _, _ in ref)
    assert any(a[0]["tracks"][3].tobytes() != b[0]["tracks"][3].tobytes() for a, b in zip(ref, ref2))   # the two orders do part on 24 pairs
    _check_batch(pkg, tc, seq, frames, ref, lk_accum=pkg.LK_ACCUM_SSE2_LEGACY)
    _check_online(pkg, seq, frames[:9], ref[:8], lk_accum=pkg.LK_ACCUM_SSE2_LEGACY)

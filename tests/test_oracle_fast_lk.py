"""Known-answer tests that pin the CPU oracle's FAST / pyramid / LK restatement (CPU only).

The reference has no tests or golden vectors (parity unpinned), so the oracle is
pinned against independent definitions: brute-force FAST by definition, a numpy pyrDown, and LK
recovering analytically known sub-pixel shifts."""
import numpy as np

from conftest import rand_image

CIRC = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3),
        (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def _fast_bruteforce(img, thr):
    """FAST-9/16 straight from the definition: corner(t) iff 9 contiguous circle pixels are all
    > v+t or all < v-t; score = the largest t for which the pixel is still a corner."""
    h, w = img.shape
    im = img.astype(np.int32)
    score = np.zeros((h, w), np.int32)
    ys, xs = np.mgrid[3:h - 3, 3:w - 3]
    v = im[ys, xs]
    ring = np.stack([im[ys + dy, xs + dx] for dx, dy in CIRC], -1)           # (H, W, 16)
    ring2 = np.concatenate([ring, ring[..., :8]], -1)
    for t in range(thr, 256):
        br = ring2 > (v + t)[..., None]
        dk = ring2 < (v - t)[..., None]
        ok = np.zeros(v.shape, bool)
        for s in range(16):
            ok |= br[..., s:s + 9].all(-1) | dk[..., s:s + 9].all(-1)
        if not ok.any():
            break
        score[3:h - 3, 3:w - 3][ok] = t
    kps = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = score[y, x]
            if s == 0:
                continue
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                kps.append((x, y, s))
    return kps, score


def test_fast_matches_bruteforce_definition(oracle):
    for seed, thr in [(1, 20), (2, 7), (3, 40)]:
        img = rand_image(48, 56, seed)
        ref, _ = _fast_bruteforce(img, thr)
        got = oracle.fast(img, thr, True)
        assert len(ref) > 5
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == ref
        assert (got["size"] == 7).all() and (got["angle"] == -1).all()
        assert (got["octave"] == 0).all() and (got["class_id"] == -1).all()


def test_fast_square_corners_and_edges(oracle):
    img = np.full((40, 40), 30, np.uint8)
    img[10:30, 10:30] = 200
    # without NMS (a perfectly flat square gives tied scores, and strict NMS drops ties)
    kp = oracle.fast(img, 20, False)
    pts = {(int(k["x"]), int(k["y"])) for k in kp}
    # the four corner pixels of the bright square are corners; no straight-edge pixel is
    for c in [(10, 10), (29, 10), (10, 29), (29, 29)]:
        assert any(abs(c[0] - p[0]) <= 1 and abs(c[1] - p[1]) <= 1 for p in pts), (c, pts)
    for p in pts:
        assert min(abs(p[0] - 10), abs(p[0] - 29)) <= 2 and min(abs(p[1] - 10), abs(p[1] - 29)) <= 2
    # flat image: nothing; tiny image: nothing
    assert len(oracle.fast(np.full((32, 32), 128, np.uint8))) == 0
    assert len(oracle.fast(np.zeros((6, 6), np.uint8))) == 0


def test_fast_row_major_order_and_no_nms(oracle):
    img = rand_image(64, 80, 5)
    kp = oracle.fast(img, 20, True)
    key = kp["y"].astype(np.int64) * 10000 + kp["x"].astype(np.int64)
    assert (np.diff(key) > 0).all()
    kp0 = oracle.fast(img, 20, False)
    assert len(kp0) >= len(kp) and (kp0["response"] == 0).all()
    s_all = {(int(k["x"]), int(k["y"])) for k in kp0}
    assert {(int(k["x"]), int(k["y"])) for k in kp} <= s_all


def _pyr_down_numpy(img):
    k = np.array([1, 4, 6, 4, 1], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")       # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    dh, dw = (h + 1) // 2, (w + 1) // 2
    out = np.zeros((dh, dw), np.int64)
    for dy in range(5):
        for dx in range(5):
            # for odd sizes the last output reads one row/col past the padded block -> pad more
            pp = np.pad(img.astype(np.int64), 3, mode="reflect")
            out += k[dy] * k[dx] * pp[1 + dy:1 + dy + 2 * dh:2, 1 + dx:1 + dx + 2 * dw:2]
    del p
    return ((out + 128) >> 8).astype(np.uint8)


def test_pyr_down_matches_numpy(oracle):
    for (h, w, seed) in [(47, 156, 1), (94, 311, 2), (60, 80, 3), (33, 35, 4)]:
        img = rand_image(h, w, seed, blocks=False)
        assert np.array_equal(oracle.pyr_down(img), _pyr_down_numpy(img))


def test_pyramid_levels_and_reflect_border(oracle):
    img = rand_image(376 // 2, 1241 // 2, 9)
    p = oracle.PyramidHandle(img, 21, 3)
    assert p.nlevels == 4
    lv = img
    for l in range(4):
        assert np.array_equal(p.level(l), lv)
        pad = p.level(l, padded=True)
        assert np.array_equal(pad[21:-21, 21:-21], lv)
        assert np.array_equal(pad, np.pad(lv, 21, mode="reflect"))
        lv = _pyr_down_numpy(lv)
    # buildOpticalFlowPyramid stops before a level would be <= the window
    small = rand_image(80, 100, 2)
    assert oracle.PyramidHandle(small, 21, 3).nlevels == 2      # 50x40 ok, 25x20 has h <= 21


def _smooth_image(h, w, shift=(0.0, 0.0), seed=0):
    """Analytic band-limited texture sampled at (x + sx, y + sy): known sub-pixel shifts."""
    rng = np.random.default_rng(seed)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    xs = xs + shift[0]
    ys = ys + shift[1]
    img = np.zeros((h, w))
    for _ in range(24):
        fx, fy = rng.uniform(-0.35, 0.35, 2)
        ph = rng.uniform(0, 2 * np.pi)
        img += rng.uniform(0.3, 1.0) * np.cos(xs * fx + ys * fy + ph)
    img = (img - img.min()) / (img.max() - img.min())
    return np.round(img * 255).astype(np.uint8)


def test_lk_recovers_known_subpixel_shift(oracle):
    h, w = 200, 320
    for shift in [(3.3, -1.7), (-0.4, 0.25), (7.6, 2.2)]:
        I = _smooth_image(h, w, (0, 0), seed=3)
        # J(x) = I(x - d)  =>  a feature at p in I is found at p + d in J
        J = _smooth_image(h, w, (-shift[0], -shift[1]), seed=3)
        pts = np.array([[x, y] for y in range(40, 161, 30) for x in range(40, 281, 40)], np.float32)
        out, st = oracle.lk_track(I, J, pts)
        assert st.all()
        d = out - pts
        # 8-bit quantisation + the 0.01 px termination threshold bound the accuracy
        assert np.abs(d[:, 0] - shift[0]).max() < 0.1 and np.abs(d[:, 1] - shift[1]).max() < 0.1
        assert np.abs(d - np.array(shift)).mean() < 0.03


def test_lk_status_semantics(oracle):
    h, w = 120, 160
    I = _smooth_image(h, w, seed=5)
    flat = np.full((h, w), 77, np.uint8)
    pts = np.array([[80, 60]], np.float32)
    # untextured window: minEig < 1e-3 at level 0 -> status 0, point stays put
    out, st = oracle.lk_track(flat, flat, pts)
    assert st[0] == 0 and np.allclose(out, pts)
    # far outside the image: level-0 window out of bounds -> status 0
    out, st = oracle.lk_track(I, I, np.array([[500, 60], [-40, -40]], np.float32))
    assert (st == 0).all()
    # identical images: zero flow, status 1, including points whose window hangs over the border
    p2 = np.array([[3, 3], [w - 2, h - 2], [80, 2]], np.float32)
    out, st = oracle.lk_track(I, I, p2)
    assert st.all() and np.abs(out - p2).max() < 1e-3
    # empty input
    out, st = oracle.lk_track(I, I, np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)


def test_lk_thread_count_does_not_change_results(oracle):
    I = _smooth_image(150, 200, seed=8)
    J = _smooth_image(150, 200, (-1.2, 0.6), seed=8)
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(-5, 205, 300), rng.uniform(-5, 155, 300)], 1).astype(np.float32)
    o1, s1 = oracle.lk_track(I, J, pts, threads=1)
    o4, s4 = oracle.lk_track(I, J, pts, threads=4)
    assert np.array_equal(o1, o4) and np.array_equal(s1, s4)


def test_circular_keep_filter(oracle):
    n = 8
    p = [np.tile(np.array([[10.0, 10.0]], np.float32), (n, 1)) for _ in range(5)]
    s = [np.ones(n, np.uint8) for _ in range(4)]
    p[3][1, 0] = -0.5            # outside frame (x < 0)
    s[2][2] = 0                  # bad status
    p[1][3, 1] = 13.5            # |y0 - y1| = 3.5 > 3
    p[1][4, 1] = 13.0            # == 3 is kept (reject only on >)
    p[3][5, 1] = 6.9             # |y2 - y3| = 3.1 > 3
    p[4][6, 1] = -1.0            # loop-closure point outside
    keep, m = oracle.circular_keep(*p, *s, match_err=3.0)
    assert keep.tolist() == [1, 0, 0, 0, 1, 0, 0, 1] and m == 3

"""GPU parity tests (run with -m gpu on an MI355X): HIP FAST / pyramid / LK through the C-ABI vs
the CPU oracle on the same seeded inputs.  Everything here is integer or canonical-recipe float,
so the bar is BIT-EXACT."""
import numpy as np
import pytest

from conftest import rand_image

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the hot path has no CPU fallback"
    return torch


def _ctx(pkg, w, h, **kw):
    return pkg.Context(w, h, device=0, **kw)


@pytest.mark.parametrize("w,h,seed,thr", [(416, 128, 1, 20), (333, 77, 2, 20), (64, 64, 3, 7),
                                          (1241, 376, 4, 20), (129, 65, 5, 40)])
def test_fast_parity_random_images(pkg, oracle, torch_cuda, w, h, seed, thr):
    img = rand_image(h, w, seed)
    ctx = _ctx(pkg, w, h, max_keypoints=1 << 17)
    ref = oracle.fast(img, thr, True)
    # device-resident input (unaligned pitch = w) and host input must both match
    got_d = ctx.fast_detect(torch_cuda.from_numpy(img).cuda(), thr, True)
    got_h = ctx.fast_detect(img, thr, True)
    assert len(ref) > 0
    assert got_d.tobytes() == ref.tobytes()
    assert got_h.tobytes() == ref.tobytes()
    ref0 = oracle.fast(img, thr, False)
    got0 = ctx.fast_detect(img, thr, False)
    assert got0.tobytes() == ref0.tobytes()
    ctx.close()


def test_fast_edge_cases(pkg, oracle, torch_cuda):
    ctx = _ctx(pkg, 96, 64)
    flat = np.full((64, 96), 99, np.uint8)
    assert len(ctx.fast_detect(flat)) == 0
    # corners hugging the 3-pixel exclusion border, and saturated values
    img = np.zeros((64, 96), np.uint8)
    img[0:5, 0:5] = 255
    img[59:64, 91:96] = 255
    img[30:34, 40:44] = 255
    img += (np.arange(96, dtype=np.uint8) % 3)[None, :]
    ref = oracle.fast(img, 20, True)
    got = ctx.fast_detect(img, 20, True)
    assert got.tobytes() == ref.tobytes()
    ctx.close()


def test_fast_capacity_overflow_is_an_error(pkg, torch_cuda):
    img = rand_image(128, 416, 8)
    ctx = _ctx(pkg, 416, 128, max_keypoints=64)
    with pytest.raises(pkg.SvoError):
        ctx.fast_detect(img)
    ctx.close()


@pytest.mark.parametrize("w,h", [(416, 128), (1241, 376), (311, 95), (100, 80)])
def test_pyramid_parity(pkg, oracle, torch_cuda, w, h):
    img = rand_image(h, w, 21, blocks=False)
    ctx = _ctx(pkg, w, h)
    ctx.build_pyramid(0, torch_cuda.from_numpy(img).cuda())
    ctx.build_pyramid(1, img)
    ref = oracle.PyramidHandle(img, 21, 3)
    assert ctx.num_levels == ref.nlevels
    for l in range(ref.nlevels):
        assert np.array_equal(ctx.read_pyramid_level(0, l), ref.level(l))
        assert np.array_equal(ctx.read_pyramid_level(1, l), ref.level(l))
    ctx.close()


def _lk_points(w, h, n, seed):
    rng = np.random.default_rng(seed)
    pts = np.stack([rng.uniform(-12, w + 12, n), rng.uniform(-12, h + 12, n)], 1).astype(np.float32)
    # exact integers, half pixels, image corners, far outside
    extra = np.array([[0, 0], [w - 1, h - 1], [10.5, 10.5], [w / 2, h / 2], [-25, 5], [w + 30, h + 30],
                      [3, h - 1], [w - 1, 3]], np.float32)
    return np.concatenate([pts, extra])


def test_lk_single_call_parity_synthetic_stereo(pkg, oracle, torch_cuda, small_seq):
    seq, frames = small_seq
    L0, R0 = frames[0]
    h, w = L0.shape
    ctx = _ctx(pkg, w, h)
    ctx.build_pyramid(0, L0)
    ctx.build_pyramid(1, R0)
    kp = oracle.fast(L0)
    pts = np.concatenate([np.stack([kp["x"], kp["y"]], 1), _lk_points(w, h, 200, 3)]).astype(np.float32)
    ref_out, ref_st = oracle.lk_track(L0, R0, pts)
    out, st = ctx.lk_track(0, 1, pts)
    assert np.array_equal(st, ref_st)
    assert out.tobytes() == ref_out.tobytes()            # bit-exact floats
    # device-memory path
    out_d, st_d = ctx.lk_track(0, 1, torch_cuda.from_numpy(pts).cuda())
    assert np.array_equal(st_d.cpu().numpy(), ref_st)
    assert out_d.cpu().numpy().tobytes() == ref_out.tobytes()
    assert ref_st.mean() > 0.5
    ctx.close()


@pytest.mark.parametrize("w,h,seed", [(320, 200, 1), (157, 111, 2)])
def test_lk_parity_random_texture_and_flat(pkg, oracle, torch_cuda, w, h, seed):
    I = rand_image(h, w, seed)
    J = np.roll(I, (1, 2), (0, 1))
    J[:, :40] = 128                                        # flat band: minEig rejections
    ctx = _ctx(pkg, w, h)
    ctx.build_pyramid(0, I)
    ctx.build_pyramid(1, J)
    pts = _lk_points(w, h, 600, seed)
    ref_out, ref_st = oracle.lk_track(I, J, pts)
    out, st = ctx.lk_track(0, 1, pts)
    assert np.array_equal(st, ref_st)
    assert out.tobytes() == ref_out.tobytes()
    # reverse direction exercises the flat source window
    ref_out, ref_st = oracle.lk_track(J, I, pts)
    out, st = ctx.lk_track(1, 0, pts)
    assert np.array_equal(st, ref_st) and out.tobytes() == ref_out.tobytes()
    assert 0 < ref_st.sum() < len(ref_st)
    ctx.close()


def test_lk_empty_and_unbuilt(pkg, torch_cuda):
    ctx = _ctx(pkg, 416, 128)
    with pytest.raises(pkg.SvoError):
        ctx.lk_track(0, 1, np.zeros((4, 2), np.float32))      # slots not built
    img = rand_image(128, 416, 1)
    ctx.build_pyramid(0, img)
    ctx.build_pyramid(1, img)
    out, st = ctx.lk_track(0, 1, np.zeros((0, 2), np.float32))
    assert out.shape == (0, 2) and st.shape == (0,)
    ctx.close()


def test_circular_match_parity(pkg, oracle, torch_cuda, small_seq):
    """The fused 4-call loop + stable filter == four oracle LK calls + deleteBadmatchFeatures."""
    seq, frames = small_seq
    (L0, R0), (L1, R1) = frames[0], frames[1]
    h, w = L0.shape
    ctx = _ctx(pkg, w, h)
    for s, im in enumerate((L0, R0, L1, R1)):
        ctx.build_pyramid(s, im)
    kp = oracle.fast(L0)
    pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    pL0, pR0, pL1, pR1 = [oracle.PyramidHandle(x) for x in (L0, R0, L1, R1)]
    t1r, s1 = oracle.lk_track(pL0, pR0, pts)
    t2r, s2 = oracle.lk_track(pR0, pR1, t1r)
    t2l, s3 = oracle.lk_track(pR1, pL1, t2r)
    ret, s4 = oracle.lk_track(pL1, pL0, t2l)
    keep, m = oracle.circular_keep(pts, t1r, t2r, t2l, ret, s1, s2, s3, s4, 3.0)
    k = keep.astype(bool)
    got = ctx.circular_match((0, 1, 2, 3), pts)
    assert got[0].shape[0] == m and m > 50
    for g, r in zip(got, (pts[k], t1r[k], t2r[k], t2l[k])):
        assert g.tobytes() == r.tobytes()
    got_d = ctx.circular_match((0, 1, 2, 3), torch_cuda.from_numpy(pts).cuda())
    for g, r in zip(got_d, (pts[k], t1r[k], t2r[k], t2l[k])):
        assert g.cpu().numpy().tobytes() == r.tobytes()
    ctx.close()


def test_lk_parity_kitti_size(pkg, oracle, torch_cuda, synth):
    """BASELINE config size (1241x376): FAST + circular LK, bit-exact."""
    seq = synth.StereoSequence(width=1241, height=376, n_frames=2, seed=20200710, supersample=1)
    fr = [tuple(x.numpy() for x in seq.render(t)) for t in range(2)]
    (L0, R0), (L1, R1) = fr
    ctx = _ctx(pkg, 1241, 376)
    kp = ctx.fast_detect(L0)
    ref_kp = oracle.fast(L0)
    assert kp.tobytes() == ref_kp.tobytes() and len(kp) > 500
    for s, im in enumerate((L0, R0, L1, R1)):
        ctx.build_pyramid(s, im)
    pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
    prm = oracle.make_params(*seq.proj())
    res, _, _ = oracle.lk_track_step(prm, L0, R0, L1, R1, ref_kp, np.eye(4), want_tracks=True, threads=8)
    got = ctx.circular_match((0, 1, 2, 3), pts)
    assert got[0].shape[0] == res["n_tracked"] and res["n_tracked"] > 300
    for k in range(4):
        assert got[k].tobytes() == res["tracks"][k].tobytes()
    ctx.close()


def test_lk_parity_hd_stress_size(pkg, oracle, torch_cuda, synth):
    """BASELINE config #4 size (1920x1080, about 2000 features): FAST at a raised threshold, the
    2000 strongest corners (ties: row-major first), circular LK -- bit-exact."""
    w, h = 1920, 1080
    seq = synth.StereoSequence(width=w, height=h, n_frames=2, seed=1, supersample=1)
    fr = [tuple(x.numpy() for x in seq.render(t)) for t in range(2)]
    (L0, R0), (L1, R1) = fr
    ctx = _ctx(pkg, w, h, max_keypoints=1 << 16)
    kp = ctx.fast_detect(L0, threshold=20)
    ref_kp = oracle.fast(L0, thr=20)
    assert kp.tobytes() == ref_kp.tobytes() and len(kp) >= 2000
    order = np.argsort(-kp["response"], kind="stable")[:2000]
    sel = ref_kp[np.sort(order)]
    for s, im in enumerate((L0, R0, L1, R1)):
        ctx.build_pyramid(s, im)
    pts = np.stack([sel["x"], sel["y"]], 1).astype(np.float32)
    prm = oracle.make_params(*seq.proj())
    res, _, _ = oracle.lk_track_step(prm, L0, R0, L1, R1, sel, np.eye(4), want_tracks=True, threads=8)
    got = ctx.circular_match((0, 1, 2, 3), pts)
    assert got[0].shape[0] == res["n_tracked"] and res["n_tracked"] > 300
    for k in range(4):
        assert got[k].tobytes() == res["tracks"][k].tobytes()
    ctx.close()

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

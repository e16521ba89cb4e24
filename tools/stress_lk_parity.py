#!/usr/bin/env python3
"""Randomised differential test of the LK kernel against the oracle on adversarial inputs: binary
0/255 block textures (largest possible Scharr responses and mismatches: the integer partial sums
run closest to their 32-bit limits), sub-pixel start points everywhere including outside the
image, large displacements (many iterations, tile restaging), flat regions (min-eigenvalue
rejections).  Every output point and status byte must be bit-identical.
Usage: python tools/stress_lk_parity.py [n_seeds=12] [exact|sse2]     (sse2: svo_config.lk_accum = SSE2 against the
oracle's accumulation mode 2 -- float sums in an x86 OpenCV's lane order, where the pair sums of a binary texture exceed
2^24 and the conversions round)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def texture(rng, h, w, kind):
    if kind == 0:                                   # binary blocks of 1..4 px
        b = int(rng.integers(1, 5))
        t = rng.integers(0, 2, ((h + b - 1) // b, (w + b - 1) // b)).astype(np.uint8) * 255
        return np.kron(t, np.ones((b, b), np.uint8))[:h, :w]
    if kind == 1:                                   # full-range noise
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind == 2:                                   # stripes (aperture problem: oscillating iterations)
        x = np.arange(w)[None, :] + np.zeros((h, 1), int)
        return (((x // int(rng.integers(2, 7))) % 2) * 255).astype(np.uint8)
    t = rng.integers(0, 256, (h // 8 + 1, w // 8 + 1)).astype(np.uint8)      # smooth blocks + flat band
    img = np.kron(t, np.ones((8, 8), np.uint8))[:h, :w].copy()
    img[:, : w // 4] = 128
    return img


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sse2 = len(sys.argv) > 2 and sys.argv[2] == "sse2"
    pkg = entry.load_package()
    O = entry.load_oracle()
    O.build()
    O.set_lk_accum(O.LK_ACCUM_FLOAT_SSE if sse2 else O.LK_ACCUM_EXACT)
    ctx_kw = dict(lk_accum=pkg.LK_ACCUM_SSE2) if sse2 else {}
    bad = 0
    for seed in range(n_seeds):
        rng = np.random.default_rng(1000 + seed)
        w, h = int(rng.integers(80, 400)), int(rng.integers(60, 300))
        I = texture(rng, h, w, seed % 4)
        dx, dy = int(rng.integers(-9, 10)), int(rng.integers(-5, 6))
        J = np.roll(I, (dy, dx), (0, 1))
        if seed % 3 == 0:
            J = np.clip(J.astype(int) + rng.integers(-40, 41, J.shape), 0, 255).astype(np.uint8)
        n = 3000
        pts = np.stack([rng.uniform(-15, w + 15, n), rng.uniform(-15, h + 15, n)], 1).astype(np.float32)
        pts[::7] = np.round(pts[::7])               # exact integers
        pts[1::7] = np.floor(pts[1::7]) + 0.5        # exact halves
        ctx = pkg.Context(w, h, device=0, **ctx_kw)
        ctx.build_pyramid(0, I)
        ctx.build_pyramid(1, J)
        for a, b_, A, B in ((0, 1, I, J), (1, 0, J, I)):
            ref_out, ref_st = O.lk_track(A, B, pts, threads=8)
            out, st = ctx.lk_track(a, b_, pts)
            same = np.array_equal(st, ref_st) and out.tobytes() == ref_out.tobytes()
            if not same:
                bad += 1
                k = np.flatnonzero((st != ref_st) | (out != ref_out).any(1))
                print(f"MISMATCH seed {seed} {w}x{h} kind {seed % 4} dir {a}->{b_}: {len(k)} points, first {k[:5]}", flush=True)
        ctx.close()
        print(f"seed {seed}: {w}x{h} kind {seed % 4} shift ({dx},{dy}) tracked {int(ref_st.sum())}/{n}", flush=True)
    print("stress result:", "OK" if bad == 0 else f"{bad} mismatching runs")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

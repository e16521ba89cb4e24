#!/usr/bin/env python3
"""ORB extraction on full-size (1241x376) adversarial textures against the oracle: binary / noise / stripe images, noise on
rendered frames, low-contrast images (minTh retries), through the default configuration (8 levels, 2000 features).
Usage: python tools/stress_orb_fullsize.py [n_seeds=12]"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_lk_parity import texture  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    pkg = entry.load_package()
    O = entry.load_oracle()
    O.build()
    synth = importlib.import_module(entry.PKG_NAME + ".synth")
    seq = synth.StereoSequence(width=1241, height=376, n_frames=2, seed=3, supersample=1)
    base = seq.render(1)[0].numpy().astype(np.int32)
    ctx = pkg.Context(1241, 376, device=0, track_mode=pkg.MODE_ORB)
    bad = refused = 0
    for seed in range(n):
        rng = np.random.default_rng(900 + seed)
        kind = seed % 6
        if kind < 4:
            img = texture(rng, 376, 1241, kind)
        elif kind == 4:
            img = np.clip(base + rng.integers(-20, 21, base.shape), 0, 255).astype(np.uint8)
        else:
            img = (110 + rng.integers(0, 14, (94, 311))).astype(np.uint8).repeat(4, 0).repeat(4, 1)[:376, :1241]
        img = np.ascontiguousarray(img)
        try:
            k, d, per = ctx.orb_extract(img)
        except pkg.SvoError as e:
            refused += 1
            print("seed", seed, "kind", kind, "refused:", str(e)[:80], flush=True)
            continue
        rk, rd, rper = O.orb_extract(img)
        same = k.tobytes() == rk.tobytes() and d.tobytes() == rd.tobytes()
        n0 = len(ctx.orb_read_candidates(0, cap=16384))
        print(f"seed {seed} kind {kind}: {len(rk)} keypoints, level-0 candidates {n0}, {'ok' if same else 'MISMATCH'}", flush=True)
        bad += not same
    print("orb full-size stress:", "OK" if bad == 0 else f"{bad} MISMATCHES", f"({refused} refused)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

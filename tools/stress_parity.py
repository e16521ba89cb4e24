#!/usr/bin/env python3
"""Randomised differential tests of the non-LK stages against the oracle on adversarial inputs.
  fast   binary / noise / stripe textures, thresholds 1..200, with and without NMS
  orb    the same textures through the whole extractor (different level counts / scale factors)
  tri    triangulation with zero, negative and huge disparities, points at the principal point
  step   whole frame steps (both modes) on frames without any stereo geometry: every stage decision
         (too few tracks, inlier ratio, gates, pose) must match
  pnp    RANSAC-EPnP + LM on coplanar scenes, exactly 5 / 6 points, duplicated points, 90 % outliers,
         points behind the camera
Integer outputs must be bit-identical; poses within 1e-9 relative.
Usage: python tools/stress_parity.py [n_seeds=8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_lk_parity import texture  # noqa: E402


# Poses: north_star asks for 1e-4 relative Frobenius.  Well-posed scenes agree to <= 1e-9; these
# scenes are ill-conditioned on purpose (no real geometry), which amplifies the last-ulp differences
# of the wave-parallel LM sums to a few 1e-9, so the stress bar is 1e-6.
POSE_TOL = 1e-6


def relerr(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def main():
    n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    pkg = entry.load_package()
    O = entry.load_oracle()
    O.build()
    bad = []
    for seed in range(n_seeds):
        rng = np.random.default_rng(500 + seed)
        w, h = int(rng.integers(70, 420)), int(rng.integers(64, 300))
        img = texture(rng, h, w, seed % 4)
        # ---- FAST
        ctx = pkg.Context(w, h, device=0, max_keypoints=1 << 17)
        for thr in (1, 7, 20, 60, 200):
            for nms in (True, False):
                if ctx.fast_detect(img, threshold=thr, nonmax=nms, cap=1 << 17).tobytes() != O.fast(img, thr=thr, nms=nms).tobytes():
                    bad.append(("fast", seed, thr, nms))
        ctx.close()
        # ---- ORB (sizes the extractor accepts: every level needs at least one 30-px cell)
        wo, ho = max(w, 160), max(h, 120)
        imo = texture(rng, ho, wo, (seed + 1) % 4)
        cfg = [dict(nlevels=8, scale_factor=1.2), dict(nlevels=3, scale_factor=1.4), dict(nlevels=1, scale_factor=1.2)][seed % 3]
        try:
            c = pkg.Context(wo, ho, device=0, track_mode=pkg.MODE_ORB, max_keypoints=16384, orb_nlevels=cfg["nlevels"],
                            orb_scale_factor=cfg["scale_factor"], orb_nfeatures=1000)
            k, d, _ = c.orb_extract(imo)
            rk, rd, _ = O.orb_extract(imo, nfeatures=1000, cap=16384, **cfg)
            if k.tobytes() != rk.tobytes() or d.tobytes() != rd.tobytes():
                bad.append(("orb", seed, cfg, len(k), len(rk)))
            c.close()
        except pkg.SvoError as e:                   # a loud capacity / geometry error is acceptable, a silent difference is not
            print("orb seed", seed, "refused:", str(e)[:90])
        # ---- triangulation
        P1 = np.array([[718.856, 0, 607.193, 0], [0, 718.856, 185.216, 0], [0, 0, 1, 0]])
        P2 = P1.copy(); P2[0, 3] = -386.1448
        n = 4000
        x1 = np.stack([rng.uniform(0, 1241, n), rng.uniform(0, 376, n)], 1).astype(np.float32)
        disp = rng.choice([0.0, -3.0, 1e-3, 0.5, 30.0, 600.0], n) + rng.uniform(-1e-4, 1e-4, n)
        x2 = (x1 - np.stack([disp, rng.uniform(-2, 2, n)], 1)).astype(np.float32)
        x1[0] = (607.193, 185.216); x2[0] = x1[0]
        c = pkg.Context(416, 128, device=0)
        got, ref = c.triangulate(P1, P2, x1, x2), O.triangulate(P1, P2, x1, x2)
        if not np.array_equal(np.nan_to_num(got, nan=7e33), np.nan_to_num(ref, nan=7e33)):
            bad.append(("tri", seed, int((got != ref).any(1).sum())))
        # ---- PnP
        K = P1[:, :3]
        scenes = []
        X = np.stack([rng.uniform(-6, 6, 300), rng.uniform(-2, 2, 300), rng.uniform(5, 30, 300)], 1)
        scenes.append(("generic+90%outliers", X, 0.9))
        Xp = X.copy(); Xp[:, 2] = 12.0
        scenes.append(("coplanar", Xp, 0.2))
        scenes.append(("five", X[:5], 0.0))
        scenes.append(("six", X[:6], 0.0))
        Xd = np.repeat(X[:40], 5, 0)
        scenes.append(("duplicates", Xd, 0.3))
        Xb = X.copy(); Xb[::3, 2] *= -1
        scenes.append(("behind", Xb, 0.3))
        ang = 0.03
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        t = np.array([0.1, -0.05, -0.8])
        for name, Xs, out_frac in scenes:
            Xc = (R @ Xs.T).T + t
            u = (K @ Xc.T).T
            with np.errstate(divide="ignore", invalid="ignore"):
                u = u[:, :2] / u[:, 2:]
            u = np.nan_to_num(u, nan=0.0, posinf=1e6, neginf=-1e6)
            k_out = int(out_frac * len(Xs))
            if k_out:
                idx = rng.choice(len(Xs), k_out, replace=False)
                u[idx] += rng.uniform(-80, 80, (k_out, 2))
            g = c.pnp_ransac(Xs.astype(np.float32), u.astype(np.float32), K)
            r = O.pnp_ransac(Xs.astype(np.float32), u.astype(np.float32), K)
            same = (g["ok"] == r["ok"] and g["n_inliers"] == r["n_inliers"] and g["ransac_iters"] == r["ransac_iters"] and
                    g["best_iter"] == r["best_iter"] and np.array_equal(g["mask"], r["mask"]))
            if same and r["ok"]:
                same = relerr(g["R"], r["R"]) < 1e-9 and relerr(g["tvec"], r["tvec"]) < 1e-7
            if not same:
                bad.append(("pnp", seed, name, g["n_inliers"], r["n_inliers"], g["ransac_iters"], r["ransac_iters"]))
        c.close()
        # ---- whole frame steps on frames with no stereo geometry at all (shifted / noisy textures):
        # whatever the oracle decides at each stage (too few tracks, inlier ratio, gates, or a pose),
        # the fused pipeline must decide the same, in both modes
        ws, hs = max(w, 200), max(h, 140)
        base = texture(rng, hs, ws, (seed + 2) % 4)
        fr = []
        for k in range(3):
            L = np.roll(base, (int(rng.integers(-2, 3)), 3 * k), (0, 1))
            R = np.roll(L, (int(rng.integers(-1, 2)), -int(rng.integers(2, 12))), (0, 1))
            if seed % 2:
                R = np.clip(R.astype(int) + rng.integers(-6, 7, R.shape), 0, 255).astype(np.uint8)
            fr.append((np.ascontiguousarray(L), np.ascontiguousarray(R)))
        prm = O.make_params(P1, P2)
        cl = pkg.Context(ws, hs, device=0, max_keypoints=1 << 16, P1=P1, P2=P2)
        kps, pose = O.fast(fr[0][0]), np.eye(4)
        cl.add_frame(*fr[0])
        for k in (1, 2):
            r, kps, pose = O.lk_track_step(prm, *fr[k - 1], *fr[k], kps, pose, threads=8)
            rc, g = cl.add_frame(*fr[k])
            ok = (int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"] and int(g["n_tracked"]) == r["n_tracked"] and
                  int(g["n_inliers"]) == r["n_inliers"] and relerr(cl.get_pose(), pose) < POSE_TOL)
            if not ok:
                bad.append(("lk-step", seed, k, int(g["fail_stage"]), r["fail_stage"], int(g["n_tracked"]), r["n_tracked"]))
        cl.close()
        try:
            co = pkg.Context(ws, hs, device=0, track_mode=pkg.MODE_ORB, max_keypoints=16384, P1=P1, P2=P2,
                             min_move2=0.05 ** 2, max_move2=10.0 ** 2)
            prm_o = O.make_params(P1, P2, min_t2=0.05 ** 2, max_t2=10.0 ** 2)
            feats = [(O.orb_extract(L, cap=16384)[:2], O.orb_extract(R, cap=16384)[:2]) for L, R in fr]
            pose = np.eye(4)
            co.add_frame(*fr[0])
            for k in (1, 2):
                (kL, dL), (kR, dR) = feats[k - 1]
                (k2, d2), _ = feats[k]
                r, pose = O.orb_track_step(prm_o, kL, dL, kR, dR, k2, d2, pose)
                rc, g = co.add_frame(*fr[k])
                ok = (int(g["ok"]) == r["ok"] and int(g["fail_stage"]) == r["fail_stage"] and int(g["n_tracked"]) == r["n_tracked"] and
                      int(g["n_inliers"]) == r["n_inliers"] and relerr(co.get_pose(), pose) < POSE_TOL)
                if not ok:
                    bad.append(("orb-step", seed, k, int(g["fail_stage"]), r["fail_stage"], int(g["n_tracked"]), r["n_tracked"],
                                int(g["n_inliers"]), r["n_inliers"], int(g["ransac_iters"]), relerr(co.get_pose(), pose),
                                relerr(g["tvec"], r["tvec"]), int(g["lm_iters"])))
            co.close()
        except pkg.SvoError as e:
            print("orb-step seed", seed, "refused:", str(e)[:90])
        print(f"seed {seed}: {w}x{h} done, {len(bad)} mismatches so far", flush=True)
    for b in bad:
        print("MISMATCH", b)
    print("stress result:", "OK" if not bad else f"{len(bad)} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    sys.exit(main())

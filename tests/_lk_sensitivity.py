"""Helper of the LK accumulation-order sensitivity tests (CPU oracle only; test infrastructure).

Runs Tracking::LK_StereoF2F_PnP_Track (reference src/tracking.cpp:258-344) stage by stage with the oracle
under each accumulation order of the LK sums (oracle/lk.c: exact int64 = canonical choice C0, upstream's
float accumulation in raster order, and in the lane order of upstream's SSE2 block) and counts what moves."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

MODES = {"exact_int64 (C0, the parity target)": 0, "float, raster order": 1, "float, SSE2 lane order": 2}


def relfro(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


def _pair(O, P1, P2, prev, cur, fast_thr=20):
    """One frame pair: the four LK calls (each call's output, failed points included, feeds the next one:
    src/tracking.cpp:583-622), deleteBadmatchFeatures (:623-660), triangulation, RANSAC-PnP, gates."""
    kps = O.fast(prev[0], thr=fast_thr)
    t1l = np.stack([kps["x"], kps["y"]], 1).astype(np.float32)
    pyr = [O.PyramidHandle(im) for im in (prev[0], prev[1], cur[1], cur[0])]      # L1, R1, R2, L2
    t1r, s1 = O.lk_track(pyr[0], pyr[1], t1l)
    t2r, s2 = O.lk_track(pyr[1], pyr[2], t1r)
    t2l, s3 = O.lk_track(pyr[2], pyr[3], t2r)
    ret, s4 = O.lk_track(pyr[3], pyr[0], t2l)
    keep, m = O.circular_keep(t1l, t1r, t2r, t2l, ret, s1, s2, s3, s4)
    k = keep.astype(bool)
    X = O.triangulate(P1, P2, t1l[k], t1r[k])
    K = np.asarray(P1, np.float64).reshape(3, 4)[:, :3].copy()
    pnp = O.pnp_ransac(X, t2l[k], K)
    rc, _, Tinv = O.gate_and_accumulate(pnp["R"], pnp["tvec"], np.eye(4))
    ok = int(rc >= 0 and m >= 5 and pnp["n_inliers"] / max(m, 1) >= 0.01)
    return dict(status=np.stack([s1, s2, s3, s4]), keep=keep, pts=np.stack([t1r, t2r, t2l, ret]), n_tracked=int(m),
                n_inliers=int(pnp["n_inliers"]), mask=pnp["mask"], ransac_iters=pnp["ransac_iters"], best_iter=pnp["best_iter"],
                T_rel_inv=Tinv if ok else np.eye(4), ok=ok)


def run_mode(O, seq, frames, mode, workers):
    P1, P2 = seq.proj()
    old = O.set_lk_accum(mode)
    try:
        with ThreadPoolExecutor(max_workers=workers) as ex:          # ctypes releases the GIL; pairs are independent
            out = list(ex.map(lambda t: _pair(O, P1, P2, frames[t - 1], frames[t]), range(1, len(frames))))
    finally:
        O.set_lk_accum(old)
    return out


def chain(recs):
    P, out = np.eye(4), []
    for r in recs:
        if r["ok"]:
            P = P @ r["T_rel_inv"]
        out.append(P.copy())
    return out


def compare(base, other):
    """What differs between two runs of the same pairs (base = C0)."""
    d = dict(pairs=len(base), points=int(sum(b["keep"].size for b in base)), status_bytes=0, status_bytes_differ=0,
             keep_flags_differ=0, coords=0, coords_bit_identical=0, max_abs_coord_diff_px=0.0, pairs_n_tracked_differ=0,
             pairs_inlier_mask_differ=0, pairs_ok_differ=0)
    per_pair = []
    for b, o in zip(base, other):
        d["status_bytes"] += b["status"].size
        d["status_bytes_differ"] += int((b["status"] != o["status"]).sum())
        d["keep_flags_differ"] += int((b["keep"] != o["keep"]).sum())
        both = (b["keep"] & o["keep"]).astype(bool)
        pb, po = b["pts"][:, both], o["pts"][:, both]
        d["coords"] += pb.size
        d["coords_bit_identical"] += int((pb.view(np.uint32) == po.view(np.uint32)).sum())
        if pb.size:
            d["max_abs_coord_diff_px"] = max(d["max_abs_coord_diff_px"], float(np.abs(pb - po).max()))
        d["pairs_n_tracked_differ"] += int(b["n_tracked"] != o["n_tracked"])
        same_mask = b["mask"].shape == o["mask"].shape and np.array_equal(b["mask"], o["mask"])
        d["pairs_inlier_mask_differ"] += int(not same_mask)
        d["pairs_ok_differ"] += int(b["ok"] != o["ok"])
        per_pair.append(relfro(o["T_rel_inv"], b["T_rel_inv"]))
    cb, co = chain(base), chain(other)
    chained = [relfro(x, y) for x, y in zip(co, cb)]
    d["rel_motion_relfro_max"], d["rel_motion_relfro_median"] = float(np.max(per_pair)), float(np.median(per_pair))
    d["rel_motion_relfro_max_same_inlier_mask"] = float(max([p for p, b, o in zip(per_pair, base, other)
                                                              if b["mask"].shape == o["mask"].shape and np.array_equal(b["mask"], o["mask"])] or [0.0]))
    d["chained_pose_relfro_max"], d["chained_pose_relfro_last"] = float(np.max(chained)), float(chained[-1])
    return d

#!/usr/bin/env python3
"""Wave-level hit rate of an integer fast path in lk_sse2_kernel (CPU only; the oracle's exactness log, oracle/lk.c).

A float sum of integers is exact in ANY order while every partial sum stays below 2^24; an order-free sufficient test for
one chain is max(sum of positive terms, sum of |negative terms|) < 2^24.  When it holds for all ten b chains of every LIVE
slot of a wave-iteration, the wave can replace the serial float chains by integer reductions (the final six float adds are
replayed).  A wave tracks four consecutive points in lockstep (lk_sse2.hip), so what counts is the share of WAVE-iterations
whose active slots all pass, not the share of point-iterations.

usage: lk_guard_model.py [pairs=2] [accum=2|3|4]  -> one JSON object on stdout"""
import sys, importlib, json, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import __graft_entry__ as g

LV = 8          # ORC_LK_MAX_LEVELS


def collect(npairs, accum):
    g.load_package(); O = g.load_oracle()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    O.set_lk_accum(accum)
    lib = O.lib()
    lib.orc_lk_set_guard_log.argtypes = [C.c_void_p]
    seq = synth.StereoSequence(width=1241, height=376, n_frames=npairs + 1, seed=20200710)
    fr = [tuple(x.numpy() for x in seq.render(i)) for i in range(npairs + 1)]
    out = []
    for t in range(npairs):
        L0, R0 = fr[t]; L1, R1 = fr[t + 1]
        kp = O.fast(L0)
        pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
        P = [O.PyramidHandle(x) for x in (L0, R0, R1, L1)]
        n = len(pts)
        masks = np.zeros((n, 4, 4, 8), np.uint32)       # point, call, level, kind
        livec = np.zeros((n, 4), bool)
        cur = pts; live = np.ones(n, bool); rej = (pts[:, 0] < 0) | (pts[:, 1] < 0); prev_y = pts[:, 1].copy()
        for c, (a, b) in enumerate([(0, 1), (1, 2), (2, 3), (3, 0)]):
            log = np.zeros((n, LV, 8), np.uint32)
            lib.orc_lk_set_guard_log(log.ctypes.data_as(C.c_void_p))
            nxt, st = O.lk_track(P[a], P[b], cur, threads=1)
            lib.orc_lk_set_guard_log(None)
            masks[:, c] = log[:, :4]
            livec[:, c] = live
            rej = rej | (nxt[:, 0] < 0) | (nxt[:, 1] < 0) | (st == 0)
            if c in (0, 2): rej |= np.abs(prev_y - nxt[:, 1]) > 3.0
            prev_y = nxt[:, 1]; live = live & ~rej; cur = nxt
        out.append((masks, livec))
    O.set_lk_accum(0)
    return out


def rates(data, kind):
    pt_it = pt_ok = pt_eq = 0
    wave_it = wave_ok = 0
    slot_it = slot_ok_in_missed = 0
    by_level = {l: [0, 0, 0, 0] for l in range(4)}       # wave-iterations, hits, point-iterations, point hits
    for masks, livec in data:
        n = masks.shape[0]
        for w0 in range(0, n, 4):
            M = masks[w0:w0 + 4]; Lv = livec[w0:w0 + 4]
            for c in range(4):
                lv = Lv[:, c]
                if not lv.any(): break
                for l in range(4):
                    ran = M[lv, c, l, 3]; ok = M[lv, c, l, kind]; eq = M[lv, c, l, 2]
                    for j in range(32):
                        act = (ran >> j) & 1
                        na = int(act.sum())
                        if na == 0: break
                        good = int((((ok >> j) & 1) & act).sum())
                        pt_it += na; pt_ok += good; pt_eq += int((((eq >> j) & 1) & act).sum())
                        wave_it += 1; slot_it += na
                        hit = good == na
                        wave_ok += hit
                        by_level[l][0] += 1; by_level[l][1] += hit; by_level[l][2] += na; by_level[l][3] += good
    return {
        "point_iterations": pt_it,
        "guard_holds_point_iterations": round(pt_ok / pt_it, 4),
        "float_equals_exact_point_iterations": round(pt_eq / pt_it, 4),
        "wave_iterations": wave_it,
        "active_slots_per_wave_iteration": round(slot_it / wave_it, 3),
        "guard_holds_wave_iterations": round(wave_ok / wave_it, 4),
        "by_level": {str(l): {"wave_iterations": v[0], "wave_hit": round(v[1] / max(v[0], 1), 4),
                              "point_hit": round(v[3] / max(v[2], 1), 4)} for l, v in by_level.items()},
    }


if __name__ == "__main__":
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    accum = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    data = collect(npairs, accum)
    out = {"workload": f"S0 1241x376, first {npairs} pairs, four chained calls, FAST corners of the left image", "oracle_accum_mode": accum,
           "pair_order_guard (modes 2, 4)": rates(data, 0), "legacy_order_guard (mode 3)": rates(data, 1),
           "six_sums": rates(data, 4), "four_sums": rates(data, 5), "two_sums (b == lk_kernel's)": rates(data, 6),
           "pair_order_guard_packed_hi16": rates(data, 7)}
    if "--brief" in sys.argv:
        out = {k: ({"point_hit": v["guard_holds_point_iterations"], "wave_hit": v["guard_holds_wave_iterations"]} if isinstance(v, dict) else v) for k, v in out.items()}
    print(json.dumps(out, indent=1))

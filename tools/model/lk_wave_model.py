#!/usr/bin/env python3
"""Cost model of the LK kernels' wave grouping, fed with the oracle's iteration counts (CPU only).

A wave tracks four points ("slots") in lockstep: a level pass runs max(iterations of its live slots) wave-iterations.
This script replays the four chained calls of a few S0 pairs with the oracle, logs the iterations of every
(point, call, level), groups the points as the kernels do (four consecutive points a wave) and prices policies with
per-kernel instruction constants (from the ISA listings, tools/gpu/isa_count.py):
  lockstep     -- what lk_kernel / lk_sse2_kernel do
  decouple(K)  -- a slot that is the LAST one iterating at a level and has run more than K iterations lets the other
                  slots go on to the next level; it keeps iterating beside them and joins the next level set-up
usage: lk_wave_model.py [pairs=2] [accum=0|2]"""
import sys, importlib, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import __graft_entry__ as g

def collect(npairs, accum):
    pkg = g.load_package(); O = g.load_oracle()
    synth = importlib.import_module(g.PKG_NAME + ".synth")
    O.set_lk_accum(accum)
    seq = synth.StereoSequence(width=1241, height=376, n_frames=npairs + 1, seed=20200710)
    fr = [tuple(x.numpy() for x in seq.render(i)) for i in range(npairs + 1)]
    out = []
    for t in range(npairs):
        L0, R0 = fr[t]; L1, R1 = fr[t + 1]
        kp = O.fast(L0)
        pts = np.stack([kp["x"], kp["y"]], 1).astype(np.float32)
        P = [O.PyramidHandle(x) for x in (L0, R0, R1, L1)]
        n = len(pts)
        its = np.zeros((n, 4, 4), np.int32)       # point, call, level
        livec = np.zeros((n, 4), bool)
        cur = pts; live = np.ones(n, bool); rej = (pts[:, 0] < 0) | (pts[:, 1] < 0); prev_y = pts[:, 1].copy()
        for c, (a, b) in enumerate([(0, 1), (1, 2), (2, 3), (3, 0)]):
            log = np.full((n, 8), -1, np.int32)
            O.lib().orc_lk_set_iter_log(log.ctypes.data_as(C.c_void_p))
            nxt, st = O.lk_track(P[a], P[b], cur, threads=1)
            O.lib().orc_lk_set_iter_log(None)
            its[:, c, :] = np.maximum(log[:, :4], 0)
            livec[:, c] = live
            rej = rej | (nxt[:, 0] < 0) | (nxt[:, 1] < 0) | (st == 0)
            if c in (0, 2): rej |= np.abs(prev_y - nxt[:, 1]) > 3.0
            prev_y = nxt[:, 1]; live = live & ~rej; cur = nxt
        out.append((its, livec))
    O.set_lk_accum(0)
    return out

def price(data, K, LPF, LPS, ITF, ITS, policy):
    """instructions per wave summed over the data"""
    tot = 0.0; wave_it = 0; slot_it = 0; lp = 0
    for its, livec in data:
        n = its.shape[0]
        for w0 in range(0, n, 4):
            I = its[w0:w0 + 4]; Lv = livec[w0:w0 + 4]
            for c in range(4):
                lv = Lv[:, c]
                if not lv.any(): break
                if policy == "lockstep":
                    for l in (3, 2, 1, 0):
                        x = I[:, c, l][lv]
                        m = int(x.max()) if len(x) else 0
                        lp += 1
                        tot += LPF + LPS * lv.sum()
                        for j in range(m):
                            a = int((x > j).sum()); tot += ITF + ITS * a; wave_it += 1; slot_it += a
                else:
                    # event simulation: slot state = (level, remaining iterations); set-up events serve all waiting slots
                    ns = int(lv.sum()); x = I[:, c, :][lv]            # [slot][level]
                    level = [3] * ns; rem = [0] * ns; done_it = [0] * ns; waiting = [True] * ns; fin = [False] * ns
                    while not all(fin):
                        # set-up for every waiting slot
                        w = [s for s in range(ns) if waiting[s] and not fin[s]]
                        if w:
                            lp += 1; tot += LPF + LPS * len(w)
                            for s in w: rem[s] = int(x[s][level[s]]); done_it[s] = 0; waiting[s] = False
                        # iterate until the policy asks for a set-up
                        while True:
                            act = [s for s in range(ns) if not fin[s] and not waiting[s] and rem[s] > 0]
                            for s in range(ns):
                                if not fin[s] and not waiting[s] and rem[s] == 0:
                                    if level[s] == 0: fin[s] = True
                                    else: level[s] -= 1; waiting[s] = True
                            act = [s for s in range(ns) if not fin[s] and not waiting[s] and rem[s] > 0]
                            nwait = sum(1 for s in range(ns) if waiting[s] and not fin[s])
                            if not act: break
                            # leave for a set-up when somebody waits and the policy says so
                            if nwait and len(act) <= policy[1] and all(done_it[s] >= K for s in act): break
                            tot += ITF + ITS * len(act); wave_it += 1; slot_it += len(act)
                            for s in act: rem[s] -= 1; done_it[s] += 1
    return tot, wave_it, slot_it, lp

if __name__ == "__main__":
    npairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    accum = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    data = collect(npairs, accum)
    allit = np.concatenate([d[0][d[1]].reshape(-1) for d in data])
    print("point-levels", allit.size, "mean iterations", allit.mean(), "at cap 30:", (allit >= 30).mean())
    for name, (LPF, LPS, ITF, ITS) in {"exact": (350, 208, 92, 32), "sse2": (800, 190, 180, 56)}.items():
        base = None
        for pol in ["lockstep", ("d", 1), ("d", 2)]:
            for K in ([0] if pol == "lockstep" else [0, 4, 8, 12]):
                t, wi, si, lp = price(data, K, LPF, LPS, ITF, ITS, pol)
                if base is None: base = t
                print(f"{name:6s} {str(pol):14s} K={K:2d}: instr {t/1e6:8.2f} M ({t/base:5.3f})  wave-iterations {wi}  active slots/iter {si/max(wi,1):.2f}  level set-ups {lp}")

#!/usr/bin/env python3
"""Bank-conflict model of lk_sse2_kernel's chain reads (ds_read_b128), CPU only.

MI355X serves a ds_read_b128 in four groups of sixteen lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same
+ 32 -- on 64 banks of 4 B; lanes of one group that touch a bank with DIFFERENT addresses serialise, identical addresses
broadcast (/opt/skills/guides/MI355X_MICROARCH.md, LDS).  `check` prices the layout the kernel uses (csrc/lk_sse2.hip:
kChainOff, kTailX, kTailY, kStageDw and the chain-lane positions of make_lane); `search` is the random search over block
orders and slot strides that found it.
`legacy` does the same for the legacy order's staging (84-term lane chains: 88 words a chain, 22 reads + one feeder read).
usage: lds_chain_layout.py [search | legacy]"""
import itertools, random, sys

G = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G = G + [[l + 32 for l in g] for g in G]


def conflicts(addr):
    """extra LDS cycles of one ds_read_b128 whose lane l reads the 4 dwords at dword address addr[l]"""
    extra = 0
    for g in G:
        banks = {}
        for l in g:
            for k in range(4):
                banks.setdefault((addr[l] + k) % 64, set()).add(addr[l])
        extra += max(len(v) for v in banks.values()) - 1
    return extra


def phase1(off, S, xs):
    ys = tuple('c%d' % (int(c[1]) + 1) for c in xs)
    addr = [None] * 64
    for lane in range(64):
        s, p = lane >> 4, lane & 15
        f, h = p & 7, p >> 3
        fc = 0 if f == 0 else min(f, 4)            # idle positions 5..7 / 13..15 follow position 4 / 12
        a = off['TY' if h else 'TX'] if fc == 0 else off[(ys if h else xs)[fc - 1]]
        addr[lane] = s * S + a
    return conflicts(addr)


def phase2(off, S, pi=tuple(range(8))):
    tot = 0
    for which in (0, 1):
        ad = [0] * 64
        for lane in range(64):
            s, p = lane >> 4, lane & 15
            f, h = p & 7, p >> 3
            ad[lane] = s * S + off['TY' if h else 'TX'] + 44 + 4 * (pi[f] + 8 * which)
        tot += conflicts(ad)
    return tot


KERNEL = dict(TY=0, c0=108, c7=152, c1=196, c5=240, c3=284, c4=328, c2=372, TX=416, c6=524)
KERNEL_LEGACY = dict(c2=0, c0=88, TY=180, c5=288, c6=376, TX=468, c1=584, c4=676, c3=764, c7=860)     # kChainOffL / kTailXL / kTailYL, stride 960
ROUND5_LEGACY = dict(TY=0, TX=108, **{'c%d' % c: 216 + 88 * c for c in range(8)})                      # stride 928


def phase2_legacy(off, S):
    ad = [0] * 64
    for lane in range(64):
        s, p = lane >> 4, lane & 15
        ad[lane] = s * S + off['TY' if p >> 3 else 'TX'] + 88 + 4 * (p & 7)
    return conflicts(ad)


def legacy():
    xs = ('c0', 'c4', 'c2', 'c6')
    print("legacy, round-5 plain order, stride 928: extra cycles per chain read (22 of them)", phase1(ROUND5_LEGACY, 928, xs), "| feeder read", phase2_legacy(ROUND5_LEGACY, 928))
    print("legacy, kernel layout, stride 960:", phase1(KERNEL_LEGACY, 960, xs), "|", phase2_legacy(KERNEL_LEGACY, 960))
    ents = ['c%d' % c for c in range(8)] + ['TX', 'TY']
    size = {e: 88 for e in ents}; size['TX'] = size['TY'] = 108
    random.seed(2)
    best = []
    for _ in range(2000):
        perm = random.sample(ents, 10)
        off, o = {}, 0
        for b in perm:
            off[b] = o; o += size[b] + random.choice([0, 0, 4, 8, 12])
        for S in range((o + 3) // 4 * 4, (o + 3) // 4 * 4 + 72, 4):
            if phase1(off, S, xs) == 0: best.append((phase2_legacy(off, S), S, tuple((b, off[b]) for b in perm)))
    best.sort(key=lambda r: (r[0], r[1]))
    for b in best[:5]: print(b)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "legacy":
        legacy()
        sys.exit(0)
    if len(sys.argv) < 2:
        xs = ('c0', 'c4', 'c2', 'c6')
        print("kernel layout, slot stride 576: phase 1 extra cycles per read", phase1(KERNEL, 576, xs),
              "| phase 2 extra cycles (two reads)", phase2(KERNEL, 576))
        print("round-4-style layout (chains at 44 c, tails at 356 / 464, stride 572):",
              phase1(dict(**{'c%d' % c: 44 * c for c in range(8)}, TX=356, TY=464), 572, xs), "|",
              phase2(dict(TX=356, TY=464), 572))
        sys.exit(0)
    ents = ['c%d' % c for c in range(8)] + ['TX', 'TY']
    size = {e: 44 for e in ents}; size['TX'] = size['TY'] = 108
    random.seed(1)
    best = []
    for perm in [ents] + [random.sample(ents, 10) for _ in range(3000)]:
        off, o = {}, 0
        for b in perm:
            off[b] = o; o += size[b]
        for S in range(568, 608, 4):
            for xs in (('c0', 'c4', 'c2', 'c6'), ('c4', 'c0', 'c2', 'c6'), ('c0', 'c4', 'c6', 'c2'), ('c2', 'c6', 'c0', 'c4')):
                if phase1(off, S, xs): continue
                best.append((phase2(off, S), S, tuple(perm), xs))
    best.sort(key=lambda r: (r[0], r[1]))
    for b in best[:8]: print(b)

#!/usr/bin/env python3
"""LDS-array cycles of every LDS access of one k_rows block (M2 = 1200, [12,10,10], 256 threads), by phase,
with the bank model of MI355X_MICROARCH.md (see lds_conflicts.py).  Slots are float4 (16 B); addresses in dwords.
usage: lds_rows_model.py [pad]   pad = slots inserted per `pad` elements (0 = none)"""
import sys
RG = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RG = RG + [[l + 32 for l in g] for g in RG]
WG128 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]
WG64 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]

def cyc(addrs, groups, nbanks, width):
    """addrs: per lane first dword address or None; width dwords per lane"""
    tot = 0
    for g in groups:
        per = {}
        for l in g:
            a = addrs[l]
            if a is None: continue
            for d in range(width):
                per.setdefault((a + d) % nbanks, set()).add(a + d)
        tot += max([len(v) for v in per.values()], default=0)
    return tot

def rd128(slots): return cyc([None if s is None else 4 * s for s in slots], RG, 64, 4), 4
def wr128(slots): return cyc([None if s is None else 4 * s for s in slots], WG128, 32, 4), 8
def wr64(dw): return cyc(dw, WG64, 32, 2), 4

M2, NT = 1200, 256
PAD = int(sys.argv[1]) if len(sys.argv) > 1 else 0
def A(e): return e + (e // PAD if PAD else 0)
GS = A(M2 - 1) + 1           # group stride (B region)
def B(e): return GS + A(e)

phases = {}
def add(name, c, ideal):
    p = phases.setdefault(name, [0, 0, 0]); p[0] += c; p[1] += ideal; p[2] += 1

for w0 in range(0, NT, 64):
    lanes = range(w0, w0 + 64)
    # fill (wide): q = t + NT*i < 600 -> slots 2q, 2q+1 of A and B
    for i in range(3):
        for off in (0, 1):
            for reg in (A, B):
                sl = [reg(2 * (t + NT * i) + off) if t + NT * i < 600 else None for t in lanes]
                add('fill w128', *wr128(sl))
    # forward stages, 2 groups
    ns = M2
    for si, R in enumerate((12, 10, 10)):
        q = ns // R; nbf = M2 // R
        items = []
        for t in lanes:
            if t >= 2 * nbf: items.append(None); continue
            g, bf = t // nbf, t % nbf
            b, j = bf // q, bf % q
            items.append((g, b * ns + j))
        for T in range(R):
            sl = [None if x is None else (A if x[0] == 0 else B)(x[1] + T * q) for x in items]
            add('fwd%d r128' % si, *rd128(sl)); add('fwd%d w128' % si, *wr128(sl))
        ns //= R
    # combine: reads A[s], B[M2-1-s]; writes b64 C2[2s] (member 0), C2[2(M2-1-s)+1] (member 1)
    for i in range(5):
        s = [t + NT * i if t + NT * i < M2 else None for t in lanes]
        add('comb r128', *rd128([None if x is None else A(x) for x in s]))
        add('comb r128', *rd128([None if x is None else B(M2 - 1 - x) for x in s]))
        add('comb w64', *wr64([None if x is None else 4 * A(x) for x in s]))
        add('comb w64', *wr64([None if x is None else 4 * A(M2 - 1 - x) + 2 for x in s]))
    # inverse stages, 1 group (A region), run backwards
    for si, R, ns in ((2, 10, 10), (1, 10, 100), (0, 12, 1200)):
        q = ns // R; nbf = M2 // R
        items = []
        for t in lanes:
            if t >= nbf: items.append(None); continue
            b, j = t // q, t % q
            items.append(b * ns + j)
        for T in range(R):
            sl = [None if x is None else A(x + T * q) for x in items]
            add('inv%d r128' % si, *rd128(sl)); add('inv%d w128' % si, *wr128(sl))
    # final reads: slots 2q, 2q+1
    for i in range(3):
        for off in (0, 1):
            sl = [A(2 * (t + NT * i) + off) if t + NT * i < 600 else None for t in lanes]
            add('final r128', *rd128(sl))
tot = ide = 0
for k, (c, i, n) in phases.items():
    xfer = 13 * n if 'w128' in k else (6 * n if 'w64' in k else 0)
    print("%-12s instr %4d  array cycles %5d  (conflict-free %5d)  store transfer floor %5d" % (k, n, c, i, xfer))
    tot += max(c, xfer); ide += max(i, xfer)
print("block total (max of array cycles and store transfer per class): %d ; conflict-free %d ; LDS slots per block %d (limit 2560 for 4 blocks/CU)" % (tot, ide, 2 * GS))
ca = sum(c for c, i, n in phases.values()); ia = sum(i for c, i, n in phases.values())
print("array cycles %d, of which conflicts %d = %.1f %%" % (ca, ca - ia, 100.0 * (ca - ia) / ca))

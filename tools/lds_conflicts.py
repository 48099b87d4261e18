#!/usr/bin/env python3
"""Bank-conflict model of the in-LDS transform stages (MI355X_MICROARCH.md, LDS table):
ds_read_b128: 4 lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32), 64 banks of 4 B;
ds_write_b128: 8 groups of 8 contiguous lanes, 32 banks.  A group takes as many LDS cycles as the
largest number of DISTINCT addresses that fall on one bank.  Reports cycles per wave-instruction
(ideal: 4 read, 8 write) for every stage of a schedule under a slot swizzle.
usage: lds_conflicts.py rows|cols N-point radix,radix,... [swizzle]"""
import sys, itertools

RGROUPS = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27], [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
RGROUPS = RGROUPS + [[l + 32 for l in g] for g in RGROUPS]
WGROUPS = [list(range(8 * i, 8 * i + 8)) for i in range(8)]

def cycles(slots, groups, nbanks_slots):
    tot = 0
    for g in groups:
        per = {}
        for l in g:
            s = slots[l]
            if s is None: continue
            per.setdefault(s % nbanks_slots, set()).add(s)
        tot += max([len(v) for v in per.values()], default=0) if per else 0
    return tot

def stage_items(kind, n, radices, threads, ngroups):
    """yield (stage, R, q, list over waves of list over lanes of (group, base_elem) or None)"""
    ns = n
    for i, R in enumerate(radices):
        q = ns // R; nbf = n // R
        total = ngroups * nbf
        waves = []
        for w0 in range(0, min(total, threads), 64):   # first pass of the thread loop is representative
            lanes = []
            for l in range(64):
                w = w0 + l
                if w >= total: lanes.append(None); continue
                if kind == 'cols': g, bf = w % ngroups, w // ngroups
                else: g, bf = w // nbf, w % nbf
                b, j = bf // q, bf % q
                lanes.append((g, b * ns + j))
            waves.append(lanes)
        yield i, R, q, waves
        ns //= R

def evaluate(kind, n, radices, swz, verbose=True):
    if kind == 'cols': ngroups, threads, es, gs = 4, 512, 4, 1
    else: ngroups, threads, es, gs = 2, 256, 1, n
    rd = wr = ideal_r = ideal_w = 0
    for i, R, q, waves in stage_items(kind, n, radices, threads, ngroups):
        srd = swr = cnt = 0
        for lanes in waves:
            for T in range(R):
                slots = [None if x is None else swz(x[0] * gs + (x[1] + T * q) * es) for x in lanes]
                srd += cycles(slots, RGROUPS, 16); swr += cycles(slots, WGROUPS, 8); cnt += 1
        if verbose: print("  stage %d radix %2d q %4d: read %.2f cycles/instr (ideal 4), write %.2f (ideal 8; transfer floor 13)" % (i, R, q, srd / cnt, swr / cnt))
        rd += srd; wr += max(swr, 13 * cnt); ideal_r += 4 * cnt; ideal_w += 13 * cnt
    if verbose: print("  total LDS cycles per block pass: read %d (ideal %d), write %d (ideal %d)" % (rd, ideal_r, wr, ideal_w))
    return rd + wr

SWZ = {
    'none': lambda e: e,
    'rot1': lambda e: (e & ~15) | ((e + (e >> 4)) & 15),
    'rot3': lambda e: (e & ~15) | ((e + 3 * (e >> 4)) & 15),
    'rot5': lambda e: (e & ~15) | ((e + 5 * (e >> 4)) & 15),
    'rot7': lambda e: (e & ~15) | ((e + 7 * (e >> 4)) & 15),
    'xor': lambda e: e ^ ((e >> 4) & 15),
    'xor8': lambda e: e ^ ((e >> 3) & 7),
}
if __name__ == '__main__':
    kind, n = sys.argv[1], int(sys.argv[2])
    radices = [int(x) for x in sys.argv[3].split(',')]
    names = sys.argv[4:] or list(SWZ)
    for nm in names:
        print(kind, n, radices, 'swizzle', nm)
        evaluate(kind, n, radices, SWZ[nm])

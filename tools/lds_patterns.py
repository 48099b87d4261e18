#!/usr/bin/env python3
"""Writes the per-lane LDS address tables that tools/micro/lds_pattern.hip replays, one dispatch per
pattern, and prints the bank model's prediction (tools/lds_rows_model.py conventions) next to each name.
usage: lds_patterns.py out.bin [pad-spec]     then, on the GPU box:
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --output-format csv -d out -- ./lds_pattern out.bin
and tools/lds_patterns_report.py out.bin.names <counter csv> to tabulate model vs measured."""
import struct, sys
NT, M2 = 256, 1200
R128, W128, W64, R64 = 0, 1, 2, 3

def rows_patterns(A, B, tag=""):
    """A(e), B(e): slot index (16-byte units) of element e of group A / B.  Returns [(name, kind, [[byte addr or -1]*NT]*ninstr)]"""
    P = []
    def slots_to(name, kind, instrs):  # instrs: list of per-thread slot lists (None = inactive)
        P.append((tag + name, kind, [[-1 if s is None else 16 * s for s in ins] for ins in instrs]))
    T = range(NT)
    # fill (wide): q = t + NT*i < 600 -> slots 2q, 2q+1 of A and B
    ins = []
    for i in range(3):
        for off in (0, 1):
            for reg in (A, B):
                ins.append([reg(2 * (t + NT * i) + off) if t + NT * i < 600 else None for t in T])
    slots_to('fill_w128', W128, ins)
    ns = M2
    for si, R in enumerate((12, 10, 10)):
        q = ns // R; nbf = M2 // R
        items = []
        for t in T:
            if t >= 2 * nbf: items.append(None); continue
            g, bf = t // nbf, t % nbf
            b, j = bf // q, bf % q
            items.append((g, b * ns + j))
        ins = [[None if x is None else (A if x[0] == 0 else B)(x[1] + u * q) for x in items] for u in range(R)]
        slots_to('fwd%d_r128' % si, R128, ins); slots_to('fwd%d_w128' % si, W128, ins)
        ns //= R
    ins = []; insw = []
    for i in range(5):
        s = [t + NT * i if t + NT * i < M2 else None for t in T]
        ins.append([None if x is None else A(x) for x in s])
        ins.append([None if x is None else B(M2 - 1 - x) for x in s])
        insw.append([-1 if x is None else 16 * A(x) for x in s])
        insw.append([-1 if x is None else 16 * A(M2 - 1 - x) + 8 for x in s])
    slots_to('comb_r128', R128, ins)
    P.append((tag + 'comb_w64', W64, insw))
    for si, R, ns in ((2, 10, 10), (1, 10, 100), (0, 12, 1200)):
        q = ns // R; nbf = M2 // R
        items = []
        for t in T:
            if t >= nbf: items.append(None); continue
            b, j = t // q, t % q
            items.append(b * ns + j)
        ins = [[None if x is None else A(x + u * q) for x in items] for u in range(R)]
        slots_to('inv%d_r128' % si, R128, ins); slots_to('inv%d_w128' % si, W128, ins)
    ins = []
    for i in range(3):
        for off in (0, 1):
            ins.append([A(2 * (t + NT * i) + off) if t + NT * i < 600 else None for t in T])
    slots_to('final_r128', R128, ins)
    return P

def calib():
    P = []
    T = range(NT)
    for st in (1, 2, 3, 4, 5, 8, 10, 12, 16):
        P.append(('cal_r128_stride%d' % st, R128, [[16 * ((st * t) % 2400) for t in T]]))
        P.append(('cal_w128_stride%d' % st, W128, [[16 * ((st * t) % 2400) for t in T]]))
    for st in (1, 2, 4, 10):
        P.append(('cal_w64_stride%dx8B' % st, W64, [[8 * ((st * t) % 4800) for t in T]]))
        P.append(('cal_r64_stride%dx8B' % st, R64, [[8 * ((st * t) % 4800) for t in T]]))
    return P

def layouts(spec):
    if spec == 'none':
        return (lambda e: e), (lambda e: M2 + e)
    if spec.startswith('pad'):      # padN: one slot after every N elements
        n = int(spec[3:]); f = lambda e: e + e // n; gs = f(M2 - 1) + 1
        return f, (lambda e: gs + f(e))
    if spec.startswith('sub'):      # subS_T: sub-blocks of 100 at stride S, sub-sub-blocks of 10 at stride T
        S, T = [int(x) for x in spec[3:].split('_')]
        f = lambda e: (e // 100) * S + ((e % 100) // 10) * T + e % 10; gs = f(M2 - 1) + 1
        return f, (lambda e: gs + f(e))
    raise SystemExit('unknown layout ' + spec)

if __name__ == '__main__':
    out = sys.argv[1]
    specs = sys.argv[2:] or ['none']
    pats = calib()
    for sp in specs:
        A, B = layouts(sp)
        pats += rows_patterns(A, B, sp + ':')
    with open(out, 'wb') as f, open(out + '.names', 'w') as fn:
        f.write(struct.pack('i', len(pats)))
        for name, kind, instrs in pats:
            assert len(instrs) <= 12, name
            mx = max(max(r) for r in instrs) + 16
            f.write(struct.pack('4i', len(instrs), kind, max(mx, 65536), 0))
            for r in instrs: f.write(struct.pack('%di' % NT, *r))
            nact = sum(1 for r in instrs for w in range(0, NT, 64) if any(a >= 0 for a in r[w:w + 64]))
            fn.write('%s %d %d\n' % (name, kind, nact))
    print(len(pats), 'patterns')

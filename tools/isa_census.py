#!/usr/bin/env python3
"""usage: tools/isa_census.py <file.s> [kernel-name-substring]
Static instruction census of a kernel from hipcc -S output: VALU / LDS / VMEM / SALU counts between
consecutive s_barriers (the phases of the transform kernels) and the most frequent VALU opcodes.
Static counts: loops and branches are counted once."""
import collections, re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S+:', l) and want in l]
for st in starts:
    end = next(i for i in range(st, len(lines)) if 's_endpgm' in lines[i])
    seg, counts = 0, collections.defaultdict(collections.Counter)
    for l in lines[st + 1:end + 1]:
        l = l.strip()
        if not l or l[0] in ';.' or l.split()[0].endswith(':'):
            continue
        op = l.split()[0]
        if op == 's_barrier':
            seg += 1
            continue
        cat = ('valu' if op.startswith('v_') else 'lds' if op.startswith('ds_') else
               'vmem' if op.startswith(('global_', 'buffer_', 'flat_')) else 'salu' if op.startswith('s_') else 'other')
        counts[seg][cat] += 1
        if cat == 'valu':
            counts[seg]['v:' + re.sub(r'_e32|_e64|_dpp|_sdwa', '', op)] += 1
        if op.startswith('s_waitcnt'):
            counts[seg]['waitcnt'] += 1
    print(lines[st][:90])
    tot = collections.Counter()
    for s in sorted(counts):
        c = counts[s]
        tot.update(c)
        print(' seg', s, {k: c[k] for k in ('valu', 'lds', 'vmem', 'salu', 'waitcnt')},
              {k[2:]: v for k, v in c.most_common(12) if k.startswith('v:')})
    print(' total', {k: tot[k] for k in ('valu', 'lds', 'vmem', 'salu', 'waitcnt')})
    print(' ', {k[2:]: v for k, v in tot.most_common(40) if k.startswith('v:')})

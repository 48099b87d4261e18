#!/usr/bin/env python3
"""usage: lds_patterns_report.py out.bin.names counter_collection.csv  -> per pattern: LDS wave-instructions per
block iteration, array cycles and conflict cycles per wave-instruction (measured)."""
import csv, sys, collections
names = [l.split() for l in open(sys.argv[1])]
c = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[2])):
    if 'k_lds' in r['Kernel_Name']:
        c[int(r['Dispatch_Id'])][r['Counter_Name']] = c[int(r['Dispatch_Id'])].get(r['Counter_Name'], 0) + float(r['Counter_Value'])
ids = sorted(c)
ITERS, BLOCKS = 64, 256
tot = collections.defaultdict(lambda: [0.0, 0.0])
for (name, kind, nact), d in zip(names, ids):
    v = c[d]; n = v['SQ_INSTS_LDS'] / ITERS / BLOCKS
    act = v['SQ_LDS_IDX_ACTIVE'] / ITERS / BLOCKS; conf = v['SQ_LDS_BANK_CONFLICT'] / ITERS / BLOCKS
    print('%-28s instr/blk %5.1f  active cyc/blk %7.1f (%.2f per instr)  conflict cyc/blk %7.1f (%.2f per instr)' % (name, n, act, act / max(n, 1e-9), conf, conf / max(n, 1e-9)))
    if ':' in name:
        t = tot[name.split(':')[0]]; t[0] += act; t[1] += conf
for k, (a, cf) in tot.items():
    print('layout %-12s active %8.1f  conflicts %8.1f  (%.1f %%)' % (k, a, cf, 100 * cf / a))

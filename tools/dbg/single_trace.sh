#!/bin/bash
# usage (GPU box): tools/dbg/single_trace.sh <outdir-under-gpurun_out> [N] : kernel trace of single-pair calls -- per-kernel
# durations and the gaps between consecutive kernels of one call
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; N=${2:-1440000}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/dbg/single_pair_loop.py $N 200 > $O/plain.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/dbg/single_pair_loop.py $N 50 > $O/trace.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
rows = []
for f in glob.glob(sys.argv[1] + '/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), k))
rows.sort()
rows = [r for r in rows if r[2].startswith('k_')]
# the last 20 calls: a call starts at k_fwd_cols*
starts = [i for i, r in enumerate(rows) if r[2].startswith('k_fwd_cols')]
dur = collections.defaultdict(list); gap = collections.defaultdict(list); span = []
for a, b in zip(starts[-21:-1], starts[-20:]):
    call = rows[a:b]
    span.append(call[-1][1] - call[0][0])
    for i, (s, e, k) in enumerate(call):
        dur[(i, k[:44])].append(e - s)
        if i: gap[(i, k[:44])].append(s - call[i - 1][1])
med = lambda v: sorted(v)[len(v) // 2] / 1e3
for key in sorted(dur):
    print("%2d %-46s %7.1f us   gap before %6.1f us" % (key[0], key[1], med(dur[key]), med(gap[key]) if key in gap else 0.0))
print("first kernel start -> last kernel end: median %.1f us" % med(span))
PY
cat $O/plain.log

#!/bin/bash
# usage: tools/pmc_any.sh <outdir> "<counters>" [bench args]  -> per-kernel average of each counter
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; CTRS=$2; shift; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/p.log 2>&1
python3 - $OUT <<'PY'
import csv, collections, glob, sys
out=sys.argv[1]
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for f in glob.glob(out+'/p/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        tot[k][r['Counter_Name']]+=float(r['Counter_Value']); cnt[k][r['Counter_Name']]+=1
for k in tot:
    if k.startswith('k_') and ('cols' in k or 'rows' in k or 'pearson_partial' in k or 'synth' in k):
        print(k, {c: round(tot[k][c]/cnt[k][c]) for c in tot[k]})
PY
tail -3 $OUT/p.log | grep -i error

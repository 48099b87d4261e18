#!/bin/bash
# round 5, GPU call 4: spectral Pearson form after the fwd_cols / prep rework: parity, A/B, kernel trace
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5d; mkdir -p $O; cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "all rc=$?"; tail -5 $O/pytest.txt
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
run() { echo -n "$1: "; shift; env "$@" $B 2>/dev/null | python3 $R/tools/brief.py; }
{
for r in 1 2; do run spectral X=1; run direct ASX_PEARSON=direct; done
for n in 144000 480000 960000; do
  echo "== N=$n x 1024"
  echo -n "spectral: "; $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py; echo -n "direct: "; ASX_PEARSON=direct $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py
done
} > $O/ab.txt 2>&1
cat $O/ab.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 5 --precondition 30 --no-cpu --no-config4 --no-single > $O/trace.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
calls=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        calls[k].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(calls.items(), key=lambda kv:-sum(kv[1])):
    if k.startswith('k_'):
        v2=sorted(v); print("%-60s calls %4d median %9.1f us  min %9.1f" % (k[:60], len(v), v2[len(v2)//2]/1e3, v2[0]/1e3))
PY

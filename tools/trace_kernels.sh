#!/bin/bash
# usage (GPU box): tools/trace_kernels.sh <outdir-under-gpurun_out> [bench args...]
# rocprofv3 --kernel-trace of one bench.py run: median / minimum duration of EVERY kernel of the library (the small tail kernels --
# finalize, refine, Pearson prep / partial / final -- that bench.py's HIP events lump into two families)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 5 --precondition 30 --no-cpu --no-config4 --no-single "$@" > $O/trace.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections
calls = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        calls[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(calls.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith('k_'):
        v2 = sorted(v)
        print("%-60s calls %4d median %9.1f us  min %9.1f" % (k[:60], len(v), v2[len(v2) // 2] / 1e3, v2[0] / 1e3))
PY

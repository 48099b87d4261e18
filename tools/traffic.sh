#!/bin/bash
# usage (on the GPU box, via gpurun): tools/traffic.sh <outdir-under-gpurun_out> [bench args...]
# HBM traffic per kernel launch from the PMC counters, collected as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots: 3 + 2),
# FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), units KiB.
# Kernel durations: rocprofv3 --kernel-trace of a run whose timed steps lie behind >= 30 untimed ones; only the calls of
# the pre-conditioned timed region are averaged (the chip's clock ramp after idle is not in min / max any more).
# Commit stamps come from tools/.collect_stamp.json (written by tools/collect.sh in the container: the box has no .git).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
TR_STEPS=10; TR_WARMUP=5; TR_PRE=30
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --precondition 0 --profile-steps 2 --no-cpu --no-config4 --no-single "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --precondition 0 --profile-steps 2 --no-cpu --no-config4 --no-single "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps $TR_STEPS --warmup $TR_WARMUP --precondition $TR_PRE --no-cpu --no-config4 --no-single "$@" > $OUT/trace.log 2>&1
export TR_STEPS TR_WARMUP TR_PRE
python3 - $OUT <<'PY'
import csv, collections, glob, json, sys
out=sys.argv[1]
def agg(d, name):
    tot=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob(out+'/'+d+'/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']!=name: continue
            k=r['Kernel_Name'].split('(')[0].replace('void ','')
            tot[k]+=float(r['Counter_Value']); cnt[k]+=1
    return {k:(tot[k]/cnt[k], cnt[k]) for k in tot}
fetch=agg('fetch','FETCH_SIZE'); write=agg('write','WRITE_SIZE')
# kernel durations from the kernel TRACE: bench.py runs (warm-up + steps) cold calls, then max(pre, warm-up) untimed and
# `steps` timed ones (the pre-conditioned region: these are averaged), then its event window and the verify step
import os
STEPS, WARMUP, PRE = int(os.environ['TR_STEPS']), int(os.environ['TR_WARMUP']), int(os.environ['TR_PRE'])
SKIP = (WARMUP + STEPS) + max(PRE, WARMUP)
calls=collections.defaultdict(list)
for f in glob.glob(out+'/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        calls[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
dur={}
rows=[]
for k,v in calls.items():
    v.sort()
    per_step = max(1, round(len(v) / float((WARMUP + STEPS) + max(PRE, WARMUP) + STEPS + 10 + max(20, STEPS) + 1)))  # launches per step (groups)
    d=[x[1] for x in v[SKIP*per_step:(SKIP+STEPS)*per_step]] or [x[1] for x in v]
    dur[k]={'calls':len(d),'avg_ns':sum(d)/len(d),'min_ns':min(d),'max_ns':max(d)}
    rows.append((sum(d),k,len(d),sum(d)/len(d),min(d),max(d)))
try: stamp=json.load(open(os.path.join(os.environ.get('GRAFT_REPO_ROOT', out+'/../../..'), 'tools', '.collect_stamp.json')))
except Exception: stamp={}
with open(out+'/kernel_stats_nowarm.csv','w') as fo:
    fo.write('# rocprofv3 --kernel-trace, calls of the pre-conditioned timed region only (%d untimed steps before them); head %s, kernel commit %s\n' % (max(PRE, WARMUP), stamp.get('head','?'), stamp.get('kernel_commit','?')))
    fo.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
    for tot,k,n,avg,mn,mx in sorted(rows,reverse=True):
        fo.write('"%s",%d,%d,%.1f,%d,%d\n' % (k,n,tot,avg,mn,mx))
bench=json.loads([l for l in open(out+'/trace.log') if l.startswith('{')][-1])
import datetime
res={'config':bench['config'],'date':datetime.date.today().isoformat(),'commit':stamp.get('kernel_commit','?'),'head':stamp.get('head','?'),
     'kernel_commit':stamp.get('kernel_commit','?'),'untimed_steps_before_stats':max(PRE, WARMUP),'kernels':{}}
for k in fetch:
    if not k.startswith('k_'): continue
    fb=2.0*fetch[k][0]*1024.0; wb=write.get(k,(0,0))[0]*1024.0
    res['kernels'][k]={'fetch_bytes_per_launch_corrected':fb,'write_bytes_per_launch':wb,'hbm_bytes_per_launch':fb+wb,
                       'avg_launch_ns':dur.get(k,{}).get('avg_ns'),'min_launch_ns':dur.get(k,{}).get('min_ns'),'max_launch_ns':dur.get(k,{}).get('max_ns'),'launches_profiled':fetch[k][1]}
json.dump(res, open(out+'/traffic.json','w'), indent=1)
for k,v in res['kernels'].items():
    ns=v['avg_launch_ns'] or 1
    print('%-28s %8.1f MB/launch  %7.1f us  %6.2f TB/s' % (k, v['hbm_bytes_per_launch']/1e6, ns/1e3, v['hbm_bytes_per_launch']/ns/1e3))
PY

#!/bin/bash
# usage (on the GPU box, via gpurun): tools/traffic.sh <outdir-under-gpurun_out> [bench args...]
# HBM traffic per kernel launch from the PMC counters, collected as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC has 4 slots: 3 + 2),
# FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), units KiB.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-config4 --no-single "$@" > $OUT/trace.log 2>&1
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
# kernel durations from the kernel TRACE, the warm-up calls of the profiled run dropped (the --stats summary averages
# them in): per kernel the calls in start order, the first WARM of them excluded
WARM=2
calls=collections.defaultdict(list)
for f in glob.glob(out+'/trace/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        calls[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])-int(r['Start_Timestamp'])))
dur={}
rows=[]
for k,v in calls.items():
    v.sort()
    d=[x[1] for x in (v[WARM:] if len(v)>WARM+2 else v)]
    dur[k]={'calls':len(d),'avg_ns':sum(d)/len(d),'min_ns':min(d),'max_ns':max(d)}
    rows.append((sum(d),k,len(d),sum(d)/len(d),min(d),max(d)))
with open(out+'/kernel_stats_nowarm.csv','w') as fo:
    fo.write('"Name","Calls","TotalDurationNs","AverageNs","MinNs","MaxNs"\n')
    for tot,k,n,avg,mn,mx in sorted(rows,reverse=True):
        fo.write('"%s",%d,%d,%.1f,%d,%d\n' % (k,n,tot,avg,mn,mx))
bench=json.loads([l for l in open(out+'/trace.log') if l.startswith('{')][-1])
import subprocess, datetime
try: commit=subprocess.run(['git','-C',out+'/../..','rev-parse','--short','HEAD'],capture_output=True,text=True).stdout.strip() or '?'
except Exception: commit='?'
res={'config':bench['config'],'date':datetime.date.today().isoformat(),'commit':commit,'warmup_calls_dropped':WARM,'kernels':{}}
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

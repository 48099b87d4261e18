#!/bin/bash
# usage: tools/pmc.sh <outdir> <bench args...>   (run on the GPU box via gpurun)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/sq2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-config4 --no-single "$@" > $OUT/trace.log 2>&1
python3 - $OUT <<'PY'
import csv, collections, glob, sys
out=sys.argv[1]
dur={}
for f in glob.glob(out+'/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        dur[r['Name'].split('(')[0][:40]]=(int(r['Calls']),float(r['AverageNs']))
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for d in ['sq1','sq2']:
    for f in glob.glob(out+'/'+d+'/*/*counter_collection.csv'):
        seen=set()
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'].split('(')[0][:40]
            tot[k][r['Counter_Name']]+=float(r['Counter_Value'])
            if d=='sq1' and (k,r['Dispatch_Id']) not in seen:
                seen.add((k,r['Dispatch_Id'])); cnt[k]+=1
for k in tot:
    if 'k_' not in k: continue
    c=tot[k]; n=max(cnt[k],1)
    wc=c['SQ_WAVE_CYCLES']; 
    line={'disp':n,'avg_us':round(dur.get(k,(0,0))[1]/1e3,1),'waves':round(c['SQ_WAVES']/n),
      'cyc/wave':round(4*wc/max(c['SQ_WAVES'],1)), 'valu/wave':round(c['SQ_INSTS_VALU']/max(c['SQ_WAVES'],1)),
      'lds/wave':round(c['SQ_INSTS_LDS']/max(c['SQ_WAVES'],1)),'vmemrd/wave':round(c['SQ_INSTS_VMEM_RD']/max(c['SQ_WAVES'],1),1),
      'valu_act%':round(100*c['SQ_ACTIVE_INST_VALU']/max(wc,1),1),'wait_any%':round(100*c['SQ_WAIT_ANY']/max(wc,1),1),
      'wait_inst%':round(100*c['SQ_WAIT_INST_ANY']/max(wc,1),1),'act_any%':round(100*c['SQ_ACTIVE_INST_ANY']/max(wc,1),1),
      'lds_conf%':round(100*c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1),1)}
    print(k, line)
PY

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5m; mkdir -p $O; cd $R
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]) if False else None
" 2>/dev/null
python3 - $O/bench.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith('{')][-1])
print('value',round(d['value']),'frac',round(d['roofline']['path']['frac'],4),'dom',d['roofline']['kernel'],round(d['roofline']['frac'],3))
print('kernels',{k:round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()})
print('config4',round(d['config4']['value']), 'capi', d.get('config4_capi'))
c=d['cpu_baseline']; print('cpu',c['value'],c['cores'],c['spread'],[ (p['workers'],p['per_s'],p['spread']) for p in c['sweep']])
print('single',d['single_pair'])
PY
bash tools/dbg/placement_spread.sh 8 > $O/placement.txt 2>&1; grep -v amdgpu $O/placement.txt

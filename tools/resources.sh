#!/bin/bash
# tools/resources.sh -- VGPRs / scratch / occupancy / LDS of every kernel in csrc/xcorr_kernels.hip (no GPU needed)
cd "$(dirname "$0")/../old-audiosync_amd" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-finite-math-only -fno-slp-vectorize \
    -I../include -Icsrc -c csrc/xcorr_kernels.hip -o /tmp/asx_res.o -Rpass-analysis=kernel-resource-usage $EXTRA 2>/tmp/asx_res.txt
python3 - <<'PY'
import re
cur=None; rows=[]
for line in open('/tmp/asx_res.txt'):
    m=re.search(r'remark: +(.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        cur={'name':t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
import subprocess
for r in rows:
    n=subprocess.run(['c++filt', r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(.*','',n).replace('void ','')
    print("%-60s vgpr %4s agpr %3s sgpr %4s scratch %5s occ %2s lds %6s" % (n[:60], r.get('VGPRs'), r.get('AGPRs'), r.get('TotalSGPRs'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
PY

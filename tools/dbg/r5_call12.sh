#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5l; mkdir -p $O; cd $R
{
bash tools/ab.sh 2 r_base r_scan r_scan2
for n in 144000 288000 480000; do echo "== N=$n"; bash tools/dbg/ab_n.sh 2 $n 1024 r_base r_scan r_scan2; done
} 2>&1 | grep -v "^$" > $O/ab.txt
grep -v amdgpu $O/ab.txt

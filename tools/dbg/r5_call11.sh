#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5k; mkdir -p $O; cd $R
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
{
bash tools/ab.sh 2 r_base r_scan
for n in 144000 480000 960000; do echo "== N=$n"; bash tools/dbg/ab_n.sh 1 $n 1024 r_base r_scan; done
} 2>&1 | grep -v "^$" > $O/ab.txt
grep -v amdgpu $O/ab.txt

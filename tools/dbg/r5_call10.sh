#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5j; mkdir -p $O; cd $R
{
bash tools/ab.sh 2 r_base r_stag
for n in 144000 288000 480000 720000 960000; do echo "== N=$n"; bash tools/dbg/ab_n.sh 2 $n 1024 r_base r_stag; done
} 2>&1 | grep -v "^$" > $O/ab.txt
grep -v amdgpu $O/ab.txt
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3

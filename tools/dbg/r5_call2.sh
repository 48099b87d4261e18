#!/bin/bash
# round 5, GPU call 2: launch ORDER of the three transform kernels (pair-fastest instead of tile- / row-fastest), all lengths
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5b; mkdir -p $O; cd $R
{
echo "== N = 1 440 000"
bash tools/ab.sh 2 r_base r_tm r_tm_rpf r_tm_fpf r_tm_rpf_fpf
for n in 960000 720000 480000 288000 144000; do
  echo "== N = $n x 1024"
  bash tools/dbg/ab_n.sh 1 $n 1024 r_base r_tm r_tm_rpf r_tm_fpf r_tm_rpf_fpf
done
} > $O/ab.txt 2>&1
grep -v "amdgpu.ids" $O/ab.txt

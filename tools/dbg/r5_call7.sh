#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5g; mkdir -p $O; cd $R
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
{
for args in "" "--sample-len 288000 --batch 1024 --steps 20" "--sample-len 480000 --batch 1024 --steps 20" "--sample-len 144000 --batch 1024 --steps 20"; do
  echo "== $args"
  for r in 1 2; do
  for l in r_v0 r_v1 r_v2 r_v0d r_v2d; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l: "; $B $args 2>/dev/null | python3 $R/tools/brief.py
  done
  cp $R/ab/r_v0.so $R/old-audiosync_amd/libaudiosync_hip.so
  echo -n "direct: "; ASX_PEARSON=direct $B $args 2>/dev/null | python3 $R/tools/brief.py
  done
done
} > $O/ab.txt 2>&1
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so
cat $O/ab.txt

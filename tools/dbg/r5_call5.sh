#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5e; mkdir -p $O; cd $R
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
{
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in 1 2; do
  for l in r_sp r_sp_nostore r_sp_tm; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l: "; $B 2>/dev/null | python3 $R/tools/brief.py
  done
  cp $R/ab/r_sp.so $R/old-audiosync_amd/libaudiosync_hip.so
  echo -n "direct: "; ASX_PEARSON=direct $B 2>/dev/null | python3 $R/tools/brief.py
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so
} > $O/ab.txt 2>&1
cat $O/ab.txt

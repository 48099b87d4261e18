#!/bin/bash
# block-size sweep of the column kernels (ab/r_nt.so carries the extra instantiations): nt_sweep.sh <N> <nt> ...
R=$GRAFT_REPO_ROOT; N=$1; shift; cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so; cp $R/ab/r_nt.so $R/old-audiosync_amd/libaudiosync_hip.so
for r in 1 2; do for nt in "$@"; do
  echo -n "NT=$nt N=$N: "; ASX_RCOL_NT=$nt python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 8 --sample-len $N --batch 1024 2>/dev/null | python3 $R/tools/brief.py
done; done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

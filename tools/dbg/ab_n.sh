#!/bin/bash
# tools/dbg/ab_n.sh <rounds> <N> <batch> <lib> ... : A/B of library builds at one sample length
R=$GRAFT_REPO_ROOT; rounds=$1; N=$2; B=$3; shift 3
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in $(seq $rounds); do
  for l in "$@"; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l N=$N: "; python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 10 --sample-len $N --batch $B 2>/dev/null | python3 $R/tools/brief.py
  done
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

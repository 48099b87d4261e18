#!/bin/bash
# tools/dbg/ab_two.sh <rounds> <lib> ... : A/B of library builds on the headline (600x2400x16) and on N=720000 (600x1200x16)
R=$GRAFT_REPO_ROOT; rounds=$1; shift
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in $(seq $rounds); do
  for l in "$@"; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l 1.44M: "; python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 30 | python3 $R/tools/brief.py
    echo -n "$l 720k : "; python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 8 --sample-len 720000 --batch 1024 | python3 $R/tools/brief.py
  done
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

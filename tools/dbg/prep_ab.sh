#!/bin/bash
# usage (GPU box): tools/dbg/prep_ab.sh <rounds> <lib> ... : per build, the single-pair call (N = 1 440 000 and 960 000) and the batched headline
R=$GRAFT_REPO_ROOT; rounds=$1; shift
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in $(seq $rounds); do
  for l in "$@"; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l: "; python3 $R/tools/dbg/single_pair_loop.py 1440000 200 2>/dev/null | tr '\n' ' '; python3 $R/tools/dbg/single_pair_loop.py 960000 200 2>/dev/null | tr '\n' ' '
    python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 20 2>/dev/null | python3 $R/tools/brief.py
  done
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

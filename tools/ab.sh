#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab.sh <rounds> <libA> <libB> ... [-- bench args]
# A/B timing of library builds kept under ab/<name>.so: each round runs every build once, in turn,
# on the same box (boxes differ by a few percent, so only same-box comparisons count).
R=$GRAFT_REPO_ROOT; rounds=$1; shift
libs=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do libs+=("$1"); shift; done; [ "$1" == "--" ] && shift
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in $(seq $rounds); do
  for l in "${libs[@]}"; do
    cp $R/ab/$l.so $R/old-audiosync_amd/libaudiosync_hip.so
    echo -n "$l: "; python3 $R/bench.py --no-cpu --no-config4 --no-single "$@" | python3 $R/tools/brief.py
  done
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

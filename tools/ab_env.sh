#!/bin/bash
# usage (on the GPU box, via gpurun): tools/ab_env.sh <rounds> <variant> ... [-- bench args]
# variant = <lib>[:ENV=VAL[,ENV=VAL...]]: ab/<lib>.so (or "main" = the library as built) run with those environment variables.
# Like tools/ab.sh, each round runs every variant once, in turn, on the same box.
R=$GRAFT_REPO_ROOT; rounds=$1; shift
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done; [ "$1" == "--" ] && shift
cp $R/old-audiosync_amd/libaudiosync_hip.so /tmp/asx_keep.so
for r in $(seq $rounds); do
  for v in "${vars[@]}"; do
    lib=${v%%:*}; envs=""; [[ "$v" == *:* ]] && envs=${v#*:}
    if [ "$lib" == "main" ]; then cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so; else cp $R/ab/$lib.so $R/old-audiosync_amd/libaudiosync_hip.so; fi
    echo -n "$v: "; env ${envs//,/ } python3 $R/bench.py --no-cpu --no-config4 --no-single "$@" 2>/dev/null | python3 $R/tools/brief.py
  done
done
cp /tmp/asx_keep.so $R/old-audiosync_amd/libaudiosync_hip.so

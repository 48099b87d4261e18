#!/bin/bash
# spread of k_fwd_cols_r over 8 processes, default planning against split = "measure" (which keeps the faster of two
# placements of the forward column kernel's output workspaces).  usage (GPU box): tools/dbg/placement_spread.sh [processes]
R=$GRAFT_REPO_ROOT; P=${1:-8}
B="python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 40"
for mode in default measure; do
  for i in $(seq $P); do
    echo -n "$mode $i: "
    if [ $mode == measure ]; then ASX_SPLIT=measure $B 2>/dev/null | python3 $R/tools/brief.py; else $B 2>/dev/null | python3 $R/tools/brief.py; fi
  done
done

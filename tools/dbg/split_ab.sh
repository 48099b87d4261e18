#!/bin/bash
# same-box A/B over (layout, split) combinations: tools/dbg/split_ab.sh <rounds> <N> "<layout>:<split>" ...
cd $GRAFT_REPO_ROOT
R=$1; N=$2; shift 2
for r in $(seq $R); do
  for c in "$@"; do
    lay=${c%%:*}; sp=${c##*:}
    echo -n "$c : "; ASX_LAYOUT=$lay python3 bench.py --no-cpu --no-config4 --no-single --steps 40 --sample-len $N --split $sp | python3 tools/brief.py
  done
done

#!/bin/bash
# same-box A/B of the two decompositions (ASX_LAYOUT=packed: the packed-sample kernels; default: real-column kernels)
cd $GRAFT_REPO_ROOT
R=${1:-3}; shift
for r in $(seq $R); do
  echo -n "packed : "; ASX_LAYOUT=packed python3 bench.py --no-cpu --no-config4 --no-single --steps 40 "$@" | python3 tools/brief.py
  echo -n "real   : "; python3 bench.py --no-cpu --no-config4 --no-single --steps 40 "$@" | python3 tools/brief.py
done

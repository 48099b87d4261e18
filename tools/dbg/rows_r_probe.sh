#!/bin/bash
# timing probe of k_rows_r against k_rows (same box, alternating): results of the probe runs are meaningless, traffic is right
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  echo -n "k_rows   : "; python3 bench.py --no-cpu --no-config4 --no-single --steps 40 | python3 tools/brief.py
  echo -n "k_rows_r : "; ASX_TIME_ROWS_R=1 python3 bench.py --no-cpu --no-config4 --no-single --steps 40 | python3 tools/brief.py
done

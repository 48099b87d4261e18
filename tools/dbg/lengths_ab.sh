#!/bin/bash
# same-box A/B of the two decompositions over the six reference lengths (batches of 1024 pairs; headline: 124)
cd $GRAFT_REPO_ROOT
for n in 144000 288000 480000 720000 960000 1440000; do
  b=1024; [ $n = 1440000 ] && b=0
  for lay in packed real; do
    echo -n "N=$n $lay : "; ASX_LAYOUT=$lay python3 bench.py --no-cpu --no-config4 --no-single --steps ${STEPS:-10} --sample-len $n --batch $b | python3 tools/brief.py
  done
done

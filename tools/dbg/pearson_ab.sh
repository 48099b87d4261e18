#!/bin/bash
# usage (GPU box): tools/dbg/pearson_ab.sh [rounds]   -- the spectral Pearson form against the direct one (ASX_PEARSON=direct), same build,
# alternating runs, the headline length and 1024-pair batches of the other five (profiles/r5_experiments/04_*)
R=$GRAFT_REPO_ROOT; rounds=${1:-2}
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
for args in "" "--sample-len 144000 --batch 1024 --steps 20" "--sample-len 288000 --batch 1024 --steps 20" "--sample-len 480000 --batch 1024 --steps 20" "--sample-len 720000 --batch 1024 --steps 20" "--sample-len 960000 --batch 1024 --steps 20"; do
  echo "== ${args:-N = 1 440 000, 124 pairs}"
  for r in $(seq $rounds); do
    echo -n "spectral: "; $B $args 2>/dev/null | python3 $R/tools/brief.py
    echo -n "direct: "; ASX_PEARSON=direct $B $args 2>/dev/null | python3 $R/tools/brief.py
  done
done

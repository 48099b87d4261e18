#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
echo -n "exact (default): "; python3 bench.py --no-cpu --no-config4 --no-single --steps 40 2>/dev/null | python3 tools/brief.py
echo -n "asynchronous   : "; ASX_EXACT=0 python3 bench.py --no-cpu --no-config4 --no-single --steps 40 2>/dev/null | python3 tools/brief.py
done

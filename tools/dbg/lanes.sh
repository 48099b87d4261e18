#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo -n "1 lane          : "; python3 bench.py --no-cpu --no-config4 --no-single --steps 30 2>/dev/null | python3 tools/brief.py
echo -n "2 lanes (2x62)  : "; ASX_LANES=2 python3 bench.py --no-cpu --no-config4 --no-single --steps 30 2>/dev/null | python3 tools/brief.py
echo -n "2 lanes, 4 groups: "; ASX_LANES=2 ASX_WS_MB=2200 python3 bench.py --no-cpu --no-config4 --no-single --steps 30 2>/dev/null | python3 tools/brief.py
echo -n "1 lane, batch 248: "; ASX_WS_MB=8192 python3 bench.py --no-cpu --no-config4 --no-single --steps 15 --batch 248 2>/dev/null | python3 tools/brief.py
done

#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5f; mkdir -p $O; cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "all rc=$?"; tail -5 $O/pytest.txt
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
{
for r in 1 2 3; do
  echo -n "spectral: "; $B 2>/dev/null | python3 $R/tools/brief.py
  echo -n "direct: "; ASX_PEARSON=direct $B 2>/dev/null | python3 $R/tools/brief.py
done
for n in 144000 288000 480000 720000 960000; do
  echo "== N=$n x 1024"
  echo -n "spectral: "; $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py; echo -n "direct: "; ASX_PEARSON=direct $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py
done
} > $O/ab.txt 2>&1
cat $O/ab.txt

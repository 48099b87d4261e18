#!/bin/bash
# round 5, GPU call 3: the spectral Pearson form -- parity, then A/B against the direct form on one box
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5c; mkdir -p $O; cd $R
python3 -m pytest tests/test_gpu_pearson_spectral.py -x -q -s > $O/pytest_spectral.txt 2>&1; echo "spectral rc=$?"; tail -25 $O/pytest_spectral.txt
python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "all rc=$?"; tail -5 $O/pytest.txt
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
run() { echo -n "$1: "; shift; env "$@" $B 2>/dev/null | python3 $R/tools/brief.py; }
{
for r in 1 2; do run spectral X=1; run direct ASX_PEARSON=direct; done
for n in 144000 480000; do
  echo "== N=$n x 1024"
  echo -n "spectral: "; $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py; echo -n "direct: "; ASX_PEARSON=direct $B --sample-len $n --batch 1024 --steps 20 2>/dev/null | python3 $R/tools/brief.py
done
} > $O/ab.txt 2>&1
cat $O/ab.txt

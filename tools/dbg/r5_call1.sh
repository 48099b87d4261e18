#!/bin/bash
# round 5, GPU call 1: parity suite on the ADVICE fix, then same-box A/B of (a) the 1200-point sub-row schedules of the
# two-half row kernel, (b) the inverse column kernel's launch order / per-wave exit, (c) Infinity-Cache upper bounds
# (pair-modulo workspaces) and REAL small launch groups on one and two lanes, (d) Q written in place over C_x.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5a; mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.txt; tail -3 $O/pytest.txt
B="python3 $R/bench.py --no-cpu --no-config4 --no-single"
run() { echo -n "$1: "; shift; env "$@" $B 2>/dev/null | python3 $R/tools/brief.py; }
{
echo "== schedules / inverse order (2 rounds, alternating)"
bash tools/ab.sh 2 r_base r_s101210 r_s101012 r_s81510 r_s81015 r_s51615 r_s151008 r_s150810 r_tm r_tmws r_ws
echo "== Infinity-Cache upper bound (wrong results by design)"
bash tools/ab.sh 1 r_base r_pm4 r_pm6
echo "== Q in place"
run base X=1; run qinplace ASX_Q_INPLACE=1; run base X=1; run qinplace ASX_Q_INPLACE=1
echo "== real small groups, one lane"
for mb in 140 210 280 420 830; do run "ws${mb}_l1" ASX_WS_MB=$mb; run "ws${mb}_l1_qin" ASX_WS_MB=$mb ASX_Q_INPLACE=1; done
echo "== real small groups, two lanes (group per lane = WS/2)"
for mb in 280 420 560 840; do run "ws${mb}_l2" ASX_WS_MB=$mb ASX_LANES=2; run "ws${mb}_l2_qin" ASX_WS_MB=$mb ASX_LANES=2 ASX_Q_INPLACE=1; done
} > $O/ab.txt 2>&1
cat $O/ab.txt

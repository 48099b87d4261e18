#!/bin/bash
# usage (GPU box, via gpurun): tools/collect_r3.sh   -- the bench / profile files of this round under gpurun_out/r3c/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3c; mkdir -p $O
python3 $R/bench.py > $O/r3_bench.json 2> $O/r3_bench.err
for n in 144000 288000 480000 720000 960000; do
  python3 $R/bench.py --sample-len $n --batch 1024 --no-cpu --no-config4 --no-single > $O/r3_bench_N$n.json 2>> $O/r3_bench.err
done
python3 $R/bench.py --mode streaming --steps 5 > $O/r3_bench_streaming.json 2>> $O/r3_bench.err
python3 $R/bench.py --mode single --steps 20 > $O/r3_bench_single.json 2>> $O/r3_bench.err
ASX_ROWS2=1 python3 $R/bench.py --no-cpu --no-config4 --no-single > $O/rows2_bench.json 2>> $O/r3_bench.err
$R/tools/pmc.sh r3pmc > $O/r3_pmc_summary.txt 2>&1
ASX_ROWS2=1 $R/tools/pmc.sh r3pmc_rows2 > $O/rows2_pmc_summary.txt 2>&1
cd $R/tools/micro && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/pipe_handoff pipe_handoff.hip 2>/dev/null && timeout -k 10 120 /tmp/pipe_handoff 128 8 > $O/pipe_handoff.txt 2>&1
for f in $O/r3_bench*.json $O/rows2_bench.json; do echo -n "$(basename $f): "; python3 $R/tools/brief.py < $f 2>/dev/null || echo "(side mode)"; done

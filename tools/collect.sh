#!/bin/bash
# usage (in the container): tools/collect.sh <round>     e.g. tools/collect.sh 4
# Regenerates every profiles/r<round>_* file from HEAD, mechanically:
#   1. refuses to run on a dirty tree (what is measured is what is committed);
#   2. writes tools/.collect_stamp.json = {head, kernel_commit (last commit that touched old-audiosync_amd/csrc), date};
#   3. runs tools/collect_box.sh on a GPU box (one gpurun call): bench lines of every BASELINE configuration, PMC traffic
#      and rocprofv3 kernel-trace stats (taken behind >= 30 untimed steps) at N = 1 440 000, 288 000 x 1024,
#      480 000 x 1024 and 720 000 x 512, the SQ counter summary -- each stamped;
#   4. copies the results to profiles/r<round>_*.
# tests/test_profiles_fresh.py fails when a committed profiles/r<round>_* stamp is older than the kernels at HEAD.
set -e
cd "$(dirname "$0")/.."
RD=${1:?round number}
if [ -n "$(git status --porcelain)" ]; then echo "tools/collect.sh: the tree is dirty; commit first" >&2; git status --short >&2; exit 1; fi
HEAD=$(git rev-parse --short HEAD); KC=$(git log -1 --format=%h -- old-audiosync_amd/csrc)
printf '{"head": "%s", "kernel_commit": "%s", "date": "%s"}\n' "$HEAD" "$KC" "$(date -u +%Y-%m-%dT%H:%MZ)" > tools/.collect_stamp.json
(cd old-audiosync_amd && make -j8 > /dev/null) && make -C oracle > /dev/null
rm -rf gpurun_out/r${RD}c
/usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/collect_box.sh $RD"
O=gpurun_out/r${RD}c; P=profiles
[ -s $O/bench.json ] || { echo "no bench line came back" >&2; exit 1; }
cp $O/bench.json $P/r${RD}_bench.json
for n in 144000 288000 480000 720000 960000; do cp $O/bench_N$n.json $P/r${RD}_bench_N$n.json; done
cp $O/bench_packed.json $P/r${RD}_bench_packed_layout.json
cp $O/bench_streaming.json $P/r${RD}_bench_streaming.json; cp $O/bench_single.json $P/r${RD}_bench_single.json
cp $O/traffic_N1440000/traffic.json $P/r${RD}_traffic.json; cp $O/traffic_N1440000/kernel_stats_nowarm.csv $P/r${RD}_kernel_stats.csv
for n in 288000 480000 720000; do cp $O/traffic_N$n/traffic.json $P/r${RD}_traffic_N$n.json; cp $O/traffic_N$n/kernel_stats_nowarm.csv $P/r${RD}_kernel_stats_N$n.csv; done
cp $O/pmc_summary.txt $P/r${RD}_pmc_summary.txt
echo "profiles/r${RD}_* regenerated at head $HEAD (kernels: $KC)"

#!/bin/bash
# usage (GPU box, via gpurun; started by tools/collect.sh): tools/collect_box.sh <round>
# The round's bench / profile files under gpurun_out/r<round>c/, every JSON / CSV stamped with the commit it was taken at
# (tools/.collect_stamp.json, written in the container by tools/collect.sh: the box has no .git).
R=$GRAFT_REPO_ROOT; RD=$1; O=$R/gpurun_out/r${RD}c; mkdir -p $O
STAMP=$R/tools/.collect_stamp.json
[ -f $STAMP ] || { echo "no tools/.collect_stamp.json: start this through tools/collect.sh"; exit 2; }
stamp() { # stamp <json file>: add head / kernel_commit to a bench line
  python3 - "$1" "$STAMP" <<'PY'
import json, sys
path, st = sys.argv[1], json.load(open(sys.argv[2]))
lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
if not lines: sys.exit(0)
d = json.loads(lines[-1]); d["head"] = st["head"]; d["kernel_commit"] = st["kernel_commit"]; d["collected"] = st["date"]
open(path, "w").write(json.dumps(d) + "\n")
PY
}
echo "== traffic + kernel trace, N=1440000"; $R/tools/traffic.sh r${RD}c/traffic_N1440000 > $O/traffic_N1440000.txt 2>&1
echo "== traffic + kernel trace, N=288000 x 1024"; $R/tools/traffic.sh r${RD}c/traffic_N288000 --sample-len 288000 --batch 1024 > $O/traffic_N288000.txt 2>&1
echo "== traffic + kernel trace, N=480000 x 1024"; $R/tools/traffic.sh r${RD}c/traffic_N480000 --sample-len 480000 --batch 1024 > $O/traffic_N480000.txt 2>&1
echo "== traffic + kernel trace, N=720000 x 512"; $R/tools/traffic.sh r${RD}c/traffic_N720000 --sample-len 720000 --batch 512 > $O/traffic_N720000.txt 2>&1
# the bench lines below quote the PMC traffic of THIS collection (bench.py reads profiles/r<round>_traffic*.json)
cp $O/traffic_N1440000/traffic.json $R/profiles/r${RD}_traffic.json; cp $O/traffic_N288000/traffic.json $R/profiles/r${RD}_traffic_N288000.json; cp $O/traffic_N480000/traffic.json $R/profiles/r${RD}_traffic_N480000.json; cp $O/traffic_N720000/traffic.json $R/profiles/r${RD}_traffic_N720000.json
echo "== bench (headline, cpu baseline, config4, single pair)"; python3 $R/bench.py > $O/bench.json 2> $O/bench.err; stamp $O/bench.json
for n in 144000 288000 480000 720000 960000; do
  echo "== bench N=$n x 1024"; python3 $R/bench.py --sample-len $n --batch 1024 --steps 20 --no-cpu --no-config4 --no-single > $O/bench_N$n.json 2>> $O/bench.err; stamp $O/bench_N$n.json
done
echo "== packed layout, same box (A/B reference)"; ASX_LAYOUT=packed python3 $R/bench.py --no-cpu --no-config4 --no-single > $O/bench_packed.json 2>> $O/bench.err; stamp $O/bench_packed.json
echo "== streaming / single"; python3 $R/bench.py --mode streaming --steps 5 > $O/bench_streaming.json 2>> $O/bench.err; stamp $O/bench_streaming.json
python3 $R/bench.py --mode single --steps 20 > $O/bench_single.json 2>> $O/bench.err; stamp $O/bench_single.json
echo "== SQ counters"; $R/tools/pmc.sh r${RD}c/pmc --precondition 0 --profile-steps 2 > $O/pmc_summary.txt 2>&1
python3 - $STAMP $O/pmc_summary.txt <<'PY'
import json, sys
st = json.load(open(sys.argv[1])); txt = open(sys.argv[2]).read()
open(sys.argv[2], "w").write("# tools/pmc.sh, head %s, kernel commit %s, %s\n" % (st["head"], st["kernel_commit"], st["date"]) + txt)
PY
for f in $O/bench.json $O/bench_N*.json $O/bench_packed.json; do echo -n "$(basename $f): "; python3 $R/tools/brief.py < $f 2>/dev/null || echo "(no line)"; done
cat $O/traffic_N1440000.txt | tail -12

cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for t in . ab/r4tree; do
    echo -n "$t streaming: "; python3 $t/bench.py --mode streaming --steps 5 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['value'],3), {k:round(v,3) for k,v in d['ms_per_interval'].items()})"
    echo -n "$t single: "; python3 $t/bench.py --mode single --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(round(d['resident_float32_ms'],4), round(d['double_abi_incl_h2d_ms'],4))"
  done
done

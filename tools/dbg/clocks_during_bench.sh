R=$GRAFT_REPO_ROOT
( for i in $(seq 1 40); do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.load(sys.stdin); c=d.get('card0',{})
    print({k:v for k,v in c.items() if any(s in k.lower() for s in ('sclk','mclk','fclk','power','temp'))})
except Exception as e: print('err',e)
"; sleep 0.25; done ) > $R/gpurun_out/clk.log 2>&1 &
sleep 1
python3 $R/bench.py --no-cpu --no-config4 --no-single --steps 2500 | python3 $R/tools/brief.py
wait

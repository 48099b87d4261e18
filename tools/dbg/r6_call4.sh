set -e
mkdir -p gpurun_out/r6
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6/06_gpu_tests.log 2>&1 || { tail -40 gpurun_out/r6/06_gpu_tests.log; exit 1; }
tail -3 gpurun_out/r6/06_gpu_tests.log
python bench.py > gpurun_out/r6/06_bench.json 2> gpurun_out/r6/06_bench.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6/06_bench.json") if l.startswith("{")][-1])
print("value", round(d["value"]), "ok", d["results_ok"], "coef_check", d["coef_check"], "modes", d.get("pearson_modes"))
print("cfg4", round(d["config4"]["value"]), d["config4"]["results_ok"], d["config4"]["coef_check"])
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("oracle_pair0"))
print("traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
PY
for r in 1 2 3; do for sp in 600x1200x16 300x2400x16; do echo -n "N=720000 split $sp: "; python bench.py --no-cpu --no-config4 --no-single --sample-len 720000 --batch 512 --split $sp 2>/dev/null | python tools/brief.py; done; done | tee gpurun_out/r6/07_split_N720000.txt

set -e
mkdir -p gpurun_out/r6
tools/ab_env.sh 4 main r_chain 2>&1 | tee gpurun_out/r6/08_band_tree_ab.txt
tools/ab_env.sh 2 main r_chain -- --sample-len 480000 --batch 1024 2>&1 | tee -a gpurun_out/r6/08_band_tree_ab.txt
timeout -k 10 400 python tools/fuzz_parity.py 150 601 > gpurun_out/r6/09_fuzz_a.txt 2>&1 || { tail -20 gpurun_out/r6/09_fuzz_a.txt; exit 1; }
tail -4 gpurun_out/r6/09_fuzz_a.txt
FUZZ_BIG=1 timeout -k 10 400 python tools/fuzz_parity.py 150 602 > gpurun_out/r6/09_fuzz_b.txt 2>&1 || { tail -20 gpurun_out/r6/09_fuzz_b.txt; exit 1; }
tail -4 gpurun_out/r6/09_fuzz_b.txt
FUZZ_HUGE=1 timeout -k 10 400 python tools/fuzz_parity.py 150 603 > gpurun_out/r6/09_fuzz_c.txt 2>&1 || { tail -20 gpurun_out/r6/09_fuzz_c.txt; exit 1; }
tail -4 gpurun_out/r6/09_fuzz_c.txt

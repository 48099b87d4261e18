cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5fuzz
(FUZZ_HUGE=1 python3 tools/fuzz_parity.py 300 601 2>&1 | grep -v amdgpu | tail -3) | tee gpurun_out/r5fuzz/huge.txt
(FUZZ_BIG=1 python3 tools/fuzz_parity.py 300 602 2>&1 | grep -v amdgpu | tail -3) | tee gpurun_out/r5fuzz/big.txt
(python3 tools/fuzz_parity.py 300 603 2>&1 | grep -v amdgpu | tail -3) | tee gpurun_out/r5fuzz/mixed.txt

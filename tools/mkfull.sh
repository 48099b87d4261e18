#!/bin/bash
# usage: tools/mkfull.sh <name> "<extra hipcc flags for every HIP translation unit>"   (no GPU needed)
# Builds ab/<name>.so with EVERY object recompiled from the current sources with the extra flags (variants that
# change a struct layout or an experiment macro seen by several translation units must never mix old objects).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/old-audiosync_amd
name=$1; flags=$2
D=/tmp/asx_full/$name; rm -rf $D; mkdir -p $R/ab $D
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-finite-math-only -fno-slp-vectorize -Wall -Wno-unused-function -I$R/include -I$P/csrc"
objs=""
for p in 1 2 4 8 16 32 64; do
  kf=""; case $p in 1|4|16) kf="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac
  /opt/rocm/bin/hipcc $F $kf $flags -DASX_PART=$p -c -o $D/k$p.o $P/csrc/xcorr_kernels.hip &
  objs="$objs $D/k$p.o"
done
/opt/rocm/bin/hipcc $F $flags -c -o $D/rlayout.o $P/csrc/rlayout.hip &
/opt/rocm/bin/hipcc $F $flags -c -o $D/pspec.o $P/csrc/pearson_spectral.hip &
/opt/rocm/bin/hipcc $F $flags -c -o $D/api.o $P/csrc/asx_api.hip &
/opt/rocm/bin/hipcc $F $flags -c -o $D/plan.o $P/csrc/plan_math.cpp &
/opt/rocm/bin/hipcc $F $flags -c -o $D/shard.o $P/csrc/shard_driver.cpp &
/opt/rocm/bin/hipcc $F $flags -c -o $D/narrow.o $P/csrc/host_narrow.cpp &
wait
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $R/ab/$name.so $D/api.o $D/plan.o $D/shard.o $D/narrow.o $D/rlayout.o $D/pspec.o $objs -ldl -lpthread
echo "ab/$name.so"

#!/bin/bash
# usage: tools/mkp.sh <name> "<extra hipcc flags>"   (no GPU needed)
# Builds ab/<name>.so with csrc/pearson_spectral.hip recompiled with the extra flags, every other object from
# old-audiosync_amd/build/ (run `make` there first).  For tools/ab.sh.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/old-audiosync_amd
name=$1; flags=$2
mkdir -p $R/ab /tmp/asx_p/$name
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-finite-math-only -fno-slp-vectorize -Wall -Wno-unused-function -I$R/include -I$P/csrc"
/opt/rocm/bin/hipcc $F $flags -c -o /tmp/asx_p/$name/pearson_spectral.o $P/csrc/pearson_spectral.hip
objs="$P/build/asx_api.o $P/build/plan_math.o $P/build/shard_driver.o $P/build/host_narrow.o $P/build/rlayout.o /tmp/asx_p/$name/pearson_spectral.o"
for p in 1 2 4 8 16 32 64; do objs="$objs $P/build/kernels_$p.o"; done
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $R/ab/$name.so $objs -ldl -lpthread
echo "ab/$name.so"

#!/bin/bash
# usage: tools/mkvariant.sh <name> "<extra hipcc flags>" [part ...]   (no GPU needed)
# Builds ab/<name>.so: the listed parts of csrc/xcorr_kernels.hip (default: all) recompiled with the extra
# flags, every other object taken from old-audiosync_amd/build/ (run `make` there first).  For tools/ab.sh.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/old-audiosync_amd
name=$1; flags=$2; shift 2; parts=${@:-1 2 4 8 16 32 64}
mkdir -p $R/ab /tmp/asx_var/$name
F="${ASX_OPT:--O3} -std=c++17 -fPIC --offload-arch=${ASX_ARCH:-gfx950} -fno-finite-math-only -fno-slp-vectorize -Wall -Wno-unused-function -I$R/include -I$P/csrc"
objs="$P/build/asx_api.o $P/build/plan_math.o $P/build/shard_driver.o $P/build/host_narrow.o $P/build/rlayout.o $P/build/pearson_spectral.o"
for p in 1 2 4 8 16 32 64; do
  if [[ " $parts " == *" $p "* ]]; then
    kf=""; case $p in 1|4|16) kf="-mllvm -amdgpu-sched-strategy=iterative-ilp";; esac   # as in the Makefile (KFLAGS_STATIC)
    [ -n "$ASX_NO_KFLAGS" ] && kf=""
    /opt/rocm/bin/hipcc $F $kf $flags -DASX_PART=$p -c -o /tmp/asx_var/$name/k$p.o $P/csrc/xcorr_kernels.hip &
    objs="$objs /tmp/asx_var/$name/k$p.o"
  else
    objs="$objs $P/build/kernels_$p.o"
  fi
done
wait
/opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o $R/ab/$name.so $objs -ldl -lpthread
echo "ab/$name.so"

#!/bin/bash
# Experiment builds of the library with different -D switches for the kernels_*.hip units (the other objects are the product's):
#   tools/build_variants.sh name "-DLMN_X=1 ..." [name2 "..."]   ->  tools/bin/variants/<name>.so
#   ALLSRC=1 tools/build_variants.sh ...   compiles EVERY source with the switches (for switches read outside the kernel units)
# A GPU session copies one over luminair_amd/csrc/libluminair_hip.so in ITS scratch copy of the repo before benchmarking.
set -eu
cd "$(dirname "$0")/../luminair_amd/csrc"
make -s > /dev/null
mkdir -p ../../tools/bin/variants
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -x hip"
while [ $# -ge 2 ]; do
  OBJS=""
  for k in kernels_trace kernels_fft kernels_merkle kernels_logup kernels_quotient; do $CC $2 -c $k.hip -o /tmp/${k}_$1.o; OBJS="$OBJS /tmp/${k}_$1.o"; done
  for src in fft_fixed.hip components.cpp context.cpp trace_gen.cpp commit.cpp oods.cpp decommit.cpp quotients.cpp prove.cpp phase_trace.cpp phase_logup.cpp phase_composition.cpp phase_oods.cpp phase_fri.cpp phase_decommit.cpp shard.cpp ops.cpp verifier.cpp capi.cpp level2.cpp; do
    base=${src%.*}
    if [ "${ALLSRC:-0}" = 1 ]; then $CC $2 -c $src -o /tmp/${base}_$1.o; OBJS="$OBJS /tmp/${base}_$1.o"; else OBJS="$OBJS $base.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../../tools/bin/variants/$1.so
  echo "built tools/bin/variants/$1.so  ($2)"
  shift 2
done

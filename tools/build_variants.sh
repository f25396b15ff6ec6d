#!/bin/bash
# Experiment builds of the library with different -D switches for kernels.hip (the other objects are the product's):
#   tools/build_variants.sh name "-DLMN_X=1 ..." [name2 "..."]   ->  tools/bin/variants/<name>.so
# A GPU session copies one over luminair_amd/csrc/libluminair_hip.so in ITS scratch copy of the repo before benchmarking.
set -eu
cd "$(dirname "$0")/../luminair_amd/csrc"
make -s > /dev/null
mkdir -p ../../tools/bin/variants
while [ $# -ge 2 ]; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -x hip $2 -c kernels.hip -o /tmp/kernels_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/kernels_$1.o fft_fixed.o prover.o verifier.o capi.o level2.o -o ../../tools/bin/variants/$1.so
  echo "built tools/bin/variants/$1.so  ($2)"
  shift 2
done

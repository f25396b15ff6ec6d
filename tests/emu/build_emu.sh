#!/bin/bash
# Builds the TEST-ONLY emulation library (same sources, LMN_EMU): tests/emu/libluminair_emu.so
set -e
cd "$(dirname "$0")"
SRC=../../luminair_amd/csrc
g++ -std=c++20 -O2 -g -fPIC -shared -DLMN_EMU -Wall -Wno-unused-function -Wno-unknown-pragmas \
  -x c++ $SRC/kernels_trace.hip -x c++ $SRC/kernels_fft.hip -x c++ $SRC/kernels_merkle.hip -x c++ $SRC/kernels_logup.hip -x c++ $SRC/kernels_quotient.hip -x c++ $SRC/fft_fixed.hip -x c++ $SRC/components.cpp -x c++ $SRC/context.cpp -x c++ $SRC/trace_gen.cpp -x c++ $SRC/commit.cpp -x c++ $SRC/oods.cpp -x c++ $SRC/decommit.cpp -x c++ $SRC/quotients.cpp -x c++ $SRC/prove.cpp -x c++ $SRC/phase_trace.cpp -x c++ $SRC/phase_logup.cpp -x c++ $SRC/phase_composition.cpp -x c++ $SRC/phase_oods.cpp -x c++ $SRC/phase_fri.cpp -x c++ $SRC/phase_decommit.cpp -x c++ $SRC/shard.cpp -x c++ $SRC/ops.cpp -x c++ $SRC/verifier.cpp -x c++ $SRC/capi.cpp -x c++ $SRC/level2.cpp -x c++ emu_runtime.cpp \
  -o libluminair_emu.so
g++ -std=c++17 -O2 -fPIC -shared -Wall stub_rccl.cpp -o libstub_rccl.so -lrt
echo built tests/emu/libluminair_emu.so

#!/bin/bash
# Builds the TEST-ONLY emulation library (same sources, LMN_EMU): tests/emu/libluminair_emu.so
#   build_emu.sh            the plain build the CPU suite loads
#   build_emu.sh asan       libluminair_emu_asan.so: -fsanitize=address,undefined (load with LD_PRELOAD=libasan.so)
#   build_emu.sh tsan       libluminair_emu_tsan.so: -fsanitize=thread           (load with LD_PRELOAD=libtsan.so)
#   build_emu.sh batch [asan|tsan]   libluminair_emu_batch[_asan|_tsan].so: the lock-step batch library (-DLMN_BATCH + csrc/batch.cpp)
#                           on the emulation runtime: lmn_batch_* for the CPU suite and the sanitizers
# (tests/test_sanitizers.py drives the sanitizer builds; emu_runtime.cpp announces its fiber switches to them)
set -e
cd "$(dirname "$0")"
SRC=../../luminair_amd/csrc
BATCH=""
SUFFIX=""
if [ "${1:-}" = batch ]; then BATCH="-DLMN_BATCH -Dlmn=lmn_b -Wl,-Bsymbolic"; SUFFIX="_batch"; shift; fi
case "${1:-}" in
  asan) OPT="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"; OUT=libluminair_emu${SUFFIX}_asan.so ;;
  tsan) OPT="-O1 -g -fno-omit-frame-pointer -fsanitize=thread"; OUT=libluminair_emu${SUFFIX}_tsan.so ;;
  "") OPT="-O2 -g"; OUT=libluminair_emu${SUFFIX}.so ;;
  *) echo "usage: build_emu.sh [batch] [asan|tsan]"; exit 2 ;;
esac
UNITS="kernels_trace.hip kernels_fft.hip kernels_merkle.hip kernels_logup.hip kernels_quotient.hip fft_fixed.hip components.cpp context.cpp trace_gen.cpp commit.cpp oods.cpp decommit.cpp quotients.cpp prove.cpp phase_trace.cpp phase_logup.cpp phase_composition.cpp phase_oods.cpp phase_fri.cpp phase_decommit.cpp shard.cpp ops.cpp verifier.cpp capi.cpp level2.cpp"
ARGS=""
[ -n "$BATCH" ] && UNITS="$UNITS batch.cpp"
for u in $UNITS; do ARGS="$ARGS -x c++ $SRC/$u"; done
g++ -std=c++20 $OPT $BATCH -pthread -fPIC -shared -DLMN_EMU -Wall -Wno-unused-function -Wno-unknown-pragmas $ARGS -x c++ emu_runtime.cpp -o $OUT
[ -z "${1:-}" ] && [ -z "$BATCH" ] && g++ -std=c++17 -O2 -fPIC -shared -Wall stub_rccl.cpp -o libstub_rccl.so -lrt
echo built tests/emu/$OUT

#!/bin/bash
# transform launch knobs after the twiddle prefetch (columns per block, threads per block)
set -u
for E in "" "LMN_FFT_CPB=1" "LMN_FFT_CPB=2" "LMN_FFT_CPB=4" "LMN_FFT_THREADS=128" "LMN_FFT_THREADS=512" "LMN_FFT_XCD=0"; do
  env $E timeout 120 python tools/fft_knobs.py 2>/dev/null | tail -1
done

#!/bin/bash
python tools/fft_knobs.py 2>/dev/null
for c in 1 2 3 4 6; do LMN_FFT_CPB=$c python tools/fft_knobs.py 2>/dev/null; done
for t in 64 128 512; do LMN_FFT_THREADS=$t python tools/fft_knobs.py 2>/dev/null; done
LMN_FFT_XCD=0 python tools/fft_knobs.py 2>/dev/null

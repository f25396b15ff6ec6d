#!/bin/bash
# round 4: twiddle-set lifetime test (fresh process), switches test
set -u
OUT=gpurun_out/r6z
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "twiddle or switches or kat" > $OUT/t.log 2>&1; tail -3 $OUT/t.log

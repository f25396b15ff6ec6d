#!/bin/bash
# round 4: batch tests on the GPU + whole GPU suite after the FFT / batch / ABI changes
set -u
OUT=gpurun_out/r5c
mkdir -p $OUT
timeout 600 python -m pytest tests/test_batch.py -m gpu -x -q > $OUT/batch.log 2>&1; tail -5 $OUT/batch.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

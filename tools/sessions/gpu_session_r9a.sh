#!/bin/bash
# round 5, long soak of the final tree: 120 small and 16 big random pies with random protocol flags, each against the C
# oracle byte for byte; 24 small ones in lock-step batches are covered by tests/test_batch.py
set -u
OUT=gpurun_out/r9a
mkdir -p $OUT
SOAK_SEED0=1000 timeout 3000 python tools/soak_random.py 120 small flags > $OUT/soak_flags.txt 2>&1; tail -1 $OUT/soak_flags.txt
SOAK_SEED0=2000 timeout 3000 python tools/soak_random.py 16 big flags > $OUT/soak_big_flags.txt 2>&1; tail -1 $OUT/soak_big_flags.txt

#!/bin/bash
# round 4: parity soak with random component mixes at sizes around the tree-storage thresholds (up to 2^20 rows per table)
set -u
OUT=gpurun_out/r6u
mkdir -p $OUT
timeout 900 python tools/soak_random.py 16 big > $OUT/soak_big.txt 2> $OUT/soak_big.err; echo "rc=$?"; tail -2 $OUT/soak_big.txt; tail -2 $OUT/soak_big.err
timeout 600 python tools/soak_random.py 32 > $OUT/soak_small.txt 2> $OUT/soak_small.err; echo "rc=$?"; tail -1 $OUT/soak_small.txt

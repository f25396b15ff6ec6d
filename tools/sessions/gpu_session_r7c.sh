#!/bin/bash
# round 4: random parity soaks on the final tree (device-built shared twiddle sets grow as the sizes do)
set -u
OUT=gpurun_out/r7c
mkdir -p $OUT
timeout 600 python tools/soak_random.py 24 > $OUT/soak_small.txt 2> $OUT/soak_small.err; echo "rc=$?"; tail -1 $OUT/soak_small.txt
timeout 900 python tools/soak_random.py 12 big > $OUT/soak_big.txt 2> $OUT/soak_big.err; echo "rc=$?"; tail -1 $OUT/soak_big.txt

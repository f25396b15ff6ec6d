#!/bin/bash
# round 5, session i: random parity soaks on the final tree (device-resident transcript, phase split): 64 random pies under
# the PINNED flags, 64 with random protocol flags per pie (verified by the product verifier too), 12 big mixes (tables up to
# 2^20 rows), each against the C oracle byte for byte; then the soak under LMN_HOST_FS=1
set -u
OUT=gpurun_out/r8i
mkdir -p $OUT
timeout 1500 python tools/soak_random.py 64 > $OUT/soak_pinned.txt 2>&1; tail -1 $OUT/soak_pinned.txt
timeout 1500 python tools/soak_random.py 64 small flags > $OUT/soak_flags.txt 2>&1; tail -1 $OUT/soak_flags.txt
timeout 2400 python tools/soak_random.py 12 big flags > $OUT/soak_big_flags.txt 2>&1; tail -1 $OUT/soak_big_flags.txt
LMN_HOST_FS=1 timeout 1500 python tools/soak_random.py 32 small flags > $OUT/soak_flags_host_fs.txt 2>&1; tail -1 $OUT/soak_flags_host_fs.txt

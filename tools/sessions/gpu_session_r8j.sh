#!/bin/bash
# round 5, session j: blow-up 4 / 8 and flag-combination GPU tests; random parity soaks with random protocol flags per pie
# (32 small, 6 big mixes with tables up to 2^20 rows; 16 under LMN_HOST_FS=1), each against the C oracle byte for byte
set -u
OUT=gpurun_out/r8j
mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blowups or flag_combinations or switches or non_default" > $OUT/gpu_tests.log 2>&1; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
timeout 1500 python tools/soak_random.py 32 small flags > $OUT/soak_flags.txt 2>&1; tail -1 $OUT/soak_flags.txt
timeout 2400 python tools/soak_random.py 6 big flags > $OUT/soak_big_flags.txt 2>&1; tail -1 $OUT/soak_big_flags.txt
LMN_HOST_FS=1 timeout 1500 python tools/soak_random.py 16 small flags > $OUT/soak_flags_host_fs.txt 2>&1; tail -1 $OUT/soak_flags_host_fs.txt

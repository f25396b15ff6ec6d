#!/bin/bash
# round 6, session j: random parity soaks of the final tree against the C oracle byte for byte (new seeds; random protocol
# flags per pie; a third run with the host-side quotient step), then the default bench line with this round's counter summary.
set -u
OUT=gpurun_out/r10j
mkdir -p $OUT
SOAK_SEED0=3000 timeout 2400 python tools/soak_random.py 96 small flags > $OUT/soak_flags.txt 2>&1; tail -1 $OUT/soak_flags.txt
SOAK_SEED0=4000 timeout 2400 python tools/soak_random.py 16 big flags > $OUT/soak_big_flags.txt 2>&1; tail -1 $OUT/soak_big_flags.txt
LMN_HOST_QUOT=1 SOAK_SEED0=5000 timeout 1200 python tools/soak_random.py 32 small flags > $OUT/soak_host_quot.txt 2>&1; tail -1 $OUT/soak_host_quot.txt
LMN_ROWS_FUSION=1 SOAK_SEED0=6000 timeout 1200 python tools/soak_random.py 8 big > $OUT/soak_rows_fusion.txt 2>&1; tail -1 $OUT/soak_rows_fusion.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 600 $OUT/bench_default.json

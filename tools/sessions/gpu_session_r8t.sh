#!/bin/bash
# round 5, session t: random parity soaks on the final tree (random protocol flags per pie, each against the C oracle byte
# for byte: 48 small, 6 big mixes with tables up to 2^20 rows, 16 with the steps as launches of their own, 16 with the host
# transcript), solo latency of the BASELINE configs
set -u
OUT=gpurun_out/r8t
mkdir -p $OUT
timeout 1500 python tools/soak_random.py 48 small flags > $OUT/soak_flags.txt 2>&1; tail -1 $OUT/soak_flags.txt
timeout 2400 python tools/soak_random.py 6 big flags > $OUT/soak_big_flags.txt 2>&1; tail -1 $OUT/soak_big_flags.txt
LMN_CHAN_STEP_SEPARATE=1 timeout 1500 python tools/soak_random.py 16 small flags > $OUT/soak_flags_separate_steps.txt 2>&1; tail -1 $OUT/soak_flags_separate_steps.txt
LMN_HOST_FS=1 timeout 1500 python tools/soak_random.py 16 small flags > $OUT/soak_flags_host_fs.txt 2>&1; tail -1 $OUT/soak_flags_host_fs.txt
timeout 900 python tools/config_latency.py > $OUT/config_latency.jsonl 2> $OUT/config_latency.err; cut -c1-160 $OUT/config_latency.jsonl

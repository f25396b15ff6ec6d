#!/bin/bash
# r3 session B: which ingredient of the host_rows failure is it?  one process per case
set -u
OUT=gpurun_out/r3c
mkdir -p $OUT
run() { name=$1; shift; timeout 300 python tools/repro_host_rows.py --rounds 40 --per-round 48 "$@" > $OUT/$name.jsonl 2> $OUT/$name.err; echo "$name rc=$?" | tee -a $OUT/summary.txt; cat $OUT/$name.jsonl | tee -a $OUT/summary.txt; tail -2 $OUT/$name.err | tee -a $OUT/summary.txt; }
run private_warm --cases private --warm
run shared_warm --cases shared --warm
run private_cold --cases private
run shared_cold --cases shared
run fresh --cases fresh
run private_warm_mode1 --cases private --warm

#!/bin/bash
# which property of the plain-op quarter round keeps it at 0.25 wave-instr/clk/SIMD
set -u
OUT=gpurun_out/r4d
mkdir -p $OUT
timeout 300 tools/bin/mb_reconcile why > $OUT/why.txt 2> $OUT/why.err; echo "rc=$?"
cut -c1-75 $OUT/why.txt

#!/bin/bash
# does the order of INDEPENDENT instructions inside one wave change the VALU issue rate?  (dependent back-to-back vs interleaved states)
set -u
OUT=gpurun_out/r4c
mkdir -p $OUT
timeout 300 tools/bin/mb_reconcile ilv > $OUT/interleave.txt 2> $OUT/interleave.err; echo "rc=$?"
cat $OUT/interleave.txt

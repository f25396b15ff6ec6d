#!/bin/bash
# M31 butterflies: compiler order vs class-grouped phases with s_setprio
set -u
OUT=gpurun_out/r4g
mkdir -p $OUT
timeout 300 tools/bin/mb_reconcile bfly > $OUT/bfly.txt 2> $OUT/bfly.err; echo "rc=$?"
cut -c1-75 $OUT/bfly.txt; tail -3 $OUT/bfly.err

#!/bin/bash
# can s_setprio line up the fast-class runs of co-resident waves so that they co-issue?
set -u
OUT=gpurun_out/r4e
mkdir -p $OUT
timeout 300 tools/bin/mb_reconcile prio > $OUT/prio.txt 2> $OUT/prio.err; echo "rc=$?"
cut -c1-75 $OUT/prio.txt; tail -3 $OUT/prio.err

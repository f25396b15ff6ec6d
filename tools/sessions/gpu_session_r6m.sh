#!/bin/bash
# round 4: host-side wall-clock marks of one solo proof (LMN_HOST_PROFILE=1), product library
set -u
OUT=gpurun_out/r6m
mkdir -p $OUT
LMN_HOST_PROFILE=1 python tools/ablate_throughput.py 1 4 2> $OUT/host_marks.txt > $OUT/out.json
grep -n "\[host\]" $OUT/host_marks.txt | tail -40

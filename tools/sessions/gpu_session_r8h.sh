#!/bin/bash
# round 5, session h: one-proof kernel timeline with the device-resident transcript (and with LMN_HOST_FS=1 beside it)
set -u
OUT=gpurun_out/r8h
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 8 --warmup 2"
for v in dev host; do
  if [ $v = host ]; then export LMN_HOST_FS=1; else unset LMN_HOST_FS; fi
  rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof_$v -o ks -- $BENCH > $OUT/bench_$v.json 2> $OUT/prof_$v.log
  KT=$(find $OUT/prof_$v -name '*kernel_trace.csv' | head -1)
  python tools/timeline.py $KT v > $OUT/timeline_$v.txt
  rm -rf $OUT/prof_$v
done
head -45 $OUT/timeline_dev.txt; grep -n "end-to-end\|_gaps" $OUT/timeline_dev.txt $OUT/timeline_host.txt

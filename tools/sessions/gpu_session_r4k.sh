#!/bin/bash
# kernel durations with 8 proofs in flight (stretch against the solo durations)
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace -d $OUT/r4_conc -o conc -- python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 64 --warmup 8 > $OUT/r4_bench_under_rocprof_inflight8.json 2> $OUT/r4_conc.log
python tools/overlap.py $(find $OUT/r4_conc -name '*kernel_trace.csv' | head -1) 16 64 > $OUT/r4_overlap.txt
rm -rf $OUT/r4_conc
cat $OUT/r4_overlap.txt

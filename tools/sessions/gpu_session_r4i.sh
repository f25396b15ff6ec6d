#!/bin/bash
# solo kernel durations (one proof in flight) after the issue-phase changes
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
BENCH="python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1"
rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/r4_prof -o ks -- $BENCH --steps 32 --warmup 4 > $OUT/r4_bench_under_rocprof_inflight1.json 2> $OUT/r4_prof.log
find $OUT/r4_prof -name '*kernel_stats.csv' -exec cp {} $OUT/r4_kernel_stats_inflight1.csv \;
rm -rf $OUT/r4_prof
python - <<'PY'
import csv
for r in list(csv.DictReader(open("gpurun_out/r4_kernel_stats_inflight1.csv")))[:16]:
    print(r["Name"][:45].ljust(46), r["Calls"].rjust(5), ("%.1f" % (float(r["AverageNs"])/1e3)).rjust(8), r["Percentage"])
PY
tail -1 $OUT/r4_bench_under_rocprof_inflight1.json | cut -c1-200

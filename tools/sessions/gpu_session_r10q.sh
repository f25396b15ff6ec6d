#!/bin/bash
# round 6, session q: the committed bench lines of the final tree (bench.py: 24 proofs in flight, sub-results on 8 contexts):
# the driver's command x3, the default run with the CPU baseline, the same under rocprofv3 --kernel-trace --stats.
set -u
OUT=gpurun_out/r10q
mkdir -p $OUT
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; done
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 900 rocprofv3 --output-format csv --kernel-trace --stats -d $OUT/prof -o ks -- python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
find $OUT/prof -name '*kernel_stats.csv' -exec cp {} $OUT/kernel_stats_default_cmd.csv \;
rm -rf $OUT/prof
python - <<PY
import json
for f in ["driver_cmd_1","driver_cmd_2","driver_cmd_3","bench_default","bench_under_rocprof"]:
    try:
        d=json.loads(open("$OUT/%s.json"%f).read().strip().splitlines()[-1])
        r=d["roofline"]; c=d.get("cpu_baseline") or {}
        print(f, round(d["value"],1), "ms_per_step", round(d["ms_per_step"],2), "solo", round(d["prove_latency_ms"],3), "frac", round(r["frac"],3), "avg_launch_ms", round(r["avg_launch_ms"],4), "alu", round(r["alu_ceiling"]["frac"],3), "host_rows", d.get("host_rows_proofs_per_s"), "pinned", d.get("host_rows_pinned_proofs_per_s"), "cpu", c.get("value"), c.get("cores"), d["errors"])
    except Exception as e: print(f, "ERR", e)
PY
head -8 $OUT/kernel_stats_default_cmd.csv | cut -c1-160

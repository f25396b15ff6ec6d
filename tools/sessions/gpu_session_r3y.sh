#!/bin/bash
set -u
OUT=gpurun_out/r3y
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --output-format csv --kernel-trace -d $OUT/conc -o conc -- python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 192 --warmup 16 > $OUT/bench_under_rocprof_inflight8.json 2> $OUT/conc.log
KT=$(find $OUT/conc -name '*kernel_trace.csv' | head -1)
python tools/overlap.py $KT 24 192 | tee $OUT/overlap.txt
rocprofv3 --output-format csv --kernel-trace -d $OUT/conc20 -o conc -- python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 20 --warmup 5 > $OUT/bench_under_rocprof_k20.json 2> $OUT/conc20.log
KT=$(find $OUT/conc20 -name '*kernel_trace.csv' | head -1)
python tools/overlap.py $KT 13 20 | tee -a $OUT/overlap.txt
python - <<'PY'
import json
for f in ("bench_under_rocprof_inflight8","bench_under_rocprof_k20"):
    d=json.loads(open("gpurun_out/r3y/%s.json"%f).read().strip().splitlines()[-1]); print(f, round(d["value"],1))
PY
rm -rf $OUT/conc $OUT/conc20

#!/bin/bash
# page-locked host rows: parity test + bench sub-results
set -u
OUT=gpurun_out/r4r
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "page_locked or host_rows" > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | tail -3
for i in 1 2 3; do
timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print(round(d["value"],1), d["errors"], {k:(round(d[k]["value"],1), round(d[k].get("prove_latency_ms",0),2)) for k in ("host_rows","host_rows_pinned") if "value" in d.get(k,{})}, {k:d[k].get("error") for k in ("host_rows","host_rows_pinned") if "error" in d.get(k,{})})
PY
done

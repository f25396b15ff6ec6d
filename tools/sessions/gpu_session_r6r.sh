#!/bin/bash
# round 4: whole GPU suite + smoke on the final tree (threshold-size tests, switch test included)
set -u
OUT=gpurun_out/r6r
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6r/bench_driver_cmd.json").read().strip().splitlines()[-1])
print("driver cmd", round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "errors", d["errors"], "host_rows", round(d["host_rows"]["value"]), round(d["host_rows_pinned"]["value"]))
PY

#!/bin/bash
# round 4: final tree (adaptive host waits included): whole GPU suite, smoke, default bench, the driver's command x3
set -u
OUT=gpurun_out/r6x
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6x/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "host_rows", round(d["host_rows"]["value"]), "pinned", round(d["host_rows_pinned"]["value"]), "2b", round(d["config_2b"]["value"]), "mul", round(d["mul_only"]["value"]), "c3", round(d["config_3"]["value"]), "small", round(d["small_proofs"]["reference_shape_32x32_add"]["value"]), round(d["small_proofs"]["config_4"]["value"]), "errors", d["errors"])
PY
for i in 1 2 3; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'short_region': d['short_region']['value'], 'solo_ms': d['prove_latency_ms']}))" | tee -a $OUT/driver_cmd_runs.jsonl
done

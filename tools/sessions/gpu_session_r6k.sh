#!/bin/bash
# round 4, final kernels (+ join fusion, 8-wave quotient launches): whole GPU suite, smoke, default
# bench, the driver's command x6, profile set (tools/profile_round.sh), marginal costs per kernel family (ablation build)
set -u
OUT=gpurun_out/r6k
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; grep -E "passed|failed|rror" $OUT/gpu_tests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6k/bench_default.json").read().strip().splitlines()[-1])
print("default", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", d["prove_latency_ms"], "host_rows", d["host_rows"].get("value"), "pinned", d["host_rows_pinned"].get("value"), "2b", d["config_2b"].get("value"), "mul", d["mul_only"].get("value"), "c3", d["config_3"].get("value"), "errors", d["errors"])
print("roofline", d["roofline"])
PY
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'short_region': d['short_region']['value'], 'solo_ms': d['prove_latency_ms']}))" | tee -a $OUT/driver_cmd_runs.jsonl
done
bash tools/profile_round.sh r6k > $OUT/profile_round.log 2>&1; tail -30 $OUT/profile_round.log
cp luminair_amd/csrc/libluminair_hip.so /tmp/product.so
cp tools/bin/variants/ablate.so luminair_amd/csrc/libluminair_hip.so
for m in 0 1 2 4 8 16 32 63 0; do
  LMN_ABLATE=$m timeout 300 python tools/ablate_throughput.py 8 192 2>> $OUT/ablate.err | tee -a $OUT/ablate.jsonl
done
cp /tmp/product.so luminair_amd/csrc/libluminair_hip.so

#!/bin/bash
# round 4: product with the 8-wave quotient launches + join fusion: parity, throughput, solo latency of the BASELINE configs
set -u
OUT=gpurun_out/r6j
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q > $OUT/parity.log 2>&1; grep -E "passed|failed|rror" $OUT/parity.log | tail -3
for i in 1 2 3; do
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$i.json 2> $OUT/bench_$i.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$i.json").read().strip().splitlines()[-1])
print("product", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), {k:v for k,v in d["stage_ms"].items() if k in ("quotients_ms","fri_ms","merkle_fused_ms","fft_ms")})
PY
done
timeout 900 python tools/config_latency.py > $OUT/config_latency.jsonl 2> $OUT/config_latency.err; cat $OUT/config_latency.jsonl | cut -c1-300

#!/bin/bash
# round 4: the driver's command six times on the final tree
set -u
OUT=gpurun_out/r7e
mkdir -p $OUT
for i in 1 2 3 4 5 6; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({'value': d['value'], 'ms_per_step': d['ms_per_step'], 'short_region': d['short_region']['value'], 'solo_ms': d['prove_latency_ms'], 'host_cpu_ms_per_proof': d['host_cpu_ms_per_proof']}))" | tee -a $OUT/driver_cmd_runs.jsonl
done

#!/bin/bash
set -u
OUT=gpurun_out/r4a
mkdir -p $OUT
for m in 0 1 2; do
  echo "== LMN_SYNC_MODE=$m"
  LMN_SYNC_MODE=$m python tools/small_proof_throughput.py 4 8 16 2>/dev/null | grep device
  for rep in 1 2; do
    LMN_SYNC_MODE=$m timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  headline k20', round(d['value'],1), 'lat', round(d['prove_latency_ms'],3))"
  done
  LMN_SYNC_MODE=$m timeout 300 python bench.py --gpus 1 --no-cpu-baseline --no-extras --no-anchor 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  headline k192', round(d['value'],1), 'lat', round(d['prove_latency_ms'],3))"
done

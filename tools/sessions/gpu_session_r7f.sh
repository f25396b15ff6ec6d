#!/bin/bash
# round 4: OODS sums, FRI roots and the decommitment gather written straight into page-locked host memory (3 copy commands
# less per proof): parity (incl. batches and sharded ranks on one GPU), solo latency / throughput / host CPU against the previous library
set -u
OUT=gpurun_out/r7f
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_batch.py tests/test_sharded_prove.py -m gpu -x -q > $OUT/parity.log 2>&1; tail -1 $OUT/parity.log
cp luminair_amd/csrc/libluminair_hip.so /tmp/new.so
for v in new prev new prev new prev; do
  if [ $v = prev ]; then cp tools/bin/variants/prev.so luminair_amd/csrc/libluminair_hip.so; else cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "cpu/proof", d["host_cpu_ms_per_proof"])
PY
  TAG=$v timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log
done
cp /tmp/new.so luminair_amd/csrc/libluminair_hip.so

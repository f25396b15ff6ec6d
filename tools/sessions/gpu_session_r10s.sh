#!/bin/bash
# round 6, session s: hardware queues per priority level (GPU_MAX_HW_QUEUES) 2 .. 6 at 12 / 24 / 36 proofs in flight.
set -u
OUT=gpurun_out/r10s
mkdir -p $OUT
for rep in 1 2 3; do
for q in 2 3 4 5 6; do
for inf in 12 24 36; do
  GPU_MAX_HW_QUEUES=$q timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $inf > $OUT/bench_q${q}_${inf}_$rep.json 2> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_q${q}_${inf}_$rep.json").read().strip().splitlines()[-1])
print("GPU_MAX_HW_QUEUES $q inflight $inf", round(d["value"],1), "short", round(d["short_region"]["value"],1))
PY
done
done
done

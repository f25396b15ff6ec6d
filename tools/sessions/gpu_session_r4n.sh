#!/bin/bash
# proofs in flight after the co-issue change
set -u
OUT=gpurun_out/r4n
mkdir -p $OUT
for N in 4 6 8 10 12 16; do
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $N --steps 384 --warmup 32 2>/dev/null | tail -1 > $OUT/inflight$N.json
  python - <<PY
import json
d=json.load(open("$OUT/inflight$N.json")); print("in flight", $N, round(d["value"],1))
PY
done

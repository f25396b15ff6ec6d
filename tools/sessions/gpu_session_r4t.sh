#!/bin/bash
# the driver's 20-step region against the number of proofs in flight (20 = 4 x 5 = 5 x 4 = 10 x 2 = 20 x 1; 8 leaves a ragged tail)
set -u
OUT=gpurun_out/r4t
mkdir -p $OUT
for N in 4 5 8 10 20; do
  for i in 1 2 3 4; do
    timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $N --steps 20 --warmup 5 2>/dev/null | tail -1 > $OUT/n${N}_$i.json
  done
  python - <<PY
import json
v=[round(json.load(open("$OUT/n${N}_%d.json"%i))["value"],1) for i in (1,2,3,4)]
print("in flight", $N, v)
PY
done

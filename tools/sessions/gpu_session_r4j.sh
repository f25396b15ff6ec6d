#!/bin/bash
# Merkle subtree depth per lane (grid size / occupancy) after the co-issue change
set -u
OUT=gpurun_out/r4j
mkdir -p $OUT
for SUB in 3 2 1; do
  LMN_MERKLE_SUB=$SUB timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 96 --warmup 16 2>/dev/null | tail -1 > $OUT/sub$SUB.json
  LMN_MERKLE_SUB=$SUB timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight 1 --steps 32 --warmup 4 2>/dev/null | tail -1 > $OUT/sub${SUB}_solo.json
  python - <<PY
import json
d=json.load(open("$OUT/sub$SUB.json")); e=json.load(open("$OUT/sub${SUB}_solo.json"))
print("sub", $SUB, "8 ctx", round(d["value"],1), "solo", round(e["value"],1), "merkle launch us", round(e["roofline"]["avg_launch_ms"]*1e3,1))
PY
done

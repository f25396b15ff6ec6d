#!/bin/bash
# round 5, session y: solo latency of the tree against the previous commit (tools/bin/variants/prev_x.so), five alternations
set -u
OUT=gpurun_out/r8y; mkdir -p $OUT
LIB=luminair_amd/csrc/libluminair_hip.so
cp $LIB /tmp/new.so
for v in new prev new prev new prev new prev new prev; do
  [ $v = new ] && cp /tmp/new.so $LIB
  [ $v = prev ] && cp tools/bin/variants/prev_x.so $LIB
  timeout 300 python bench.py --no-cpu-baseline --no-extras --no-anchor --steps 6 --warmup 2 > $OUT/b.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$OUT/b.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3))
PY
done
cp /tmp/new.so $LIB

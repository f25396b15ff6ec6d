#!/bin/bash
# round 6, session o: is the gain of more proofs in flight (session n) the longer timed region? Equal numbers of proofs: 576 and 2 304.
set -u
OUT=gpurun_out/r10o
mkdir -p $OUT
run() {
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $1 --steps $2 --warmup $3 > $OUT/bench_$1_$2_$4.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$1_$2_$4.json").read().strip().splitlines()[-1])
print("inflight $1 steps $2 (proofs", $1*$2, ")", round(d["value"],1))
PY
}
for rep in 1 2 3; do
  run 8 72 6 $rep
  run 24 24 2 $rep
  run 12 48 4 $rep
  run 8 288 6 $rep
  run 24 96 2 $rep
  run 12 192 4 $rep
done

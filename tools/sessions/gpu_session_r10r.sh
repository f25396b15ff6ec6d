#!/bin/bash
# round 6, session r: the CU-masked streams of session l (":1" masks) at 24 / 36 / 48 proofs in flight against the default.
set -u
OUT=gpurun_out/r10r
mkdir -p $OUT
run() {
  if [ "$1" = none ]; then unset LMN_CU_SPLIT; else export LMN_CU_SPLIT=$1; fi
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $2 > $OUT/bench_${1/:/x}_$2_$3.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${1/:/x}_$2_$3.json").read().strip().splitlines()[-1])
print("split $1 inflight $2", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3))
PY
}
for rep in 1 2 3; do
  for inf in 24 48; do
    for sp in none 2:1 4:1 8:1; do run $sp $inf $rep; done
  done
done

#!/bin/bash
# round 6, session m: session l's side result - streams without the round-robin priorities made more proofs/s at 16 proofs in
# flight.  Matrix: LMN_STREAM_PRIO_CYCLE 1 | 0  x  8 / 12 / 16 / 24 / 32 proofs in flight, alternating.
set -u
OUT=gpurun_out/r10m
mkdir -p $OUT
run() {
  LMN_STREAM_PRIO_CYCLE=$1 timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $2 > $OUT/bench_$1_$2_$3.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$1_$2_$3.json").read().strip().splitlines()[-1])
print("prio_cycle $1 inflight $2", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "host_cpu", d["host_cpu_ms_per_proof"])
PY
}
for rep in 1 2 3; do
  for inf in 8 12 16 24 32; do
    for c in 1 0; do run $c $inf $rep; done
  done
done

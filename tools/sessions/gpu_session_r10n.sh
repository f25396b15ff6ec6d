#!/bin/bash
# round 6, session n: proofs in flight, finer: multiples of the three stream priority levels against their neighbours.
set -u
OUT=gpurun_out/r10n
mkdir -p $OUT
run() {
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor --inflight $1 > $OUT/bench_$1_$2.json 2> $OUT/bench.err || tail -3 $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$1_$2.json").read().strip().splitlines()[-1])
print("inflight $1", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "host_cpu", d["host_cpu_ms_per_proof"])
PY
}
for rep in 1 2 3; do
  for inf in 8 9 10 12 15 18 20 21 24 27 30 36 48; do run $inf $rep; done
done
# the driver's command at 8 / 12 / 24
for inf in 8 12 24 8 12 24; do
  timeout 600 python bench.py --steps 20 --warmup 5 --inflight $inf --no-cpu-baseline --no-extras --no-anchor > $OUT/driver_$inf.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/driver_$inf.json").read().strip().splitlines()[-1])
print("driver command, inflight $inf", round(d["value"],1), "ms_per_step", round(d["ms_per_step"],3))
PY
done

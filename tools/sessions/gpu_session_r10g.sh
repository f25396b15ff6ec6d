#!/bin/bash
# round 6, session g: device quotient step with the polling limit at 3000 us (a solo proof's one 2 ms wait stays a polling
# wait): solo latency / throughput against LMN_HOST_QUOT=1, alternating; small proofs; host CPU per proof.
set -u
OUT=gpurun_out/r10g
mkdir -p $OUT
for rep in 1 2 3 4; do
for v in dev host; do
  unset LMN_HOST_QUOT
  [ $v = host ] && export LMN_HOST_QUOT=1
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_${v}_$rep.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_${v}_$rep.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3), "host_cpu_ms", d["host_cpu_ms_per_proof"])
PY
done
done
unset LMN_HOST_QUOT
for rep in 1 2; do
TAG=dev timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1
LMN_HOST_QUOT=1 TAG=host timeout 120 python tools/small_latency.py 101 2>> $OUT/err.log | tail -1
done
timeout 600 python tools/config_latency.py > $OUT/config_latency_dev.jsonl 2>> $OUT/err.log; cat $OUT/config_latency_dev.jsonl | cut -c1-200
LMN_HOST_QUOT=1 timeout 600 python tools/config_latency.py > $OUT/config_latency_host.jsonl 2>> $OUT/err.log; cat $OUT/config_latency_host.jsonl | cut -c1-200

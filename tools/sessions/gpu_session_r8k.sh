#!/bin/bash
# round 5, session k: the smaller quotient column on a second stream only while the proof has the GPU to itself
# (LMN_FRI_OVERLAP unset = auto) against never (0) and always (1): parity subset, then solo latency / throughput alternating
set -u
OUT=gpurun_out/r8k
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_batch.py -m gpu -x -q -k "kat or full or config or random or ragged or sizes or blowups or batch" > $OUT/parity.log 2>&1; grep -n "passed\|failed" $OUT/parity.log | tail -2
for v in auto never auto never always auto never; do
  unset LMN_FRI_OVERLAP
  [ $v = never ] && export LMN_FRI_OVERLAP=0
  [ $v = always ] && export LMN_FRI_OVERLAP=1
  timeout 600 python bench.py --no-cpu-baseline --no-extras --no-anchor > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_$v.json").read().strip().splitlines()[-1])
print("$v", round(d["value"],1), "short", round(d["short_region"]["value"],1), "solo", round(d["prove_latency_ms"],3), "p95", round(d["prove_latency_p95_ms"],3))
PY
done
unset LMN_FRI_OVERLAP
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/driver_cmd_$i.json 2> $OUT/driver_cmd_$i.err; python -c "
import json; d=json.loads(open('$OUT/driver_cmd_$i.json').read().strip().splitlines()[-1]); print('driver cmd', round(d['value'],1), round(d['prove_latency_ms'],3), d['errors'])"; done
